"""Worker for multi-process tests: one rank per process, torch.distributed (gloo or nccl), all ranks
on cuda:0 unless LOCAL_RANK maps to distinct devices. Checks the sharded V-cycle against a
single-rank run on rank 0."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=a.backend)
    from pressurepoissonsolver_amd import capi, dist as tedist, problems
    n = 8
    mesh = capi.Mesh.uniform(3, 2)
    H = capi.Hierarchy(mesh, n, rank=rank, nranks=world)
    g = capi.GMG(H, device=dev)
    tedist.attach(g, dist)
    t = H.tables(0)
    f_all = problems.random_rhs(t["id"], n ** 3)
    idx = H.l2g(0)
    f = g.new_vector(0, f_all.reshape(-1, n ** 3)[idx].ravel())
    u = g.new_vector(0)
    g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), f, u)
    mine = torch.zeros(len(t["id"]) * n ** 3, dtype=torch.float64)
    mine.view(-1, n ** 3)[torch.as_tensor(idx, dtype=torch.long)] = torch.from_numpy(u.download()).view(-1, n ** 3)
    if a.backend == "nccl":
        mine = mine.cuda()
    dist.all_reduce(mine)
    if rank == 0:
        H1 = capi.Hierarchy(mesh, n)
        g1 = capi.GMG(H1, device=dev)
        u1 = g1.new_vector(0)
        g1.cycle(g1.default_opts(smoother=capi.SMOOTH_RBGS), g1.new_vector(0, f_all), u1)
        assert np.array_equal(mine.cpu().numpy(), u1.download()), "sharded V-cycle differs from single-rank"
        print("MR_WORKER_OK", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
