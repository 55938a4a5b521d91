"""Worker for multi-process tests: one rank per process, torch.distributed (gloo or nccl), all ranks
on cuda:0 unless LOCAL_RANK maps to distinct devices. With TE_OVERLAP_MIN=0 the level-0 face exchange
is issued on the library's communication stream by the overlapped operator application and on the
solver stream by the fused cycle: the sequence apply -> cycle -> BiCGStab -> apply below visits both,
in both orders, through ONE attached callback. Every result is compared with a single-rank run on
rank 0 (bit for bit for the per-cell operations)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--divides", type=int, default=2)
    ap.add_argument("--hang-rank", type=int, default=-1, help="this rank never enters the cycle (watchdog test)")
    ap.add_argument("--push", action="store_true", help="the direct-store transport (te_gmg_use_push) for the exchanges that have one: "
                    "real hipIpc mappings between the processes, the attached back-end for everything else")
    ap.add_argument("--cells", type=int, default=8, help="cells per patch axis")
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--smoother", default="rbgs", help="rbgs | patch_solve (the reference's block Jacobi)")
    ap.add_argument("--push-fault", action="store_true", help="TE_PUSH_FAULT: the direct-store transport never delivers; te_gmg_autotune must "
                    "reject it on all ranks and everything must then run correctly through the attached back-end")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=a.backend)
    from pressurepoissonsolver_amd import capi, dist as tedist, problems, solver
    n = a.cells
    nc = n ** a.dim
    SM = capi.SMOOTH_PATCH_SOLVE if a.smoother == "patch_solve" else capi.SMOOTH_RBGS
    mesh = capi.Mesh.uniform(a.dim, a.divides)
    H = capi.Hierarchy(mesh, n, rank=rank, nranks=world)
    g = capi.GMG(H, device=dev)
    tedist.attach(g, dist)
    report = None
    if a.push:
        g.use_push(True)
        # the transports are compared on this very machine: identical results, then the faster one; force the direct one afterwards
        if a.push_fault:
            g.set_option("TE_PUSH_FAULT", "1")
        _, report = g.autotune(g.default_opts(smoother=SM), reps=3)
        if a.push_fault:
            assert "REJECTED" in report, report
            try:
                g.use_push(True)  # (not usable any more for this solver: set up again would be a second collective; it must refuse or stay off)
            except capi.TeError:
                pass
            g.use_push(False)
        else:
            assert "transport:" in report and "REJECTED" not in report and "results identical" in report, report
            g.use_push(True)
    t = H.tables(0)
    f_all = problems.random_rhs(t["id"], nc)
    b_all, _ = (problems.init_dirichlet if a.dim == 3 else problems.init_dirichlet_2d)(t, n)
    idx = H.l2g(0)
    local = lambda v: v.reshape(-1, nc)[idx].ravel()  # noqa: E731
    opts = g.default_opts(smoother=SM)

    def gather(vec):
        mine = torch.zeros(len(t["id"]) * nc, dtype=torch.float64)
        mine.view(-1, nc)[torch.as_tensor(idx, dtype=torch.long)] = torch.from_numpy(vec.download()).view(-1, nc)
        if a.backend == "nccl":
            mine = mine.cuda()
        dist.all_reduce(mine)
        return mine.cpu().numpy()

    if a.hang_rank >= 0:  # watchdog test: the peers' first exchange can never complete
        import time
        if rank == a.hang_rank:
            time.sleep(90)
        else:
            fh, uh = g.new_vector(0, local(f_all)), g.new_vector(0)
            g.cycle(opts, fh, uh)
            g.sync()
            print("MR_WORKER_OK (unexpected: the exchange completed)", flush=True)
        return
    got = {}
    f, u, au = g.new_vector(0, local(f_all)), g.new_vector(0), g.new_vector(0)
    g.apply(f, au)                      # overlapped (communication stream) when TE_OVERLAP_MIN=0
    got["apply_first"] = gather(au)
    g.cycle(opts, f, u)                 # fused cycle: the same exchange in line on the solver stream
    got["cycle"] = gather(u)
    b, x = g.new_vector(0, local(b_all)), g.new_vector(0)
    its, rr = solver.bicgstab(g, x, b, opts)  # native te_bicgstab, scalars through the all-reduce dist.attach registered
    got["bicg"] = gather(x)
    g.apply(u, au)                      # and the overlapped path again after the in-line one
    got["apply_last"] = gather(au)
    if rank == 0:
        H1 = capi.Hierarchy(mesh, n)
        g1 = capi.GMG(H1, device=dev)
        f1, u1, au1 = g1.new_vector(0, f_all), g1.new_vector(0), g1.new_vector(0)
        g1.apply(f1, au1)
        assert np.array_equal(got["apply_first"], au1.download()), "sharded apply differs from single-rank"
        g1.cycle(g1.default_opts(smoother=SM), f1, u1)
        assert np.array_equal(got["cycle"], u1.download()), "sharded V-cycle differs from single-rank"
        x1 = g1.new_vector(0)
        its1, rr1 = g1.bicgstab(x1, g1.new_vector(0, b_all), g1.default_opts(smoother=SM))
        want = x1.download()
        assert abs(its - its1) <= 1 and rr <= 1e-12, (its, its1, rr)
        assert np.linalg.norm(got["bicg"] - want) <= 1e-9 * np.linalg.norm(want), "sharded BiCGStab differs"
        g1.apply(u1, au1)
        assert np.array_equal(got["apply_last"], au1.download()), "sharded apply (after the cycle) differs"
        print("MR_WORKER_OK", report or "", flush=True)
    assert not g.push_failed(), "a direct-store wait gave up (or a rejected transport left its error word set)"
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
