#!/usr/bin/env python3
"""Tooling: the per-rank time budget of an N-rank 512^3 V-cycle, measured on ONE GPU (there is no multi-GPU node to
measure on): rank r of N builds its shard of the hierarchy and runs its cycle ALONE on the device, with the native RCCL
back-end in loop-back mode (TE_RCCL_LOOPBACK: every peer replaced by the rank itself -- the same ncclGroupStart /
ncclRecv / ncclSend / ncclGroupEnd calls with the same message sizes between scratch buffers). Real in this
measurement: the rank's own kernels with the GPU to themselves, the host's enqueue time including the RCCL group
calls, RCCL's launch overhead. Not real: the wire (xGMI) and the waiting for peers; the exchanged data are garbage, so
nothing is checked here (tests/test_gpu_multirank_production.py checks the same plans bit for bit).

    python tools/mr8_budget.py [--size 512] [--smoother rbgs] [--out profiles/r03_mr8_budget.txt]

Per (N, TE_AGGLOMERATE) and for rank 0 (owns the gathered coarse levels) and rank N-1 (does not):
  host   = median host time to enqueue one cycle (stream drained before each)
  wall   = back-to-back cycles, per cycle (GPU time when the host stays ahead)
  serial tail = wall(rank 0) - wall(rank N-1)
and the kernel-class table of rank 0. DESIGN.md 6 turns these into the predicted 2/4/8-GPU numbers."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--smoother", default="rbgs")
ap.add_argument("--out", default=None)
ap.add_argument("--ranks", default="1,2,4,8")
ap.add_argument("--agg", default="16,4,0")
ap.add_argument("--dim", type=int, default=3)
ap.add_argument("--patch", type=int, default=None, help="cells per patch axis (default 32 in 3D, 64 in 2D)")
ap.add_argument("--mesh", default=None, help="octree file instead of the uniform grid (config C4: tests/golden/2refine.bin --divide 2)")
ap.add_argument("--divide", type=int, default=0)
ap.add_argument("--push", action="store_true", help="the direct-store transport (te_gmg_use_push) in loop-back form: every peer's buffers and "
                "flags are this rank's own -- the two launches per exchange are real, the wire and the waiting for peers are not")
a = ap.parse_args()

os.environ["TE_RCCL_LOOPBACK"] = "1"
os.environ["TE_NO_VERIFY"] = "1"
from pressurepoissonsolver_amd import capi, dist as tedist  # noqa: E402

n = a.patch or (32 if a.dim == 3 else 64)
div = int(round(np.log2(a.size // n)))
sm = {"rbgs": capi.SMOOTH_RBGS, "patch_solve": capi.SMOOTH_PATCH_SOLVE}[a.smoother]
lines = []


def emit(s=""):
    print(s, flush=True)
    lines.append(s)


def run(nranks, rank, agg):
    os.environ["TE_AGGLOMERATE"] = str(agg)
    if a.mesh:
        mesh = capi.Mesh.read(a.mesh, a.dim)
        for _ in range(a.divide):
            mesh.refine_leaves()
    else:
        mesh = capi.Mesh.uniform(a.dim, div)
    H = capi.Hierarchy(mesh, n, rank=rank, nranks=nranks)
    g = capi.GMG(H)
    if nranks > 1:
        tedist.attach_rccl(g, None, 0, 1)  # a communicator of one: this rank is every peer (loop-back)
        if a.push:
            g.use_push(True)
    f, u = g.new_vector(0), g.new_vector(0)
    g.init_problem(f, None, problem=capi.PROBLEM_RANDOM)
    o = g.default_opts(smoother=sm)
    for _ in range(5):
        g.cycle(o, f, u)
    g.sync()
    tq = []
    for _ in range(40):
        g.sync()
        t0 = time.perf_counter()
        g.cycle(o, f, u)
        tq.append(time.perf_counter() - t0)
    g.sync()
    t0 = time.perf_counter()
    for _ in range(40):
        g.cycle(o, f, u)
    g.sync()
    wall = (time.perf_counter() - t0) / 40
    g.profile(True)
    g.profile_reset()
    for _ in range(5):
        g.cycle(o, f, u)
    rows = g.profile_rows()
    g.profile(False)
    sizes = [H.sizes(l) for l in range(H.num_levels)]
    return dict(host=float(np.median(tq)), wall=wall, rows=rows, sizes=sizes)


emit(f"# tools/mr8_budget.py {('--mesh ' + os.path.basename(a.mesh) + ' --divide ' + str(a.divide)) if a.mesh else '--size ' + str(a.size)} --smoother {a.smoother}{' --push' if a.push else ''}: one rank of N alone on one MI355X, "
     + ("direct-store transport in loop-back form (exchanges without a direct form: RCCL in loop-back mode)" if a.push else "native RCCL back-end in loop-back mode"))
emit("# host = host time to enqueue one cycle incl. the RCCL group calls; wall = GPU time per cycle of back-to-back cycles; us")
for nranks in [int(x) for x in a.ranks.split(",")]:
    for agg in ([16] if nranks == 1 else [int(x) for x in a.agg.split(",")]):
        r0 = run(nranks, 0, agg)
        rl = run(nranks, nranks - 1, agg) if nranks > 1 else r0
        calls = sum(v["calls"] for v in r0["rows"].values()) / 5
        ex = r0["rows"].get("exchange", dict(calls=0, ms=0.0))
        emit(f"N={nranks} TE_AGGLOMERATE={agg}: rank 0 levels (local, global) {r0['sizes']}")
        emit(f"   rank 0: host {r0['host'] * 1e6:7.1f}  wall {r0['wall'] * 1e6:7.1f}   rank {nranks - 1}: host {rl['host'] * 1e6:7.1f}  wall {rl['wall'] * 1e6:7.1f}"
             f"   serial tail on rank 0 {max(r0['wall'] - rl['wall'], 0) * 1e6:6.1f}   launches+exchanges/cycle {calls:.0f}"
             f"   exchanges/cycle {ex['calls'] / 5:.0f} ({ex['ms'] / 5 * 1e3:.1f} us in loop-back)")
        for k, v in sorted(r0["rows"].items(), key=lambda kv: -kv[1]["ms"]):
            emit(f"      {k:40s} calls/cycle {v['calls'] / 5:5.1f}   us/cycle {v['ms'] / 5 * 1e3:8.1f}")
if a.out:
    open(a.out, "w").write("\n".join(lines) + "\n")
