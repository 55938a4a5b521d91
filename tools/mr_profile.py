"""Tooling: launch / exchange counts per V-cycle on one rank of an N-rank 512^3 run (virtual ranks on one GPU;
times are NOT representative -- all ranks share the device -- only the counts and kernel classes are)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from pressurepoissonsolver_amd import capi, problems, dist as tedist

nr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sm = {"rbgs": capi.SMOOTH_RBGS, "patch_solve": capi.SMOOTH_PATCH_SOLVE}[sys.argv[3] if len(sys.argv) > 3 else "rbgs"]
dim = int(sys.argv[4]) if len(sys.argv) > 4 else 3  # mr_profile.py 8 4096 rbgs 2 = config C5 on 8 ranks
n = 32 if dim == 3 else 64
div = int(round(np.log2(size // n)))
fab = tedist.LocalFabric(nr)
out = {}

def body(rank):
    mesh = capi.Mesh.uniform(dim, div)
    H = capi.Hierarchy(mesh, n, rank=rank, nranks=nr)
    g = capi.GMG(H)
    fab.attach(g, rank)
    ids = H.tables(0)["id"][H.l2g(0)]
    f = g.new_vector(0, problems.random_rhs(ids, n ** dim))
    u = g.new_vector(0)
    o = g.default_opts(smoother=sm)
    for _ in range(2):
        g.cycle(o, f, u)
    g.sync(); fab.barrier.wait()
    g.profile(True); g.profile_reset()
    t0 = time.time()
    for _ in range(5):
        g.cycle(o, f, u)
    g.sync(); fab.barrier.wait()
    out[rank] = (time.time() - t0, g.profile_rows(), [H.sizes(l) for l in range(H.num_levels)])

fab.run(body)
dt, rows, sizes = out[0]
print("rank 0 of", nr, "levels (local, global):", sizes, " wall %.2f ms/cycle (shared GPU)" % (dt / 5 * 1e3))
tot = 0
for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-24s calls/cycle %5.1f  ms/cycle %.4f" % (k, v["calls"] / 5, v["ms"] / 5)); tot += v["calls"] / 5
print("  launches+exchanges per cycle:", tot)
