"""Tooling: host time to enqueue one V-cycle against its wall time on the GPU (512^3, 256^3, 4096^2 in 2D): the stream must stay queued ahead."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pressurepoissonsolver_amd import capi, problems
for dim, n, div in ((3, 32, 4), (3, 32, 3), (2, 64, 6)):
    mesh = capi.Mesh.uniform(dim, div)
    H = capi.Hierarchy(mesh, n)
    g = capi.GMG(H)
    f = g.new_vector(0); u = g.new_vector(0)
    g.init_problem(f, None, problem=capi.PROBLEM_RANDOM)
    o = g.default_opts(smoother=capi.SMOOTH_RBGS)
    for _ in range(5): g.cycle(o, f, u)
    g.sync()
    # host time to ENQUEUE a cycle (no sync inside): the queue is kept short by syncing every cycle before timing the next
    tq = []
    for _ in range(50):
        g.sync(); t0 = time.perf_counter(); g.cycle(o, f, u); tq.append(time.perf_counter() - t0)
    g.sync(); t0 = time.perf_counter()
    for _ in range(50): g.cycle(o, f, u)
    g.sync(); wall = (time.perf_counter() - t0) / 50
    print(f"dim {dim} n {n} div {div}: host enqueue {np.median(tq)*1e6:.0f} us per cycle, wall {wall*1e6:.0f} us per cycle")
