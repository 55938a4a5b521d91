"""Tooling: four RB-GS solves of the trig problem on 512^3 in one process, wall ms each; argv[1] = another build of the library
(pressurepoissonsolver_amd.build.build_variant) for same-box A/B runs. Process-to-process scatter on one box is +-1.3 ms."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from pressurepoissonsolver_amd import capi
if len(sys.argv) > 1: capi.LIB_PATH = os.path.abspath(sys.argv[1])
H = capi.Hierarchy(capi.Mesh.uniform(3, 4), 32)
g = capi.GMG(H)
f, x = g.new_vector(0), g.new_vector(0)
g.init_problem(f, None, problem=capi.PROBLEM_TRIG)
o = g.default_opts(smoother=capi.SMOOTH_RBGS)
ts = []
for rep in range(4):
    x.set(0.0); g.sync(); t0 = time.perf_counter()
    its, rr = g.bicgstab(x, f, o, 200, 1e-12)
    g.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print(sys.argv[1:] or "default", its, " ".join(f"{t:.2f}" for t in ts))
