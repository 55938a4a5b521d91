import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pressurepoissonsolver_amd import capi
if len(sys.argv) > 2: capi.LIB_PATH = os.path.abspath(sys.argv[2])
size = int(sys.argv[1])
H = capi.Hierarchy(capi.Mesh.uniform(3, int(round(np.log2(size // 32)))), 32)
g = capi.GMG(H)
b, x = g.new_vector(0), g.new_vector(0)
g.init_problem(b, None, problem=capi.PROBLEM_TRIG)
o = g.default_opts(smoother=capi.SMOOTH_RBGS)
ts = []
for k in range(4):
    x.set(0.0); g.sync(); t0 = time.perf_counter(); its, rr = g.bicgstab(x, b, o); g.sync(); ts.append((time.perf_counter() - t0) * 1e3)
f = g.new_vector(0); g.init_problem(f, None, problem=capi.PROBLEM_RANDOM); u = g.new_vector(0)
for _ in range(3): g.cycle(o, f, u)
g.sync(); t0 = time.perf_counter()
for _ in range(20): g.cycle(o, f, u)
g.sync(); cyc = (time.perf_counter() - t0) / 20 * 1e3
print(f"size {size} lib {os.path.basename(capi.LIB_PATH)}: solve {its} its, ms {[round(t, 2) for t in ts]}, cycle {cyc:.4f} ms")
