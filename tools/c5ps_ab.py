"""Tooling: wall time per V-cycle of the reference smoother at 4096^2 (64^2 patches), 50 cycles after 5; argv[1] = another build
of the library (build.build_variant) for same-box A/B runs."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from pressurepoissonsolver_amd import capi
if len(sys.argv) > 1: capi.LIB_PATH = os.path.abspath(sys.argv[1])
from tests import util
H = capi.Hierarchy(util.mesh("uniform", 6, 2), 64)
g = capi.GMG(H)
f, u = g.new_vector(0), g.new_vector(0)
g.init_problem(f, None, capi.PROBLEM_TRIG)
o = g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE)
for _ in range(5): g.cycle(o, f, u)
g.sync(); t = time.perf_counter()
for _ in range(50): g.cycle(o, f, u)
g.sync(); print(sys.argv[1:] or "default", f"{(time.perf_counter() - t) / 50 * 1e6:.1f} us/cycle")
