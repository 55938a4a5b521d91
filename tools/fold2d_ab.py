import sys, os, time, json
sys.path.insert(0, os.getcwd())
from pressurepoissonsolver_amd import capi
from tests import util
m = util.mesh("uniform", 6, 2)
H = capi.Hierarchy(m, 64)
g = capi.GMG(H)
f = g.new_vector(0); u = g.new_vector(0)
g.init_problem(f, None, capi.PROBLEM_TRIG)
o = g.default_opts(smoother=capi.SMOOTH_RBGS)
for mode in (None, "1", None, "1"):
    g.set_option("TE_2D_NO_FOLD", mode)
    for _ in range(5): g.cycle(o, f, u)
    g.sync(); t = time.perf_counter()
    for _ in range(50): g.cycle(o, f, u)
    g.sync(); dt = (time.perf_counter() - t) / 50
    g.profile(True); g.profile_reset()
    for _ in range(10): g.cycle(o, f, u)
    rows = g.profile_rows(); g.profile(False)
    print("NO_FOLD" if mode else "fold   ", f"{dt*1e6:8.1f} us/cycle", {k: (v["calls"] // 10, round(v["ms"] * 100, 1)) for k, v in rows.items()})
