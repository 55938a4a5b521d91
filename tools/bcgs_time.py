"""V-cycle at 4096^2 (config C5: 4096 patches of 64^2) with each 2D smoother; kernel-class times of the patch solvers and the
iteration counts of TE_SMOOTH_PATCH_BCGS (PatchSolvers/BiCGStabSolver.h). python tools/bcgs_time.py on a GPU box."""
import time, json, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from pressurepoissonsolver_amd import capi, problems
from tests import util
m = util.mesh("uniform", 6, 2)
H = capi.Hierarchy(m, 64)
g = capi.GMG(H)
P = len(H.tables(0)["id"])
f = g.new_vector(0); u = g.new_vector(0)
g.init_problem(f, None, capi.PROBLEM_TRIG)
out = {}
for sm, name in ((0, "patch_solve"), (3, "patch_bcgs"), (2, "rbgs")):
    o = g.default_opts(smoother=sm)
    for _ in range(2): g.cycle(o, f, u)
    g.sync()
    g.profile(True); g.profile_reset()
    t = time.time()
    for _ in range(5): g.cycle(o, f, u)
    g.sync()
    dt = (time.time() - t) / 5
    rows = g.profile_rows(); g.profile(False)
    out[name] = dict(ms_per_cycle=dt * 1e3, rows={k: v for k, v in rows.items() if k in ("patch_bcgs", "patch_rhs", "patch_solve_mfma")})
    if sm == 3:
        out[name]["its"] = [int(x) for x in np.percentile(g.patch_bcgs_iterations(0, P), [0, 50, 100])]
print(json.dumps(out, indent=1, default=str))
