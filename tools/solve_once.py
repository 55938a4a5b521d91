"""Tooling: ONE te_bicgstab + V-cycle solve of the trig problem on 512^3 (RB-GS), for counter passes over the solve's own kernels
(tools/solve_traffic.sh). argv[1] = an option to set (e.g. TE_NO_BICG_XF) or nothing."""
import os
import sys
sys.path.insert(0, os.getcwd())
from pressurepoissonsolver_amd import capi
H = capi.Hierarchy(capi.Mesh.uniform(3, 4), 32)
g = capi.GMG(H)
if len(sys.argv) > 1:
    g.set_option(sys.argv[1], "1")
f, x = g.new_vector(0), g.new_vector(0)
g.init_problem(f, None, problem=capi.PROBLEM_TRIG)
its, rr = g.bicgstab(x, f, g.default_opts(), 200, 1e-12)
g.sync()
print(its, rr)
