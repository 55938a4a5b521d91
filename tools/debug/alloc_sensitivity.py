"""debug: do the level-0 kernel times depend on where the vectors happen to be allocated? several solver instances in one
process, dummy allocations in between to move the addresses"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pressurepoissonsolver_amd import capi

mesh = capi.Mesh.uniform(3, 4)
H = capi.Hierarchy(mesh, 32)
keep = []
rng = np.random.default_rng(1)
for trial in range(10):
    pad = 0 if trial == 0 else int(rng.integers(1, 4000)) * 4096 + int(rng.integers(0, 16)) * 256
    keep.append(torch.empty(pad, dtype=torch.uint8, device="cuda"))
    f0 = None
    g = capi.GMG(H)
    f, u = g.new_vector(0), g.new_vector(0)
    g.init_problem(f, None, problem=capi.PROBLEM_RANDOM)
    o = g.default_opts(smoother=capi.SMOOTH_RBGS)
    for _ in range(3):
        g.cycle(o, f, u)
    g.sync()
    res = []
    for rep in range(3):
        g.profile(True); g.profile_reset()
        for _ in range(5):
            g.cycle(o, f, u)
        rows = g.profile_rows(); g.profile(False)
        res.append((rows["rbgs_zero_resid_restrict_faces"]["ms"] / 5 * 1e3, rows["rbgs_resweep_prolong"]["ms"] / 5 * 1e3))
    pf = capi.lib().te_vec_device_ptr(f.h); pu = capi.lib().te_vec_device_ptr(u.h)
    print(f"trial {trial} pad {pad:9d} f {pf:#x} u {pu:#x}  zero_resid/resweep us: " + "  ".join(f"{a:.1f}/{b:.1f}" for a, b in res), flush=True)
    del f, u, g
