#!/usr/bin/env python3
"""Tooling: the kernel-class table of one te_bicgstab solve (trig problem, 1e-12) per smoother: time, calls, GB/s of algorithmic bytes.
usage: solve_rows.py [--size 512]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from pressurepoissonsolver_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
a = ap.parse_args()
n = 32
H = capi.Hierarchy(capi.Mesh.uniform(3, int(round(np.log2(a.size // n)))), n)
g = capi.GMG(H)
b, x = g.new_vector(0), g.new_vector(0)
g.init_problem(b, None, problem=capi.PROBLEM_TRIG)
for name, sm in (("rbgs", capi.SMOOTH_RBGS), ("patch_solve", capi.SMOOTH_PATCH_SOLVE)):
    o = g.default_opts(smoother=sm)
    x.set(0.0)
    g.bicgstab(x, b, o)
    x.set(0.0)
    g.sync()
    t0 = time.perf_counter()
    its, rr = g.bicgstab(x, b, o)
    g.sync()
    dt = time.perf_counter() - t0
    x.set(0.0)
    g.profile(True)
    g.profile_reset()
    g.bicgstab(x, b, o)
    rows = g.profile_rows()
    g.profile(False)
    print(f"== {name}: {its} iterations, {dt * 1e3:.2f} ms, rel resid {rr:.2e}")
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
        bts = bench.ALG_BYTES.get(k, 8.0) * v["cells"]
        print(f"   {k:40s} calls {v['calls']:4d}  ms {v['ms']:8.3f}  us/call {v['ms'] / v['calls'] * 1e3:8.1f}  alg {bts / max(v['ms'], 1e-9) / 1e6:8.0f} GB/s")
