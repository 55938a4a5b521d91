#!/usr/bin/env python3
"""Tooling: V-cycle time and per-kernel times under different environment switches, in ONE process (the library
reads its TE_* switches once at creation: the variants are set through te_gmg_set_option), with a checksum of the result so that variants that must be bit-identical
can be seen to be.   usage: variant_bench.py [--size 512] [--smoother rbgs] [--steps 20] "A=1,B=2" "A=3" ...
("" = no switch)."""
import argparse
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from pressurepoissonsolver_amd import capi, problems  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--dim", type=int, default=3)
ap.add_argument("--patch", type=int, default=32)
ap.add_argument("--mesh", default=None)
ap.add_argument("--divide", type=int, default=0)
ap.add_argument("--smoother", default="rbgs")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--lib", default=None, help="another build of the library (pressurepoissonsolver_amd.build.build_variant) instead of libte_hip.so")
ap.add_argument("variants", nargs="*", default=[""])
a = ap.parse_args()
if a.lib:
    capi.LIB_PATH = os.path.abspath(a.lib)
n = a.patch
if a.mesh:
    mesh = capi.Mesh.read(a.mesh, a.dim)
    for _ in range(a.divide):
        mesh.refine_leaves()
else:
    mesh = capi.Mesh.uniform(a.dim, int(round(np.log2(a.size // n))))
H = capi.Hierarchy(mesh, n)
g = capi.GMG(H)
sm = {"rbgs": capi.SMOOTH_RBGS, "jacobi": capi.SMOOTH_JACOBI, "patch_solve": capi.SMOOTH_PATCH_SOLVE}[a.smoother]
opts = g.default_opts(smoother=sm)
f = g.new_vector(0, problems.random_rhs(H.tables(0)["id"], n ** a.dim))
u = g.new_vector(0)
cells = H.cells(0)
for var in a.variants:
    keys = []
    for kv in filter(None, var.split(",")):
        k, v = kv.split("=")
        g.set_option(k, v)
        keys.append(k)
    for _ in range(3):
        g.cycle(opts, f, u)
    g.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.cycle(opts, f, u)
    g.sync()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    g.profile(True)
    g.profile_reset()
    for _ in range(5):
        g.cycle(opts, f, u)
    rows = g.profile_rows()
    g.profile(False)
    digest = hashlib.sha1(u.download().tobytes()).hexdigest()[:12]
    top = sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("VB_TOP", "6"))]
    print(f"[{var or 'default':28s}] {ms:.4f} ms/cycle  {cells / ms / 1e6:.1f} G updates/s  sha {digest}  | "
          + "  ".join(f"{k} {v['ms'] / v['calls'] * 1e3:.1f}us x{v['calls'] // 5}" for k, v in top), flush=True)
    for k in keys:
        g.set_option(k, None)
