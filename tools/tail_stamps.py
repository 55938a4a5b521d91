#!/usr/bin/env python3
"""Tooling: where the time of the small-level launches of a V-cycle goes (the levels with <= 64 patches: z-slab sweeps, z-slab
stencils, the three passes of the coarsest exact solve). Runs a diagnostic build of the library (-DTE_STAMPS=1: wave 0 of every
workgroup of those kernels keeps s_memrealtime stamps of the links of its dependent chain, kernels3d.hpp Stamps) and prints,
per launch of one cycle, the gap to the previous launch and the median over workgroups of each link.
   usage: tail_stamps.py [--size 512] [--smoother rbgs] [--cycles 5]       (build the variant first, in the build container:
          python -c "from pressurepoissonsolver_amd import build; build.build_variant('stamps', ['-DTE_STAMPS=1'])")"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from pressurepoissonsolver_amd import capi, problems  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--dim", type=int, default=3)
ap.add_argument("--patch", type=int, default=32)
ap.add_argument("--smoother", default="rbgs")
ap.add_argument("--cycles", type=int, default=5)
ap.add_argument("--lib", default=os.path.join(os.path.dirname(capi.LIB_PATH), "libte_hip_stamps.so"))
ap.add_argument("options", nargs="*", help="TE_* switches as K=V")
a = ap.parse_args()
capi.LIB_PATH = os.path.abspath(a.lib)
L = capi.lib()
L.te_stamps_begin.restype = C.c_int
L.te_stamps_begin.argtypes = [C.c_void_p]
L.te_stamps_read.restype = C.c_int
L.te_stamps_read.argtypes = [C.c_void_p] * 4 + [C.c_int]
MAXL, MAXWG, NST = 64, 1024, 8

n = a.patch
mesh = capi.Mesh.uniform(a.dim, int(round(np.log2(a.size // n))))
H = capi.Hierarchy(mesh, n)
g = capi.GMG(H)
for kv in a.options:
    k, v = kv.split("=")
    g.set_option(k, v)
sm = {"rbgs": capi.SMOOTH_RBGS, "jacobi": capi.SMOOTH_JACOBI, "patch_solve": capi.SMOOTH_PATCH_SOLVE}[a.smoother]
opts = g.default_opts(smoother=sm)
f = g.new_vector(0, problems.random_rhs(H.tables(0)["id"], n ** a.dim))
u = g.new_vector(0)
for _ in range(5):
    g.cycle(opts, f, u)
g.sync()
print(f"# tools/tail_stamps.py --size {a.size} --smoother {a.smoother} {' '.join(a.options)}: levels {[H.sizes(l)[1] for l in range(H.num_levels)]} patches of {n}^{a.dim}")
print("# per launch, us: gap = first workgroup's entry - last workgroup's end of the previous instrumented launch (other launches may lie between);")
print("# span = first entry .. last end; then medians over the launch's workgroups of the links of one workgroup's chain:")
print("# skew = entry after the launch's first entry | tables = entry -> per-patch tables there | data = -> the prologue's requests there |")
print("# march = -> last result formed (s3, s4: intermediate points of it, from the same origin: slab kernels: first step done / main loop done;")
print("# ps xy: x products done) | drain = -> stores retired | wg = entry -> end")
acc = {}
for c in range(a.cycles):
    capi.check(L.te_stamps_begin(g.h))
    g.cycle(opts, f, u)
    out = np.zeros((MAXL, MAXWG, NST), np.uint64)
    names = C.create_string_buffer(64 * MAXL)
    wgs = np.zeros(MAXL, np.int32)
    nl = L.te_stamps_read(g.h, out.ctypes.data_as(C.c_void_p), C.cast(names, C.c_void_p), wgs.ctypes.data_as(C.c_void_p), MAXL)
    if nl < 0:
        capi.check(nl)
    prev_end = None
    for i in range(nl):
        name = names.raw[64 * i:64 * (i + 1)].split(b"\0")[0].decode()
        t = out[i, :wgs[i]].astype(np.int64)
        t = t[t[:, 0] > 0]  # (workgroups past the end of the work list return at once)
        if len(t) == 0:
            continue
        us = lambda x: x * 0.01  # noqa: E731  (100 MHz)
        e0, end = t[:, 0].min(), t[:, 6].max()
        row = dict(wgs=len(t), gap=us(e0 - prev_end) if prev_end is not None else float("nan"), span=us(end - e0),
                   skew=us(np.median(t[:, 0] - e0)), tables=us(np.median(t[:, 1] - t[:, 0])), data=us(np.median(t[:, 2] - t[:, 1])),
                   s3=us(np.median(t[:, 3] - t[:, 2])) if (t[:, 3] > 0).all() else float("nan"),
                   s4=us(np.median(t[:, 4] - t[:, 2])) if (t[:, 4] > 0).all() else float("nan"),
                   march=us(np.median(t[:, 5] - t[:, 2])), drain=us(np.median(t[:, 6] - t[:, 5])), wg=us(np.median(t[:, 6] - t[:, 0])))
        prev_end = end
        acc.setdefault((i, name), []).append(row)
tot_span = tot_gap = 0.0
print(f"{'launch':34s} {'wgs':>5s} {'gap':>6s} {'span':>6s} | {'skew':>6s} {'tables':>6s} {'data':>6s} {'s3':>6s} {'s4':>6s} {'march':>6s} {'drain':>6s} {'wg':>6s}")
for (i, name), rows in sorted(acc.items()):
    m = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    print(f"{i:2d} {name:31s} {int(m['wgs']):5d} {m['gap']:6.2f} {m['span']:6.2f} | {m['skew']:6.2f} {m['tables']:6.2f} {m['data']:6.2f} {m['s3']:6.2f} {m['s4']:6.2f} {m['march']:6.2f} {m['drain']:6.2f} {m['wg']:6.2f}")
    tot_span += m["span"]
    if i > 0 and not np.isnan(m["gap"]):
        tot_gap += m["gap"]
print(f"# sum of spans {tot_span:.1f} us, sum of gaps between instrumented launches {tot_gap:.1f} us (these include the launches that are not instrumented)")
