"""debug: sharded vs single-rank on the refined tree, which options/patches differ"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pressurepoissonsolver_amd import capi, dist as tedist, problems
from tests import util

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
div = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mesh = util.mesh("2refine.bin", div, 3)
nc = n ** 3
H1 = capi.Hierarchy(mesh, n)
P = H1.sizes(0)[1]
print("levels", [H1.sizes(l) for l in range(H1.num_levels)], flush=True)
f = problems.random_rhs(H1.tables(0)["id"], nc)
for nranks, omin, agg in itertools.product((2, 4), ("0", "128"), ("0", "16")):
    os.environ["TE_OVERLAP_MIN"] = omin
    os.environ["TE_AGGLOMERATE"] = agg
    g1 = capi.GMG(H1)
    want = {}
    for fuse in (1, 2, 3):
        df, du = g1.new_vector(0, f), g1.new_vector(0)
        g1.cycle(g1.default_opts(smoother=capi.SMOOTH_RBGS, fuse=fuse), df, du)
        want[fuse] = du.download()
    fab = tedist.LocalFabric(nranks)
    hs = [capi.Hierarchy(mesh, n, rank=r, nranks=nranks) for r in range(nranks)]
    gs = [capi.GMG(h) for h in hs]
    for r, g in enumerate(gs):
        fab.attach(g, r)
    def per_rank(r):
        H, g = hs[r], gs[r]
        idx = H.l2g(0)
        out = {}
        for fuse in (1, 2, 3):
            df, du = g.new_vector(0, f.reshape(-1, nc)[idx].ravel()), g.new_vector(0)
            g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS, fuse=fuse), df, du)
            out[fuse] = du.download()
        return idx, out
    outs = fab.run(per_rank)
    for fuse in (1, 2, 3):
        got = np.zeros(P * nc)
        for idx, o in outs:
            got.reshape(P, nc)[idx] = o[fuse].reshape(len(idx), nc)
        d = np.abs(got - want[fuse]).reshape(P, nc).max(axis=1)
        bad = np.nonzero(d > 0)[0]
        print(f"nranks {nranks} overlap_min {omin} agg {agg} fuse {fuse}: {'OK' if len(bad) == 0 else 'DIFF'} bad patches {len(bad)} max {d.max():.3e} (|u| {np.abs(want[fuse]).max():.3e})", flush=True)
        if len(bad):
            t = H1.tables(0)
            print("   first bad:", [(int(p), int(t['rank'][p]) if False else None, float(t['lengths'][p, 0])) for p in bad[:6]])
            tr = hs[0].tables(0)
            print("   ranks of bad:", sorted(set(int(tr['rank'][p]) for p in bad)), " sizes:", sorted(set(float(t['lengths'][p,0]) for p in bad)))
    del gs, hs, g1
