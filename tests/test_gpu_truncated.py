"""-m gpu: TRUNCATED hierarchies, i.e. what the reference does on N ranks instead of gathering small levels: the level loop of
CycleFactory3d.cpp:98-127 stops after `max_levels` levels or at the first level with fewer than `patches_per_proc` patches per
rank (GMG/CycleOpts.h:55-63; :101-104). The coarsest level then has many patches and is "solved" by `coarse_sweeps` sweeps of
the smoother like any other level. te_hier_build takes both numbers; here the cycles on such hierarchies run on the GPU
against the CPU oracle's cycle over the SAME truncated level list (tolerance 1e-10 relative 2-norm, as every cycle test), V and
W, both smoothers, on 256^3 (config C2's grid) and on the refined tree of config C4 -- and sharded over 4 virtual ranks they
equal the single-rank run bit for bit."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from pressurepoissonsolver_amd import capi, dist as tedist, problems
from tests import util

pytestmark = pytest.mark.gpu

# (mesh, divides, n, max_levels, patches_per_proc, nranks, expected patches per level)
CASES = {
    "256^3-max_levels=2": ("uniform", 3, 32, 2, 0.0, 1, [512, 64]),
    "256^3-max_levels=3": ("uniform", 3, 32, 3, 0.0, 1, [512, 64, 8]),
    "256^3-patches_per_proc=16-4ranks": ("uniform", 3, 32, 0, 16.0, 4, [512, 64]),  # 8 patches / 4 ranks < 16: the loop stops at 64
    "2refine-div2-max_levels=2": ("2refine.bin", 2, 32, 2, 0.0, 1, [960, 512]),
    "2refine-div2-max_levels=3": ("2refine.bin", 2, 32, 3, 0.0, 1, [960, 512, 64]),
    "2refine-div2-patches_per_proc=16-4ranks": ("2refine.bin", 2, 32, 0, 16.0, 4, [960, 512, 64]),
}


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("cycle_type", [0, 1], ids=["V", "W"])
@pytest.mark.parametrize("smoother", [capi.SMOOTH_RBGS, capi.SMOOTH_PATCH_SOLVE], ids=["rbgs", "patch_solve"])
@pytest.mark.parametrize("case", list(CASES), ids=list(CASES))
def test_cycle_on_a_truncated_hierarchy(case, smoother, cycle_type, monkeypatch):
    for k in ("TE_AGGLOMERATE", "TE_AGGLOMERATE_MAX", "TE_REPLICATE", "TE_OVERLAP_MIN"):
        monkeypatch.delenv(k, raising=False)
    name, div, n, max_levels, ppp, nranks, sizes = CASES[case]
    mesh = util.mesh(name, div, 3)
    nc = n ** 3
    orc.set_threads(min(os.cpu_count() or 1, 16))
    # the level list every rank agrees on: built for `nranks` (patches_per_proc counts patches per rank), computed by one
    H1 = capi.Hierarchy(mesh, n, max_levels=max_levels, patches_per_proc=ppp * nranks)  # (one rank: the same cut as nranks ranks with ppp)
    assert [H1.sizes(l)[1] for l in range(H1.num_levels)] == sizes
    levels = orc.levels_from_hierarchy(H1)
    f = problems.random_rhs(H1.tables(0)["id"], nc)
    g1 = capi.GMG(H1)
    o = g1.default_opts(smoother=smoother, cycle_type=cycle_type)
    df, du = g1.new_vector(0, f), g1.new_vector(0)
    g1.cycle(o, df, du)
    got1 = du.download()
    want = orc.cycle(levels, orc.cycle_opts(smoother=smoother, cycle_type=cycle_type), f)
    assert rel(got1, want) <= 1e-10, rel(got1, want)
    # the unfused sequence (one kernel per reference call) on the same hierarchy
    g1.cycle(g1.default_opts(smoother=smoother, cycle_type=cycle_type, fuse=0), df, du)
    assert rel(du.download(), want) <= 1e-10
    del df, du, g1
    if nranks == 1:
        return
    # sharded: every rank builds the same truncated list (te_hier_build's own rule with `patches_per_proc` per rank)
    fab = tedist.LocalFabric(nranks)
    fab.timeout = 600.0
    hs = [capi.Hierarchy(mesh, n, max_levels=max_levels, patches_per_proc=ppp, rank=r, nranks=nranks) for r in range(nranks)]
    assert [hs[0].sizes(l)[1] for l in range(hs[0].num_levels)] == sizes
    gs = [capi.GMG(h) for h in hs]
    for r, g in enumerate(gs):
        fab.attach(g, r)

    def per_rank(r):
        H, g = hs[r], gs[r]
        idx = H.l2g(0)
        d_f, d_u = g.new_vector(0, f.reshape(-1, nc)[idx].ravel()), g.new_vector(0)
        g.cycle(g.default_opts(smoother=smoother, cycle_type=cycle_type), d_f, d_u)
        return idx, d_u.download()

    outs = fab.run(per_rank)
    P = sizes[0]
    got = np.zeros(P * nc)
    for idx, u in outs:
        got.reshape(P, nc)[idx] = u.reshape(len(idx), nc)
    assert np.array_equal(got, got1), np.abs(got - got1).max()
