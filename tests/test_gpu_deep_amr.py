"""-m gpu: deep adaptively refined trees -- apps/3d/meshes/multi_refine.bin (137 nodes, 5 levels), multi_refine_8.bin (321
nodes, 9 levels) and apps/2d/meshes/multi_refine_8.bin (213 nodes, 9 levels), copied as data fixtures -- at n = 16, the
patch size of apps/3d/config/gmg_example.ini (num_cells=16, neumann=true, problem=gauss, mesh=multi_refine_8.bin,
prec=GMG, tolerance 1e-12, BiCGStab). Copy-through chains over many levels, coarse/fine faces on every level.

The oracle's level tables here do NOT come from the product: they are built from the tree's node table by
oracle/levels_bfs.py, the breadth-first walk of ThundereggDomGen.h:127-222 restated in Python (patch order per level taken
from the product, which is free to choose it)."""
import numpy as np
import pytest

from oracle import levels_bfs
from oracle import oracle as orc
from pressurepoissonsolver_amd import capi, dist as tedist, problems
from tests import util

pytestmark = pytest.mark.gpu

MESHES = [("multi_refine.bin", 3), ("multi_refine_8.bin", 3), ("2d_multi_refine_8.bin", 2)]
N = 16


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def setup(name, dim, neumann=False):
    m = util.mesh(name, 0, dim)
    H = capi.Hierarchy(m, N, neumann=neumann)
    nodes = m.nodes()
    tabs = levels_bfs.tables_in_order(levels_bfs.extract_levels(nodes, dim), nodes, dim,
                                      [H.tables(l)["id"] for l in range(H.num_levels)])
    levels = [orc.Level.from_tables(t, dim, N, neumann) for t in tabs]
    return m, H, levels


@pytest.mark.parametrize("name,dim", MESHES)
@pytest.mark.parametrize("smoother", [capi.SMOOTH_PATCH_SOLVE, capi.SMOOTH_RBGS], ids=["patch_solve", "rbgs"])
def test_vcycle_on_deep_tree_equals_oracle(name, dim, smoother):
    m, H, levels = setup(name, dim)
    g = capi.GMG(H)
    f = util.rand_vec(levels[0].size, 90) / levels[0].a["h"].min() ** 2
    want = orc.cycle(levels, orc.cycle_opts(smoother=smoother), f)
    got = {}
    for fuse in (0, 3):
        df, du = g.new_vector(0, f), g.new_vector(0)
        g.cycle(g.default_opts(smoother=smoother, fuse=fuse), df, du)
        got[fuse] = du.download()
        assert rel(got[fuse], want) <= 1e-10, (fuse, rel(got[fuse], want))
    # the operators level by level on these tables as well (coarse/fine faces everywhere)
    for l in range(H.num_levels):
        u = util.rand_vec(levels[l].size, 91 + l)
        du, dr = g.new_vector(l, u), g.new_vector(l)
        g.apply(du, dr, level=l)
        assert np.abs(dr.download() - orc.apply(levels[l], u)).max() <= util.op_tol(levels[l], u)


def test_gmg_example_config_neumann_gauss_bicgstab():
    """apps/3d/config/gmg_example.ini as the driver runs it (apps/3d/steady.cpp:318-334, 519-549): Init::initNeumann of the
    gauss problem, f -= integrate(f)/volume, BiCGStab preconditioned by the V-cycle to 1e-12, solution compared after
    removing the constant; against the oracle on independently extracted level tables."""
    m, H, levels = setup("multi_refine_8.bin", 3, neumann=True)
    g = capi.GMG(H)
    t = H.tables(0)
    f, exact = problems.init_neumann(t, N, problem="gauss")
    cell = np.prod(t["lengths"] / N, axis=1)
    df, de = g.new_vector(0, f), g.new_vector(0, exact)
    vol = g.volume()
    df.shift(-g.integrate(df) / vol)
    fz = df.download()
    for sm in (capi.SMOOTH_PATCH_SOLVE, capi.SMOOTH_RBGS):
        dx = g.new_vector(0)
        its, rr = g.bicgstab(dx, df, g.default_opts(smoother=sm), tol=1e-12)
        assert rr <= 1e-12 and its <= 60, (its, rr)
        x_ref, its_ref, rr_ref = orc.bicgstab(levels, orc.cycle_opts(smoother=sm), fz, tol=1e-12)
        assert abs(its - its_ref) <= 2, (its, its_ref)
        uavg = g.integrate(dx) / vol
        x = dx.download()
        xr = x_ref - np.sum(x_ref.reshape(len(cell), -1).sum(axis=1) * cell) / vol
        assert rel(x - uavg, xr) <= 1e-6
        eavg = g.integrate(de) / vol
        err = rel(x - uavg, exact - eavg)
        assert err <= 0.05, err  # the gauss problem on this mesh at n = 16: a few per cent at most (second order)


@pytest.mark.parametrize("name,dim", MESHES)
def test_deep_tree_four_virtual_ranks_equal_one(name, dim, monkeypatch):
    monkeypatch.setenv("TE_OVERLAP_MIN", "0")
    mesh = util.mesh(name, 0, dim)
    H1 = capi.Hierarchy(mesh, N)
    g1 = capi.GMG(H1)
    nc = N ** dim
    P = H1.sizes(0)[1]
    f = util.rand_vec(P * nc, 95)
    u0 = util.rand_vec(P * nc, 96)
    want = {}
    for sm in (capi.SMOOTH_RBGS, capi.SMOOTH_PATCH_SOLVE):
        df, du = g1.new_vector(0, f), g1.new_vector(0)
        g1.cycle(g1.default_opts(smoother=sm), df, du)
        want[sm] = du.download()
    du, dr = g1.new_vector(0, u0), g1.new_vector(0)
    g1.apply(du, dr)
    want["apply"] = dr.download()

    nranks = 4
    fab = tedist.LocalFabric(nranks)
    hs = [capi.Hierarchy(mesh, N, rank=r, nranks=nranks) for r in range(nranks)]
    gs = [capi.GMG(h) for h in hs]
    for r, g in enumerate(gs):
        fab.attach(g, r)

    def per_rank(r):
        H, g = hs[r], gs[r]
        idx = H.l2g(0)
        out = {}
        for sm in (capi.SMOOTH_RBGS, capi.SMOOTH_PATCH_SOLVE):
            df, du = g.new_vector(0, f.reshape(-1, nc)[idx].ravel()), g.new_vector(0)
            g.cycle(g.default_opts(smoother=sm), df, du)
            out[sm] = du.download()
        du, dr = g.new_vector(0, u0.reshape(-1, nc)[idx].ravel()), g.new_vector(0)
        g.apply(du, dr)
        out["apply"] = dr.download()
        return idx, out

    outs = fab.run(per_rank)
    for k in want:
        got = np.zeros(P * nc)
        for idx, o in outs:
            got.reshape(P, nc)[idx] = o[k].reshape(len(idx), nc)
        assert np.array_equal(got, want[k]), (k, np.abs(got - want[k]).max())


@pytest.mark.parametrize("kw", [dict(cycle_type=1), dict(smoother=capi.SMOOTH_JACOBI, omega=0.8), dict(pre_sweeps=2, post_sweeps=0, coarse_sweeps=3)],
                         ids=["W-cycle", "jacobi", "V(2,0) coarse 3"])
def test_other_cycle_shapes_on_deep_tree(kw):
    """W-cycle (WCycle.h:45-68), the weighted Jacobi smoother and other sweep counts over the nine levels of
    multi_refine_8.bin, against the oracle on independently extracted tables"""
    m, H, levels = setup("multi_refine_8.bin", 3)
    g = capi.GMG(H)
    f = util.rand_vec(levels[0].size, 97) / levels[0].a["h"].min() ** 2
    names = dict(pre_sweeps="pre", post_sweeps="post", coarse_sweeps="coarse")
    okw = {names.get(k, k): v for k, v in kw.items()}
    okw.setdefault("smoother", capi.SMOOTH_RBGS)
    want = orc.cycle(levels, orc.cycle_opts(**okw), f)
    gkw = dict(kw)
    gkw.setdefault("smoother", capi.SMOOTH_RBGS)
    for fuse in (0, 3):
        df, du = g.new_vector(0, f), g.new_vector(0)
        g.cycle(g.default_opts(fuse=fuse, **gkw), df, du)
        assert rel(du.download(), want) <= 1e-10, (fuse, rel(du.download(), want))
