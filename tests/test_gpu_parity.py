"""-m gpu: HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp64):
  operator-level (apply, residual, Jacobi, RB-GS): |x_gpu - x_cpu|_inf <= 32 eps * (4 dim / h^2) |u|_inf
     — a backward-error bound: the two sides round (l - 2m + u)/h^2 differently (division vs
     reciprocal multiply, FMA contraction), so the error scales with the stencil's terms, not its sum.
  restrict: bit-exact (same summation order, exact /8).  prolong-add: bit-exact.
  patch solve / block-Jacobi sweep: 1e-11 relative 2-norm (dense transform rounding ~ n eps per axis).
  V-cycle: 1e-10 relative 2-norm.  BiCGStab: same iteration count +-1 and ||u_gpu-u_cpu||/||u|| <= 1e-8.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from pressurepoissonsolver_amd import capi, problems
from tests import util

pytestmark = pytest.mark.gpu

# (mesh, n, divides, neumann): uniform and refined (coarse/fine faces), single patch, every patch size the
# kernels are instantiated for, Dirichlet and Neumann physical boundaries (StarPatchOp.h:49-65)
CASES = [("2uni.bin", 8, 0, False), ("2refine.bin", 8, 0, False), ("3uni.bin", 4, 0, False), ("2uni.bin", 16, 1, False),
         ("2refine.bin", 16, 1, False), ("uniform", 32, 1, False), ("1uni.bin", 8, 0, False),
         ("2uni.bin", 8, 0, True), ("2refine.bin", 8, 0, True), ("1uni.bin", 16, 0, True),
         ("uniform", 32, 1, True), ("2refine.bin", 32, 0, False),  # 32^3: the matrix-core patch solve, every transform type
         # 2D twins (configs C1: one 256^2 patch; C5-style: 64^2 patches; refined quadtree; Neumann)
         ("2d2uni.bin", 8, 0, False), ("2d2ref.bin", 8, 0, False), ("2d2ref.bin", 16, 1, False), ("2d2ref.bin", 8, 0, True),
         ("uniform2d", 256, 0, False), ("uniform2d", 64, 2, False),
         ("uniform2d", 64, 2, True)]  # 64^2 Neumann: every transform type and the null mode through k_patch_solve2d_mfma


@pytest.fixture(scope="module", params=CASES, ids=lambda c: f"{c[0]}-n{c[1]}-d{c[2]}{'-neumann' if c[3] else ''}")
def case(request):
    name, n, div, neu = request.param
    dim = 2 if name.startswith(("2d", "uniform2d")) else 3
    m, H, levels = util.setup("uniform" if name == "uniform2d" else name, n, div, neumann=neu, dim=dim)
    g = capi.GMG(H)
    return dict(H=H, levels=levels, g=g, n=n, neumann=neu, dim=dim)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_apply_and_residual(case):
    g, L = case["g"], case["levels"][0]
    u = util.rand_vec(L.size, 1)
    f = util.rand_vec(L.size, 2)
    du, df, dr = g.new_vector(0, u), g.new_vector(0, f), g.new_vector(0)
    g.apply(du, dr)
    ref = orc.apply(L, u)
    tol = util.op_tol(L, u)
    assert np.abs(dr.download() - ref).max() <= tol
    g.residual(du, df, dr)
    assert np.abs(dr.download() - (f - ref)).max() <= tol + 4 * util.EPS


def test_apply_all_levels(case):
    g = case["g"]
    for l, L in enumerate(case["levels"]):
        u = util.rand_vec(L.size, 10 + l)
        du, dr = g.new_vector(l, u), g.new_vector(l)
        g.apply(du, dr, level=l)
        assert np.abs(dr.download() - orc.apply(L, u)).max() <= util.op_tol(L, u)


def test_jacobi_and_rbgs(case):
    g, L = case["g"], case["levels"][0]
    u = util.rand_vec(L.size, 3)
    h2 = L.a["h"].min() ** 2
    f = util.rand_vec(L.size, 4) / h2
    for sm, ref in ((capi.SMOOTH_JACOBI, orc.jacobi(L, f, u, 0.8)), (capi.SMOOTH_RBGS, orc.patch_rbgs(L, f, u))):
        du, df = g.new_vector(0, u), g.new_vector(0, f)
        g.smooth(df, du, smoother=sm, omega=0.8)
        got = du.download()
        # the update divides by the diagonal ~ 2 dim / h^2: error ~ ulps of |u| + |f| h^2
        assert np.abs(got - ref).max() <= 256 * util.EPS * (np.abs(u).max() + np.abs(f).max() * h2)


def test_restrict_prolong_bit_exact(case):
    g, levels = case["g"], case["levels"]
    for l in range(len(levels) - 1):
        fv = util.rand_vec(levels[l].size, 20 + l)
        cv = util.rand_vec(levels[l + 1].size, 30 + l)
        dfv, dcv = g.new_vector(l, fv), g.new_vector(l + 1)
        g.restrict(dcv, dfv, fine_level=l)
        assert np.array_equal(dcv.download(), orc.restrict(levels[l], levels[l + 1], fv))
        dcv.upload(cv)
        g.interpolate(dcv, dfv, fine_level=l)
        assert np.array_equal(dfv.download(), orc.prolong_add(levels[l], levels[l + 1], cv, fv))


def test_block_jacobi_patch_solve(case):
    g = case["g"]
    for l, L in enumerate(case["levels"]):
        u = util.rand_vec(L.size, 40 + l)
        f = util.rand_vec(L.size, 50 + l) / L.a["h"].min() ** 2
        du, df = g.new_vector(l, u), g.new_vector(l, f)
        g.smooth(df, du, level=l, smoother=capi.SMOOTH_PATCH_SOLVE)
        assert rel(du.download(), orc.smooth(L, f, u)) <= 1e-11


def test_rbgs_z_slabs_bit_identical(case, monkeypatch):
    """Levels with few patches split each patch into z-slabs (one workgroup each, the red values of the two
    adjacent planes recomputed): same bits as the whole-patch sweep, for the plain, zero-guess and
    fused-prolongation variants (the latter two run inside a cycle)."""
    if case["dim"] != 3 or case["n"] < 8:
        pytest.skip("z-slabs exist for 3D patches of 8^3 and larger")
    g, L = case["g"], case["levels"][0]
    u = util.rand_vec(L.size, 42)
    f = util.rand_vec(L.size, 52) / L.a["h"].min() ** 2
    got = {}
    for slabs in (True, False):
        g.set_option("TE_RBGS_NOSLAB", None if slabs else "1")
        du, df, dc = g.new_vector(0, u), g.new_vector(0, f), g.new_vector(0)
        g.smooth(df, du, level=0, smoother=capi.SMOOTH_RBGS)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS, pre_sweeps=2, post_sweeps=2), df, dc)
        got[slabs] = (du.download(), dc.download())
    g.set_option("TE_RBGS_NOSLAB", None)
    assert np.array_equal(got[True][0], got[False][0])
    assert np.array_equal(got[True][1], got[False][1])


def test_patch_solve_variants_agree(case, monkeypatch):
    """The 32^3 patch solve has four implementations of one algorithm: the three-pass MFMA kernels (one
    workgroup per patch, or split over several on small levels), the single-pass kernel k_ps_fused
    (patch resident in registers/LDS, same MFMA sequence -> bit-identical), and the single-pass kernel
    with half-size transforms k_ps_sym (pure DST/DCT axes only; different summation -> a few ulp).
    With and without interface terms (smooth from a random iterate; zero-guess sweep inside a cycle)."""
    if case["n"] != 32 or case["dim"] != 3:
        pytest.skip("the matrix-core patch solve is the 32^3 path")
    g, L = case["g"], case["levels"][0]
    u = util.rand_vec(L.size, 41)
    f = util.rand_vec(L.size, 51) / L.a["h"].min() ** 2
    got = {}
    for mode in ("1pass", "1pass-dense", "3pass", None):
        g.set_option("TE_PS_MODE", mode)
        du, df, dc = g.new_vector(0, u), g.new_vector(0, f), g.new_vector(0)
        g.smooth(df, du, level=0, smoother=capi.SMOOTH_PATCH_SOLVE)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE), df, dc)
        got[mode] = (du.download(), dc.download())
    # (round 6) levels of at most 8 patches run the three passes with one wave per half plane (k_ps_xy_half / k_ps_z_half: the
    # default here); TE_PS_NO_HALF puts the four-wave workgroups back
    g.set_option("TE_PS_MODE", None)
    g.set_option("TE_PS_NO_HALF", "1")
    du, df, dc = g.new_vector(0, u), g.new_vector(0, f), g.new_vector(0)
    g.smooth(df, du, level=0, smoother=capi.SMOOTH_PATCH_SOLVE)
    g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE), df, dc)
    got["nohalf"] = (du.download(), dc.download())
    g.set_option("TE_PS_NO_HALF", None)
    for mode in ("3pass", None, "nohalf"):
        assert np.array_equal(got["1pass-dense"][0], got[mode][0])
        assert np.array_equal(got["1pass-dense"][1], got[mode][1])
    want = orc.smooth(L, f, u)
    assert rel(got["1pass"][0], want) <= 1e-11
    assert rel(got["1pass"][0], got["3pass"][0]) <= 1e-13
    assert rel(got["1pass"][1], got["3pass"][1]) <= 1e-13


@pytest.mark.parametrize("mesh,n,div,neumann", [("uniform", 8, 4, False), ("uniform", 32, 4, True), ("2refine.bin", 8, 3, False), ("2refine.bin", 4, 3, True),
                                                ("2refine.bin", 16, 3, False)])
def test_exported_ghost_terms_gather_variants_bit_identical(mesh, n, div, neumann):
    """Two fused levels in a row: the ghost terms of the restricted residual reach the coarse level through k_fcorr_gather3d_v2
    (round 6: per-level descriptors, 16-byte data loads only), through the kernel it replaces (TE_NO_GTAB2: a table word per entry on
    uniform levels, the walk child -> face -> offset per entry on refined ones) or through the fix-up pass (TE_NO_FCORR): the same
    bits. Uniform (finished sums of the neighbours) and refined (2x2 sums over face layers, copy-through patches) fine levels."""
    m, H, levels = util.setup(mesh, n, div, neumann=neumann, dim=3)
    g = capi.GMG(H)
    f = util.rand_vec(H.cells(0), 77) / levels[0].a["h"].min() ** 2
    got = {}
    for name, opt in (("v2", None), ("v1", "TE_NO_GTAB2"), ("walk", "TE_NO_GTAB"), ("layers", "TE_NO_RS6_CF"), ("fixup", "TE_NO_FCORR"),
                      ("fixup-layers", "TE_NO_FCORR,TE_NO_RS6_FIXUP")):  # (the fix-up pass from exported sums / from the face layers)
        for o1 in (opt.split(",") if opt else ()):
            g.set_option(o1, "1")
        df, du = g.new_vector(0, f), g.new_vector(0)
        g.profile(True)
        g.profile_reset()
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
        rows = g.profile_rows()
        g.profile(False)
        for o1 in (opt.split(",") if opt else ()):
            g.set_option(o1, None)
        assert ("fcorr_gather" in rows) == (not name.startswith("fixup")), (name, sorted(rows))
        got[name] = du.download()
    for name in ("v1", "walk", "layers", "fixup", "fixup-layers"):
        assert np.array_equal(got["v2"], got[name]), name


@pytest.mark.parametrize("n,neumann,mesh,div", [(4, False, "uniform", 3), (8, False, "uniform", 3), (8, True, "uniform", 3),
                                                (16, False, "uniform", 3), (32, False, "uniform", 3),
                                                # refined: patches that copy through, coarse/fine faces (960 patches)
                                                (8, False, "2refine.bin", 2), (8, True, "2refine.bin", 2)])
def test_fused_presweep_residual_restrict(n, neumann, mesh, div):
    """opts.fuse = 2 (default): on uniformly refined levels with >= 256 patches the zero-guess RB-GS pre-sweep, the
    residual and its restriction are one pass (k_rbgs_zero_resid3d) plus a fix-up of the coarse cells along patch
    faces (k_restrict_fixup3d). Not bit-identical to fuse = 1 (the ghost term is added separately): a few ulp."""
    m, H, levels = util.setup(mesh, n, div, neumann=neumann, dim=3)  # uniform: 8^3 = 512 patches on the finest level
    g, L = capi.GMG(H), levels[0]
    f = util.rand_vec(L.size, 54) / L.a["h"].min() ** 2
    got = {}
    for fuse in (1, 2, 3):
        df, dc = g.new_vector(0, f), g.new_vector(0)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS, fuse=fuse), df, dc)
        got[fuse] = dc.download()
    assert not np.array_equal(got[1], got[2])  # the fused path really ran (it is not bit-identical)
    assert rel(got[2], got[1]) <= 1e-13
    # fuse = 3 never stores the iterate between the two sweeps (the post-sweep kernel recomputes it from f): same bits
    assert np.array_equal(got[3], got[2])
    if n <= 16:
        o = orc.cycle_opts(smoother=2)
        assert rel(got[2], orc.cycle(levels, o, f)) <= 1e-10
    # block-Jacobi smoother: the residual after exact patch solves from zero is taken on the face layers only
    outs = []
    for fuse in (1, 2):
        df, dc = g.new_vector(0, f), g.new_vector(0)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE, fuse=fuse), df, dc)
        outs.append(dc.download())
    assert rel(outs[1], outs[0]) <= 1e-11
    if n <= 16:
        assert rel(outs[1], orc.cycle(levels, orc.cycle_opts(smoother=0), f)) <= 1e-10
    # W-cycle and two pre-sweeps fall back to the bit-identical path
    for kw in (dict(cycle_type=1), dict(pre_sweeps=2)):
        outs = []
        for fuse in (1, 2):
            df, dc = g.new_vector(0, f), g.new_vector(0)
            g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS, fuse=fuse, **kw), df, dc)
            outs.append(dc.download())
        if "pre_sweeps" in kw:
            assert np.array_equal(outs[0], outs[1])
        else:
            assert rel(outs[1], outs[0]) <= 1e-13


@pytest.mark.parametrize("n,neumann,mesh,div", [(8, False, "uniform2d", 3), (16, True, "uniform2d", 3), (64, False, "uniform2d", 2),
                                                  (12, True, "uniform2d", 2), (32, False, "uniform2d", 2), (64, True, "uniform2d", 2)])
def test_fused_presweep_residual_restrict_2d(n, neumann, mesh, div):
    """The 2D twins of the fused kernels (k_rbgs_zero_resid2d_lds, k_restrict_fixup2d, k_rbgs_resweep_prolong2d_lds) on
    uniform quadtree levels: fuse = 2 vs 1 to rounding (the ghost term of the residual is added separately), fuse = 3
    (the iterate between the sweeps is recomputed in LDS, never stored) == fuse = 2 bit for bit, all against the oracle."""
    m, H, levels = util.setup("uniform", n, div, neumann=neumann, dim=2)
    g, L = capi.GMG(H), levels[0]
    f = util.rand_vec(L.size, 55) / L.a["h"].min() ** 2
    got, fixups = {}, {}
    for fuse in (1, 2, 3):
        df, dc = g.new_vector(0, f), g.new_vector(0)
        g.profile(True)
        g.profile_reset()
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS, fuse=fuse), df, dc)
        rows = g.profile_rows()
        g.profile(False)
        got[fuse] = dc.download()
        fixups[fuse] = rows.get("restrict_fixup", {"calls": 0})["calls"]
        assert ("rbgs_resweep_prolong" in rows) == (fuse == 3) and ("restrict_fixup" in rows) == (fuse >= 2)
    # the ghost terms of a coarse right-hand side are added by the coarse level's own pre-sweep kernel (FOLD) wherever that level
    # runs one: only the last fused level still launches k_restrict_fixup2d. Against the fix-up pass on every level: bit for bit.
    assert fixups[2] == 1 and fixups[3] == 1
    g.set_option("TE_2D_NO_FOLD", "1")
    for fuse in (2, 3):
        df, dc = g.new_vector(0, f), g.new_vector(0)
        g.profile(True)
        g.profile_reset()
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS, fuse=fuse), df, dc)
        rows = g.profile_rows()
        g.profile(False)
        assert rows["restrict_fixup"]["calls"] == div and np.array_equal(dc.download(), got[fuse])
    g.set_option("TE_2D_NO_FOLD", None)
    assert not np.array_equal(got[1], got[2])
    assert rel(got[2], got[1]) <= 1e-13
    assert np.array_equal(got[3], got[2])
    assert rel(got[3], orc.cycle(levels, orc.cycle_opts(smoother=2), f)) <= 1e-10
    # W-cycle: the stored-iterate path (fuse = 2 kernels, explicit post-sweep on u + P e)
    outs = []
    for fuse in (1, 3):
        df, dc = g.new_vector(0, f), g.new_vector(0)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS, fuse=fuse, cycle_type=1), df, dc)
        outs.append(dc.download())
    assert rel(outs[1], outs[0]) <= 1e-13


def test_patch_solve_split_between_pure_and_mixed_axes(monkeypatch):
    """Neumann boundaries: patches that touch the boundary have a Dirichlet(interface)/Neumann axis (type-IV
    transforms, k_ps_fused), interior patches have pure DST axes (k_ps_sym); the level is split per patch.
    4^3 patches of 32^3: 8 interior + 56 boundary."""
    m, H, levels = util.setup("uniform", 32, 2, neumann=True, dim=3)
    g, L = capi.GMG(H), levels[0]
    u = util.rand_vec(L.size, 43)
    f = util.rand_vec(L.size, 53) / L.a["h"].min() ** 2
    got = {}
    for mode in ("1pass", "3pass"):
        g.set_option("TE_PS_MODE", mode)
        du, df, dc = g.new_vector(0, u), g.new_vector(0, f), g.new_vector(0)
        g.smooth(df, du, level=0, smoother=capi.SMOOTH_PATCH_SOLVE)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE), df, dc)
        got[mode] = (du.download(), dc.download())
    assert rel(got["1pass"][0], orc.smooth(L, f, u)) <= 1e-11
    assert rel(got["1pass"][0], got["3pass"][0]) <= 1e-13
    assert rel(got["1pass"][1], got["3pass"][1]) <= 1e-12
    # the mixed patches alone run the same MFMA sequence as the three-pass kernels
    ids = np.asarray(L.a["id"]) if "id" in L.a else None
    diff = np.abs(got["1pass"][0] - got["3pass"][0]).reshape(L.P, -1).max(axis=1)
    assert (diff == 0).sum() >= 56 and (diff > 0).sum() <= 8


def test_reference_smoother_pre_sweep_stores_face_layers_only(monkeypatch):
    """opts.fuse = 3 with the reference smoother: the zero-guess pre-sweep of a level that takes the single-pass solve stores
    the six face layers of its result and nothing else (k_ps_sym<false, FACES>) -- its residual lives on the faces and the
    post-sweep overwrites the iterate, reading the old one through its interface terms only. Bit-identical to the stored
    iterate (TE_NO_PS_FACES, fuse = 2) and equal to the oracle; 512 patches of 32^3, V(1,1) and V(1,2)."""
    m, H, levels = util.setup("uniform", 32, 3)
    g, L = capi.GMG(H), levels[0]
    f = util.rand_vec(L.size, 77) / L.a["h"].min() ** 2
    for kw in (dict(), dict(post_sweeps=2)):
        got = {}
        for name, env, fuse in (("faces", None, 3), ("stored", "1", 3), ("fuse2", None, 2)):
            g.set_option("TE_NO_PS_FACES", env)
            df, du = g.new_vector(0, f), g.new_vector(0)
            du.set(3.0)
            g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE, fuse=fuse, **kw), df, du)
            got[name] = du.download()
        assert np.array_equal(got["faces"], got["stored"]) and np.array_equal(got["faces"], got["fuse2"])
        names = dict(post_sweeps="post")
        want = orc.cycle(levels, orc.cycle_opts(smoother=capi.SMOOTH_PATCH_SOLVE, **{names[k]: v for k, v in kw.items()}), f)
        assert rel(got["faces"], want) <= 1e-10


def test_blas1(case):
    g, L = case["g"], case["levels"][0]
    a, b, c = (util.rand_vec(L.size, s) for s in (60, 61, 62))
    va, vb, vc = g.new_vector(0, a), g.new_vector(0, b), g.new_vector(0, c)
    assert abs(va.dot(vb) - a @ b) <= 1e-12 * np.abs(a).sum()
    assert abs(va.twoNorm() - np.linalg.norm(a)) <= 1e-13 * np.linalg.norm(a)
    assert va.infNorm() == np.abs(a).max()
    va.addScaled(0.5, vb, -2.0, vc)
    a = a + (b * 0.5 + c * -2.0)
    assert np.abs(va.download() - a).max() <= 4 * util.EPS * 4
    va.scaleThenAdd(-1.0, vb)
    a = -1.0 * a + b
    assert np.abs(va.download() - a).max() <= 4 * util.EPS * 8
    va.scaleThenAddScaled(0.25, 3.0, vb, -1.5, vc)
    a = 0.25 * a + 3.0 * b + -1.5 * c
    assert np.abs(va.download() - a).max() <= 4 * util.EPS * 16
    va.shift(1.5); va.scale(2.0); a = (a + 1.5) * 2.0
    assert np.abs(va.download() - a).max() <= 4 * util.EPS * 32
    va.copy(vb); va.add(vc)
    assert np.array_equal(va.download(), b + c)
    va.set(3.25)
    assert np.all(va.download() == 3.25)


@pytest.mark.parametrize("smoother", [capi.SMOOTH_PATCH_SOLVE, capi.SMOOTH_JACOBI, capi.SMOOTH_RBGS])
@pytest.mark.parametrize("cycle_type", [0, 1])
def test_cycle(case, smoother, cycle_type):
    g, levels = case["g"], case["levels"]
    f = util.rand_vec(levels[0].size, 70)
    oo = orc.cycle_opts(smoother=smoother, cycle_type=cycle_type, omega=0.8)
    want = orc.cycle(levels, oo, f)
    got = {}
    for fuse in (0, 1):
        o = g.default_opts(smoother=smoother, cycle_type=cycle_type, omega=0.8, fuse=fuse)
        df, du = g.new_vector(0, f), g.new_vector(0)
        du.set(123.0)  # Cycle::apply must ignore the incoming u (Cycle.h:118)
        g.cycle(o, df, du)
        got[fuse] = du.download()
        assert rel(got[fuse], want) <= 1e-10
    # fused residual+restrict and the zero-guess sweep change the number of HBM passes, not one bit
    assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize("sweeps", [(0, 1, 1), (2, 0, 3), (1, 2, 0)])
def test_cycle_sweep_counts(case, sweeps):
    g, levels = case["g"], case["levels"]
    f = util.rand_vec(levels[0].size, 71)
    pre, post, coarse = sweeps
    want = orc.cycle(levels, orc.cycle_opts(pre=pre, post=post, coarse=coarse, smoother=2), f)
    for fuse in (0, 1):
        o = g.default_opts(smoother=capi.SMOOTH_RBGS, pre_sweeps=pre, post_sweeps=post, coarse_sweeps=coarse, fuse=fuse)
        df, du = g.new_vector(0, f), g.new_vector(0)
        g.cycle(o, df, du)
        assert rel(du.download(), want) <= 1e-10


def test_neumann_solve_with_zero_mean_rhs(case):
    """Pure-Neumann problems as the driver runs them (apps/3d/steady.cpp:318-334, 539-549): Init::initNeumann,
    f -= integrate(f)/volume, BiCGStab + GMG, error measured after shifting the means together.
    te_integrate / te_volume are Domain<D>::integrate / volume (Domain.h:237-278)."""
    if not case["neumann"] or case["dim"] != 3:
        pytest.skip("3D Neumann cases")
    g, levels, H, n = case["g"], case["levels"], case["H"], case["n"]
    t = H.tables(0)
    f, exact = problems.init_neumann(t, n)
    cell = np.prod(t["lengths"] / n, axis=1)
    df, de = g.new_vector(0, f), g.new_vector(0, exact)
    vol = g.volume()
    assert abs(vol - np.sum(np.prod(t["lengths"], axis=1))) <= 1e-14 * vol
    want = np.sum(f.reshape(len(cell), -1).sum(axis=1) * cell)
    assert abs(g.integrate(df) - want) <= 1e-12 * np.abs(f).sum() * cell.max()
    df.shift(-g.integrate(df) / vol)
    fz = df.download()
    for sm in (capi.SMOOTH_PATCH_SOLVE, capi.SMOOTH_RBGS):
        dx = g.new_vector(0)
        its, rr = g.bicgstab(dx, df, g.default_opts(smoother=sm), tol=1e-10)
        assert rr <= 1e-10 and its <= 40
        x_ref, its_ref, rr_ref = orc.bicgstab(levels, orc.cycle_opts(smoother=sm), fz, tol=1e-10)
        assert abs(its - its_ref) <= 2
        # compare after removing the (arbitrary) constant, as the driver does for its error norm
        uavg, eavg = g.integrate(dx) / vol, g.integrate(de) / vol
        x = dx.download()
        xr = x_ref - np.sum(x_ref.reshape(len(cell), -1).sum(axis=1) * cell) / vol
        assert rel(x - uavg, xr) <= 1e-6
        err = rel(x - uavg, exact - eavg)
        assert err <= 12.0 * (t["lengths"][:, 0].min() / n) ** 2  # second order in h


def test_bicgstab_trig(case):
    if case["neumann"]:
        pytest.skip("pure-Neumann solves: test_neumann_solve_with_zero_mean_rhs")
    g, levels, H = case["g"], case["levels"], case["H"]
    init = problems.init_dirichlet if case["dim"] == 3 else problems.init_dirichlet_2d
    f, exact = init(H.tables(0), case["n"])
    for sm in (capi.SMOOTH_PATCH_SOLVE, capi.SMOOTH_RBGS):
        o = g.default_opts(smoother=sm)
        x_ref, its_ref, rr_ref = orc.bicgstab(levels, orc.cycle_opts(smoother=sm), f)
        df, dx = g.new_vector(0, f), g.new_vector(0)
        its, rr = g.bicgstab(dx, df, o)
        x = dx.download()
        assert rr <= 1e-12 and abs(its - its_ref) <= 1
        assert rel(x, x_ref) <= 1e-8
        # same discretisation error against the analytic solution (3 significant digits)
        e, e_ref = rel(x, exact), rel(x_ref, exact)
        assert abs(e - e_ref) <= 1e-3 * e_ref


def test_fused_sums_of_the_stencil_kernel(case):
    """The residual norm and BiCGStab's dot products formed inside the stencil kernel (k_stencil3d RED: per thread in plane
    order, wave shuffles, LDS, one partial per workgroup, fixed-order final pass) against the separate reduction passes:
    te_residual_norm_sq == ||te_residual||^2 to rounding (another summation order), r itself bit-identical; te_bicgstab
    with and without the fused sums (TE_NO_BICG_FUSE): same iteration count, same solution to 1e-10."""
    g, L, H = case["g"], case["levels"][0], case["H"]
    u = util.rand_vec(L.size, 81)
    f = util.rand_vec(L.size, 82) / L.a["h"].min() ** 2
    du, df, r0, r1 = g.new_vector(0, u), g.new_vector(0, f), g.new_vector(0), g.new_vector(0)
    g.residual(du, df, r0)
    nsq = g.residual_norm_sq(du, df, r1)
    assert np.array_equal(r0.download(), r1.download())
    want = float(np.sum(r0.download() ** 2))
    assert abs(nsq - want) <= 1e-13 * want
    if case["neumann"]:
        return
    init = problems.init_dirichlet if case["dim"] == 3 else problems.init_dirichlet_2d
    b, _ = init(H.tables(0), case["n"])
    got = {}
    for fuse in (True, False):
        g.set_option("TE_NO_BICG_FUSE", None if fuse else "1")
        for o in (None, g.default_opts(smoother=capi.SMOOTH_RBGS)):
            db, dx = g.new_vector(0, b), g.new_vector(0)
            its, rr = g.bicgstab(dx, db, o, tol=1e-12 if o is not None else 1e-8, max_it=400)
            got[(fuse, o is None)] = (its, rr, dx.download())
    g.set_option("TE_NO_BICG_FUSE", None)
    for unprec in (True, False):
        a, c = got[(True, unprec)], got[(False, unprec)]
        # (unpreconditioned BiCGStab on these meshes converges erratically: a change of summation order moves the count by a few)
        assert abs(a[0] - c[0]) <= (1 if not unprec else max(3, c[0] // 8)), (a[0], c[0])
        # (unpreconditioned: both stop at a relative residual of 1e-8; the solutions agree to that times the condition number)
        assert rel(a[2], c[2]) <= (1e-10 if not unprec else 1e-4)


def test_post_sweep_tuning_variants_bit_identical():
    """k_rbgs_resweep_prolong3d's variants (TE_RESWEEP_V: which black values go back to LDS, when the top neighbour's plane is
    loaded, non-temporal loads of f / stores of u, two / three / four ring slots for the planes of f in flight) and k_rbgs_zero_resid3d's prefetch distance (TE_ZR_AHEAD) change the
    schedule, not one bit: 256^3 (the default there is 19) and the level-1 variant with exported ghost terms of a 16^3-patch
    grid of 8^3 patches."""
    for n, div in ((32, 3), (8, 4)):
        m, H, levels = util.setup("uniform", n, div)
        g = capi.GMG(H)
        f = util.rand_vec(levels[0].size, 88) / levels[0].a["h"].min() ** 2
        got = {}
        for v in (None, "0", "3", "19", "27", "31", "59"):
            g.set_option("TE_RESWEEP_V", v)
            for ah in (None, "1"):
                g.set_option("TE_ZR_AHEAD", ah)
                df, du = g.new_vector(0, f), g.new_vector(0)
                g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
                got[(v, ah)] = du.download()
        g.set_option("TE_RESWEEP_V", None)
        g.set_option("TE_ZR_AHEAD", None)
        ref = got[(None, None)]
        for k, x in got.items():
            assert np.array_equal(x, ref), k


def test_refined_post_sweep_variants_bit_identical():
    """the recomputing post-sweep of a REFINED level (copy-through patches, coarse/fine ghost slots: k_rbgs_resweep_prolong3d<..., CFP>) in
    its three-workgroups-per-CU form (TE_RESWEEP_V=27; it spills nine registers) and in the two-workgroup form a large finest level
    takes since round 6 (59: no spill; 1107 -> 1072 us per launch on `2refine --divide 3`; TE_NO_CFP59 = the former choice): same bits"""
    m, H, levels = util.setup("2refine.bin", 32, 2)
    g = capi.GMG(H)
    f = util.rand_vec(levels[0].size, 89) / levels[0].a["h"].min() ** 2
    got = {}
    for name, opts in (("default", {}), ("v59", {"TE_RESWEEP_V": "59"}), ("v27", {"TE_RESWEEP_V": "27"}), ("no59", {"TE_RESWEEP_V": "59", "TE_NO_CFP59": "1"})):
        for k, v in opts.items():
            g.set_option(k, v)
        df, du = g.new_vector(0, f), g.new_vector(0)
        g.profile(True)
        g.profile_reset()
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
        rows = g.profile_rows()
        g.profile(False)
        for k in opts:
            g.set_option(k, None)
        assert "rbgs_resweep_prolong" in rows, sorted(rows)
        got[name] = du.download()
    for name in ("v59", "v27", "no59"):
        assert np.array_equal(got["default"], got[name]), name
    assert np.linalg.norm(got["default"] - orc.cycle(levels, orc.cycle_opts(smoother=capi.SMOOTH_RBGS), f)) <= 1e-10 * np.linalg.norm(got["default"])


@pytest.mark.parametrize("div,neumann", [(2, False), (3, False), (2, True)])
def test_fused_block_jacobi_cycle_2d(div, neumann):
    """The reference smoother's V-cycle on uniform 2D levels of 64^2 patches (config C5's shape): fuse = 1 adds the prolongation
    inside the post-sweep (a block-Jacobi sweep reads the old iterate on patch edges only: k_patch_solve2d_mfma<.., PROLONG>; the
    same additions: bit-identical to fuse = 0); fuse >= 2 also takes the residual after the exact patch solves on the patch edges
    only (interfaceResidRestrict2d: it vanishes inside a patch up to the rounding of the solve) -- no k_prolong2d and no
    residual pass on those levels; against the unfused cycle <= 1e-11, against the oracle <= 1e-10."""
    m, H, levels = util.setup("uniform", 64, div, neumann=neumann, dim=2)
    g, L = capi.GMG(H), levels[0]
    f = util.rand_vec(L.size, 57) / L.a["h"].min() ** 2
    got, rows = {}, {}
    for fuse in (0, 1, 2, 3):
        df, dc = g.new_vector(0, f), g.new_vector(0)
        g.profile(True)
        g.profile_reset()
        g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE, fuse=fuse), df, dc)
        rows[fuse] = g.profile_rows()
        g.profile(False)
        got[fuse] = dc.download()
    assert np.array_equal(got[0], got[1])
    assert "prolong_add" in rows[0] and "prolong_add" not in rows[1]
    assert "resid_restrict" in rows[1] and "resid_restrict" not in rows[2] and "restrict_fixup" in rows[2]
    assert np.array_equal(got[2], got[3])
    assert rel(got[2], got[1]) <= 1e-11 and not np.array_equal(got[2], got[1])
    assert rel(got[3], orc.cycle(levels, orc.cycle_opts(smoother=0), f)) <= 1e-10
    # W-cycle: the pre-sweep and its interface residual, then mid- and post-sweeps with the prolongation inside
    outs = []
    for fuse in (0, 3):
        df, dc = g.new_vector(0, f), g.new_vector(0)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE, fuse=fuse, cycle_type=1), df, dc)
        outs.append(dc.download())
    assert rel(outs[1], outs[0]) <= 1e-11


@pytest.mark.parametrize("div,neumann", [(2, False), (3, True), (0, False), (0, True)])
def test_half_size_transforms_2d(div, neumann):
    """k_patch_solve2d_sym (64^2 patches whose two axes close the same way on both sides: half-size transforms on the sums and
    differences of mirrored entries) against k_patch_solve2d_mfma (TE_2D_NO_SYM) and the oracle: a sweep from a random iterate, a
    sweep from zero, a V-cycle. Neumann: the boundary patches have a mixed axis and keep the full transforms (two launches per
    level); one all-Neumann patch: DCT axes, the zero mode."""
    m, H, levels = util.setup("uniform", 64, div, neumann=neumann, dim=2)
    g, L = capi.GMG(H), levels[0]
    u0 = util.rand_vec(L.size, 61)
    f = util.rand_vec(L.size, 62) / L.a["h"].min() ** 2
    if neumann and L.P == 1:
        f -= f.mean()
    got = {}
    for mode in (None, "1"):
        g.set_option("TE_2D_NO_SYM", mode)
        du, dz, df, dc = g.new_vector(0, u0), g.new_vector(0), g.new_vector(0, f), g.new_vector(0)
        g.smooth(df, du, smoother=capi.SMOOTH_PATCH_SOLVE)
        g.smooth(df, dz, smoother=capi.SMOOTH_PATCH_SOLVE)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE), df, dc)
        got[mode] = (du.download(), dz.download(), dc.download())
    g.set_option("TE_2D_NO_SYM", None)
    for a, b in zip(got[None], got["1"]):
        assert rel(a, b) <= 1e-13 and not np.array_equal(a, b)
    assert rel(got[None][0], orc.smooth(L, f, u0)) <= 1e-11
    assert rel(got[None][2], orc.cycle(levels, orc.cycle_opts(smoother=0), f)) <= 1e-10
