"""CPU: the oracle (oracle/te_oracle.cpp) against golden vectors produced by the reference's own
compiled code (oracle/gen_golden.py -> tests/golden/ref_*.npz), plus the known-answer patterns of
the reference's test/GMG.cpp and an independent DST/DCT cross-check (scipy.fft) for the patch solve.
"""
import glob
import os

import numpy as np
import pytest
import scipy.fft

from oracle import oracle as orc
from tests import util

FIXTURES = sorted(glob.glob(os.path.join(util.GOLDEN, "ref_*_n*.npz")))


def level_of(d):
    return orc.Level(int(d["dim"]), int(d["n"]), d["t_id"], d["t_h"], d["t_nbr_kind"], d["t_nbr"], d["t_nbr_orth"],
                     d["t_neumann"], d["t_parent"], d["t_orth_on_parent"])


@pytest.fixture(params=FIXTURES, ids=[os.path.basename(f)[4:-4] for f in FIXTURES])
def gold(request):
    d = dict(np.load(request.param))
    return d, level_of(d)


def test_fixtures_present():
    assert len(FIXTURES) >= 9


def test_iface_numbering(gold):
    d, L = gold
    assert L.num_ifaces() == int(d["num_ifaces"])
    assert np.array_equal(L.iface_index(), d["iface_index"])


def test_interp_a6(gold):
    d, L = gold
    if L.num_ifaces() == 0:
        pytest.skip("single patch: no interfaces")
    g = orc.interp(L, d["u"])
    assert np.abs(g - d["gamma"]).max() <= 4 * util.EPS * np.abs(d["u"]).max()


def test_apply_with_interface_a3(gold):
    d, L = gold
    tol = util.op_tol(L, np.concatenate([d["u"], d["gamma_in"]]))
    assert np.abs(orc.apply_with_gamma(L, d["u"], d["gamma_in"]) - d["apply_with_gamma"]).max() <= tol
    assert np.abs(orc.apply(L, d["u"]) - d["apply"]).max() <= tol


def test_patch_apply_a4(gold):
    d, L = gold
    assert np.abs(orc.patch_apply(L, d["u"]) - d["patch_apply"]).max() <= util.op_tol(L, d["u"])


def test_add_interface_to_rhs_a5(gold):
    d, L = gold
    got = orc.add_iface_rhs(L, d["gamma_in"], d["f"])
    tol = util.op_tol(L, d["gamma_in"]) if L.num_ifaces() else 0.0
    assert np.abs(got - d["add_iface_rhs"]).max() <= tol


def test_patch_solve_inverts_reference_operator_a9(gold):
    """u = patch_solve(gamma, f) must satisfy the reference's own per-patch operator:
    StarPatchOp::apply(u) == f - (2/h^2) gamma|faces (both right-hand sides are golden)."""
    d, L = gold
    if int(d["neumann"]) and L.P == 1:
        pytest.skip("pure Neumann single patch: singular, covered by test_neumann_single_patch")
    rhs = d["add_iface_rhs"]  # reference's f - 2 gamma / h^2
    u = orc.patch_solve(L, d["gamma_in"], d["f"])
    back = orc.patch_apply(L, u)  # pinned to the reference by test_patch_apply_a4
    scale = np.abs(rhs).max()
    assert np.abs(back - rhs).max() <= 1e-11 * scale


def test_neumann_single_patch():
    d = dict(np.load(os.path.join(util.GOLDEN, "ref_1uni_n8_neumann.npz")))
    L = level_of(d)
    f = d["f"] - d["f"].mean()  # compatible right-hand side
    u = orc.patch_solve(L, np.zeros(0), f)
    assert np.abs(orc.patch_apply(L, u) - f).max() <= 1e-11 * np.abs(f).max()
    # FftwPatchSolver.h:197 zeroes mode 0: the DCT-II mean of the solution vanishes
    assert abs(u.mean()) <= 1e-12 * np.abs(u).max()


def test_bicgstab_against_reference(gold):
    d, L = gold
    if "bicg_x" not in d:
        pytest.skip("no solve stored for Neumann fixtures")
    x, its, rr = orc.bicgstab([L], orc.cycle_opts(), d["f"], use_prec=False)
    assert rr <= 1e-12
    assert abs(its - int(d["bicg_its"])) <= max(3, int(d["bicg_its"]) // 10)  # rounding-order sensitive
    assert np.linalg.norm(x - d["bicg_x"]) <= 1e-9 * np.linalg.norm(d["bicg_x"])


def test_patch_solve_matches_scipy_dst():
    """Dirichlet single patch: u = DST-III( DST-II(f) / eig ) in scipy's (FFTW-compatible) convention."""
    n, h = 8, 1.0 / 8
    L = orc.Level(3, n, [0], [[h, h, h]], np.zeros((1, 6), np.int32), -np.ones((1, 6, 4), np.int32),
                  -np.ones((1, 6), np.int32), [0], [-1], [-1])
    f = util.rand_vec(n ** 3, 5)
    u = orc.patch_solve(L, np.zeros(0), f).reshape(n, n, n)
    F = scipy.fft.dstn(f.reshape(n, n, n), type=2)
    k = np.arange(n)
    lam = -4 / h ** 2 * np.sin((k + 1) * np.pi / (2 * n)) ** 2
    eig = lam[:, None, None] + lam[None, :, None] + lam[None, None, :]
    ref = scipy.fft.idstn(F / eig, type=2)
    assert np.abs(u - ref).max() <= 1e-12 * np.abs(ref).max()


# ---- a11 / a12: known-answer patterns of the reference's (disabled) test/GMG.cpp:261-435 ----------
def _two_levels(name):
    m, H, levels = util.setup(name, 4)
    return H, levels


@pytest.mark.parametrize("name", ["2uni.bin", "2refine.bin"])
def test_restrict_known_answer(name):
    """fine patch value = id + xi/2 + yi/2*n + zi/2*n^2 (id + xi + yi*n + zi*n^2 for patches that do
    not coarsen) restricts exactly to `octFill`: the coarse cell of orthant o holds child_id + ..."""
    H, levels = _two_levels(name)
    fine, coarse = levels[0], levels[1]
    n = 4
    zi, yi, xi = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    fv = np.zeros((fine.P, n, n, n))
    expect = np.zeros((coarse.P, n, n, n))
    for p in range(fine.P):
        pid, o, par = fine.a["id"][p], fine.a["orth_on_parent"][p], fine.a["parent"][p]
        if o >= 0:
            fv[p] = pid + xi // 2 + yi // 2 * n + zi // 2 * n * n
            h = n // 2
            sl = (slice(h * ((o >> 2) & 1), h * ((o >> 2) & 1) + h), slice(h * ((o >> 1) & 1), h * ((o >> 1) & 1) + h),
                  slice(h * (o & 1), h * (o & 1) + h))
            expect[par][sl] = pid + xi[:h, :h, :h] + yi[:h, :h, :h] * n + zi[:h, :h, :h] * n * n
        else:
            fv[p] = pid + xi + yi * n + zi * n * n
            expect[par] = fv[p]
    got = orc.restrict(fine, coarse, fv.ravel())
    assert np.array_equal(got, expect.ravel())  # exact equality, as the reference test demands
    # DrctIntp is the transpose without the 1/8: prolong the expected coarse vector onto zeros
    back = orc.prolong_add(fine, coarse, expect.ravel(), np.zeros(fine.size))
    assert np.array_equal(back, fv.ravel())


def test_cycle_reduces_residual():
    m, H, levels = util.setup("2refine.bin", 8)
    f = util.rand_vec(levels[0].size, 9)
    for sm, bound in ((0, 0.2), (1, 0.75), (2, 0.6)):
        u = orc.cycle(levels, orc.cycle_opts(smoother=sm), f)
        r = f - orc.apply(levels[0], u)
        assert np.linalg.norm(r) <= bound * np.linalg.norm(f)


@pytest.mark.skipif(not __import__("oracle.refslice", fromlist=["x"]).available(), reason="reference slice not built here")
@pytest.mark.parametrize("name,n,div", [("3uni.bin", 4, 0), ("2refine.bin", 4, 1), ("2d2ref.bin", 8, 1)])
def test_live_against_reference_slice(name, n, div):
    """Where /root/reference was available at build time: same comparisons on fresh seeded inputs."""
    from oracle import refslice
    dim = 2 if name.startswith("2d") else 3
    m, H, levels = util.setup(name, n, div, dim=dim)
    L = levels[0]
    u = util.rand_vec(L.size, 77)
    assert np.abs(orc.interp(L, u) - refslice.interp(L, u)).max() <= 4 * util.EPS
    assert np.abs(orc.apply(L, u) - refslice.apply(L, u)).max() <= util.op_tol(L, u)
    assert np.abs(orc.patch_apply(L, u) - refslice.patch_apply(L, u)).max() <= util.op_tol(L, u)


# ---- the reference's Krylov patch solver (PatchSolvers/BiCGStabSolver.h:114-132, apps/2d/steady.cpp:326-327) ----------------------
BCGS_FIXTURES = sorted(glob.glob(os.path.join(util.GOLDEN, "bcgs_ref_*.npz")))


@pytest.mark.parametrize("path", BCGS_FIXTURES, ids=[os.path.basename(f)[9:-4] for f in BCGS_FIXTURES])
def test_patch_bicgstab_sweep_against_reference(path):
    """orc_smooth_bcgs against a sweep of the reference's own BiCGStab<D>::solve per patch on StarPatchOp<D>::apply
    (oracle/gen_golden.py bcgs_case): the patch systems are solved to 1e-12, so the iterates agree far below that times the
    condition number; iteration counts are rounding-order sensitive by one or two."""
    d = dict(np.load(path))
    L = level_of(d)
    u, its = orc.smooth_bcgs(L, d["f"], d["u"], float(d["tol"]), int(d["max_it"]))
    assert np.abs(u - d["u_out"]).max() <= 1e-9 * np.abs(d["u_out"]).max()
    assert np.abs(its.astype(int) - d["its"].astype(int)).max() <= 2
    # each patch's system is solved: StarPatchOp::apply(u_new) == f - interface terms of the OLD iterate
    rhs = orc.add_iface_rhs(L, orc.interp(L, d["u"]), d["f"])
    assert np.abs(orc.patch_apply(L, u) - rhs).max() <= 1e-9 * np.abs(rhs).max()


def test_patch_bicgstab_fixtures_present():
    assert len(BCGS_FIXTURES) >= 4


def test_patch_bicgstab_sweep_is_the_exact_block_jacobi_sweep_at_tight_tolerance():
    """both patch solvers solve the same patch systems: one sweep of either from the same iterate agrees to the tolerance"""
    m, H, levels = util.setup("2d2ref.bin", 8, 1, dim=2)
    L = levels[0]
    f, u0 = util.rand_vec(L.size, 5), util.rand_vec(L.size, 6)
    a = orc.smooth(L, f, u0)
    b, its = orc.smooth_bcgs(L, f, u0, 1e-13, 1000)
    assert np.abs(a - b).max() <= 1e-9 * np.abs(a).max() and its.max() < 100


def test_cycle_with_the_krylov_patch_solver_converges_like_the_exact_one():
    m, H, levels = util.setup("2d2ref.bin", 8, 1, dim=2)
    f = util.rand_vec(levels[0].size, 9)
    a = orc.cycle(levels, orc.cycle_opts(smoother=0), f)
    b = orc.cycle(levels, orc.cycle_opts(smoother=3), f)
    assert np.abs(a - b).max() <= 1e-8 * np.abs(a).max()
