"""-m gpu: the sharded path at BASELINE.json's PRODUCTION shapes, default options, default switches (default
TE_OVERLAP_MIN / TE_AGGLOMERATE; C3 also with the overlap path forced on at its 512 local patches), both smoothers, against
the single-rank run BIT FOR BIT:

  C3  512^3 uniform, 4096 patches of 32^3, 8 ranks (2x2x2 octants of 8^3 patches)
  C4  2refine.bin --divide 2 (960 patches of 32^3, coarse/fine faces cut by rank boundaries), 4 ranks
  C5  4096^2 uniform, 4096 patches of 64^2, 8 ranks

Virtual ranks (dist.LocalFabric: one thread per rank on the one GPU of the test box, device-to-device copies behind the
same exchange plans). What this replaces in the reference: the VecScatter pairs of SchurHelper.h:123-150 and
GMG/InterLevelComm.h:169-189. The native RCCL back-end cannot carry these runs on a one-GPU box: RCCL refuses two ranks
on one device (ncclCommInitRank: duplicate GPU), so it is exercised with the rank as its own peer
(test_gpu_multirank.py::test_native_rccl_backend_selftest) and through bench.py on real multi-GPU nodes only; the
exchange PLANS (who sends what to whom, in which order) are the same objects for every back-end."""
import numpy as np
import pytest

from pressurepoissonsolver_amd import capi, dist as tedist, problems
from tests import util

pytestmark = pytest.mark.gpu

CONFIGS = {
    "C3-512^3-8ranks": dict(mesh="uniform", divides=4, n=32, dim=3, nranks=8),
    "C4-2refine-div2-4ranks": dict(mesh="2refine.bin", divides=2, n=32, dim=3, nranks=4),
    "C5-4096^2-8ranks": dict(mesh="uniform", divides=6, n=64, dim=2, nranks=8),
}


def cycle_single(mesh, n, sm, f):
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    df, du = g1.new_vector(0, f), g1.new_vector(0)
    g1.profile(True)
    g1.profile_reset()
    g1.cycle(g1.default_opts(smoother=sm), df, du)
    rows = g1.profile_rows()
    g1.profile(False)
    want = du.download()
    del df, du, g1
    return want, rows


# (name, TE_OVERLAP_MIN, TE_REPLICATE); None = the default
# replicate = "packed": levels on every rank, but with round 3's transport (every exchange behind a pack kernel, the restricted
# blocks through send / receive buffers, a second face exchange for level 1's post-sweep): TE_PACK_FACES, TE_REPL_BLOCKS, TE_POST_EXCHANGE
CASES = [(name, None, None) for name in CONFIGS] + [("C3-512^3-8ranks", "128", None), ("C3-512^3-8ranks", None, "0"),
                                                      ("C4-2refine-div2-4ranks", None, "0"), ("C3-512^3-8ranks", None, "packed")]


def case_id(c):
    return c[0] + ("" if c[1] is None else "-overlap-forced") + ("" if c[2] is None else ("-gathered-on-rank0" if c[2] == "0" else "-packed"))


@pytest.mark.parametrize("smoother", [capi.SMOOTH_RBGS, capi.SMOOTH_PATCH_SOLVE], ids=["rbgs", "patch_solve"])
@pytest.mark.parametrize("name,overlap_min,replicate", CASES, ids=[case_id(c) for c in CASES])
def test_production_shape_sharded_cycle_equals_single_rank(name, overlap_min, replicate, smoother, monkeypatch):
    packed = replicate == "packed"
    for k in ("TE_OVERLAP_MIN", "TE_AGGLOMERATE", "TE_AGGLOMERATE_MAX", "TE_NO_OVERLAP", "TE_REPLICATE", "TE_POST_EXCHANGE", "TE_REPL_BLOCKS", "TE_PACK_FACES"):
        monkeypatch.delenv(k, raising=False)  # the defaults are what is under test
    if replicate == "packed":
        monkeypatch.setenv("TE_POST_EXCHANGE", "1")
        monkeypatch.setenv("TE_REPL_BLOCKS", "1")
        monkeypatch.setenv("TE_PACK_FACES", "1")
        replicate = None
    if overlap_min is not None:  # the interior/boundary split at a shape where the default (768 local patches) leaves it off
        monkeypatch.setenv("TE_OVERLAP_MIN", overlap_min)
    if replicate is not None:  # the gathered levels on rank 0 alone (round 2's form) instead of on every rank
        monkeypatch.setenv("TE_REPLICATE", replicate)
    c = CONFIGS[name]
    n, dim, nranks = c["n"], c["dim"], c["nranks"]
    mesh = util.mesh(c["mesh"], c["divides"], dim)
    nc = n ** dim
    H0 = capi.Hierarchy(mesh, n)
    P = H0.sizes(0)[1]
    f = problems.random_rhs(H0.tables(0)["id"], nc)  # the timing input of bench.py
    del H0
    want, rows1 = cycle_single(mesh, n, smoother, f)
    if smoother == capi.SMOOTH_RBGS and c["mesh"] == "uniform":
        assert "rbgs_resweep_prolong" in rows1  # the benchmarked fused path is what is being compared

    fab = tedist.LocalFabric(nranks)
    fab.timeout = 600.0
    hs = [capi.Hierarchy(mesh, n, rank=r, nranks=nranks) for r in range(nranks)]
    gs = [capi.GMG(h) for h in hs]
    for r, g in enumerate(gs):
        fab.attach(g, r)

    def per_rank(r):
        H, g = hs[r], gs[r]
        idx = H.l2g(0)
        df, du = g.new_vector(0, f.reshape(-1, nc)[idx].ravel()), g.new_vector(0)
        g.profile(True)
        g.profile_reset()
        g.cycle(g.default_opts(smoother=smoother), df, du)
        rows = g.profile_rows()
        g.profile(False)
        return idx, du.download(), rows

    outs = fab.run(per_rank)
    got = np.zeros(P * nc)
    for idx, u, _ in outs:
        got.reshape(P, nc)[idx] = u.reshape(len(idx), nc)
    assert np.array_equal(got, want), np.abs(got - want).max()

    rows = [o[2] for o in outs]
    if name.startswith("C3") and smoother == capi.SMOOTH_RBGS:
        # forced: interior (343) and boundary (169) patches of the level-0 post-sweep are two launches with the exchange under the
        # first; default: 512 local patches < TE_OVERLAP_MIN (768), one launch behind the exchange. Level 1 (64 local patches)
        # runs in one launch of the FCORR symbol either way
        for r in rows:
            assert r["rbgs_resweep_prolong"]["calls"] == (2 if overlap_min is not None else 1), r["rbgs_resweep_prolong"]
            assert r["rbgs_zero_resid_restrict_faces"]["calls"] == 1
            if replicate is None and not packed:
                # the face layers of the two pre-sweeps leave from where they lie (LevelHost::f6off), level 1's restricted blocks are
                # whole coarse patches exchanged in place, and level 1's post-sweep needs nothing from other ranks (its parents live
                # on every rank): ONE pack launch per cycle (level 0's face layers of v + P e) and four exchanges
                # (an exchange on the communication stream -- the overlapped level-0 post-sweep -- is not in the solver stream's table)
                assert r["pack"]["calls"] == 1 and r["exchange"]["calls"] == (4 if overlap_min is None else 3), (r["pack"], r["exchange"])
            elif packed:
                assert r["pack"]["calls"] == 6 and r["exchange"]["calls"] == 5, (r["pack"], r["exchange"])
            else:
                assert r["pack"]["calls"] >= 2 and r["exchange"]["calls"] >= 1
    if name.startswith("C3"):
        # levels with 64, 8 and 1 patches are gathered: on every rank (each computes them itself; nothing travels back up), or on
        # rank 0 alone (the other ranks launch nothing there)
        assert hs[0].sizes(2)[0] == 64 and hs[1].sizes(2)[0] == (0 if replicate == "0" else 64)
        up = [r.get("exchange", {"calls": 0})["calls"] for r in rows]
        assert len(set(up)) == 1 or replicate == "0", up  # replicated: every rank issues the same exchanges
    if name.startswith("C5") and smoother == capi.SMOOTH_RBGS:
        for r in rows:
            assert r["rbgs_resweep_prolong"]["calls"] >= 2  # levels 0 and 1 are cut by rank boundaries and stay fused


def test_production_shape_sharded_bicgstab_c3():
    """te_bicgstab at C3's shape on 8 virtual ranks: the level-0 pre-sweeps form the Krylov vectors s and p themselves (FSrc
    variants of k_rbgs_zero_resid3d), the dot products come out of the stencil kernel, scalars are summed over the ranks.
    Same iteration count as one rank, same solution to the accuracy of the reordered sums."""
    c = CONFIGS["C3-512^3-8ranks"]
    n, nranks = c["n"], c["nranks"]
    mesh = util.mesh(c["mesh"], c["divides"], 3)
    nc = n ** 3
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    P = H1.sizes(0)[1]
    b1, x1 = g1.new_vector(0), g1.new_vector(0)
    g1.init_problem(b1, None, problem=capi.PROBLEM_TRIG)
    its1, rr1 = g1.bicgstab(x1, b1, g1.default_opts(smoother=capi.SMOOTH_RBGS))
    want = x1.download()
    del b1, x1, g1
    fab = tedist.LocalFabric(nranks)
    fab.timeout = 600.0
    hs = [capi.Hierarchy(mesh, n, rank=r, nranks=nranks) for r in range(nranks)]
    gs = [capi.GMG(h) for h in hs]
    for r, g in enumerate(gs):
        fab.attach(g, r)

    def per_rank(r):
        g = gs[r]
        b, x = g.new_vector(0), g.new_vector(0)
        g.init_problem(b, None, problem=capi.PROBLEM_TRIG)
        its, rr = g.bicgstab(x, b, g.default_opts(smoother=capi.SMOOTH_RBGS))
        return hs[r].l2g(0), x.download(), its, rr

    outs = fab.run(per_rank)
    got = np.zeros(P * nc)
    for idx, x, _, _ in outs:
        got.reshape(P, nc)[idx] = x.reshape(len(idx), nc)
    assert len({o[2] for o in outs}) == 1 and abs(outs[0][2] - its1) <= 1, ([o[2] for o in outs], its1)
    assert max(o[3] for o in outs) <= 1e-12 and rr1 <= 1e-12
    assert np.linalg.norm(got - want) <= 1e-9 * np.linalg.norm(want)
