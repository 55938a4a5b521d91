"""-m gpu: the native multi-rank path (remote ghost faces, cross-rank restriction / prolongation)
with 2 / 4 / 8 VIRTUAL ranks sharing the one GPU of the test box (dist.LocalFabric: one thread per
rank, device-to-device copies). Every per-cell operation is independent of the partition, so the
sharded result must equal the single-rank result BIT FOR BIT; BiCGStab differs only through the
order of the dot-product sums (tolerance 1e-9 on the solution, same iteration count +-1).
The torch.distributed binding of the same callback is covered by tests/test_dist_gloo.py (CPU)
and rehearsed with real processes in test_two_processes_gloo below."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pressurepoissonsolver_amd import capi, dist as tedist, problems, solver
from tests import util

pytestmark = pytest.mark.gpu


def shard_run(mesh, n, nranks, fn, dim=3):
    """fn(rank, H, g) -> dict of local arrays on level 0; returns them assembled in global order"""
    fab = tedist.LocalFabric(nranks)
    hs = [capi.Hierarchy(mesh, n, rank=r, nranks=nranks) for r in range(nranks)]
    gs = [capi.GMG(h) for h in hs]
    for r, g in enumerate(gs):
        fab.attach(g, r)
    outs = fab.run(lambda r: fn(r, hs[r], gs[r], fab))
    P = hs[0].sizes(0)[1]
    nc = n ** dim
    merged = {}
    for k in outs[0]:
        if isinstance(outs[0][k], np.ndarray):
            full = np.zeros(P * nc)
            for r in range(nranks):
                idx = hs[r].l2g(0)
                full.reshape(P, nc)[idx] = outs[r][k].reshape(len(idx), nc)
            merged[k] = full
        else:
            merged[k] = [o[k] for o in outs]
    return merged


@pytest.mark.parametrize("nranks", [2, 3, 4, 8])
@pytest.mark.parametrize("name,divides,n,dim", [("uniform", 2, 8, 3), ("uniform", 3, 4, 3), ("uniform", 3, 8, 2),
                                                # refined trees: coarse/fine faces cut by rank boundaries (config C4)
                                                ("2refine.bin", 1, 8, 3), ("2d2ref.bin", 2, 8, 2),
                                                # >= 256 patches on the refined finest level: the default fuse = 3 path there
                                                # (copy-through patches and coarse/fine faces with neighbours on other ranks)
                                                ("2refine.bin", 2, 4, 3)])
def test_sharded_ops_equal_single_rank(nranks, name, divides, n, dim, monkeypatch):
    # levels with fewer than 768 local patches skip the interior/boundary overlap by default: these small meshes must
    # exercise it (the 8-rank run keeps the default, i.e. covers the non-overlapped path too); the 2-rank runs take its
    # second form (interior patches on the second stream, exchange + boundary patches on the solver stream)
    if nranks != 8:
        monkeypatch.setenv("TE_OVERLAP_MIN", "0")
    if nranks == 2:
        monkeypatch.setenv("TE_OVERLAP_MODE", "2")
    # coarse levels with few patches per rank are gathered by default (TE_AGGLOMERATE = 16 per rank) -- on every rank in 3D
    # (TE_REPLICATE, the default: 2 and 8 ranks here), on rank 0 alone otherwise (3 ranks here); the 4-rank runs keep every
    # level spread out: all three placements are compared with the single-rank result
    monkeypatch.setenv("TE_AGGLOMERATE", "0" if nranks == 4 else "16")
    monkeypatch.setenv("TE_REPLICATE", "0" if nranks == 3 else "1")
    mesh = util.mesh(name, divides, dim)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    size = H1.cells(0)
    u = util.rand_vec(size, 1)
    f = util.rand_vec(size, 2)
    nc = n ** dim

    def single(op):
        du, df, dr = g1.new_vector(0, u), g1.new_vector(0, f), g1.new_vector(0)
        op(g1, du, df, dr)
        return dr.download()

    ops = {
        "apply": lambda g, du, df, dr: g.apply(du, dr),
        "resid": lambda g, du, df, dr: g.residual(du, df, dr),
        "rbgs": lambda g, du, df, dr: (g.smooth(df, du, smoother=capi.SMOOTH_RBGS), dr.copy(du)),
        "jacobi": lambda g, du, df, dr: (g.smooth(df, du, smoother=capi.SMOOTH_JACOBI, omega=0.8), dr.copy(du)),
        "patch": lambda g, du, df, dr: (g.smooth(df, du, smoother=capi.SMOOTH_PATCH_SOLVE), dr.copy(du)),
        "vcycle_rbgs": lambda g, du, df, dr: g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, dr),
        "vcycle_rbgs_unfused": lambda g, du, df, dr: g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS, fuse=0), df, dr),
        "vcycle_patch": lambda g, du, df, dr: g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE), df, dr),
        "wcycle_jacobi": lambda g, du, df, dr: g.cycle(g.default_opts(smoother=capi.SMOOTH_JACOBI, cycle_type=1,
                                                                      omega=0.8), df, dr),
    }
    want = {k: single(op) for k, op in ops.items()}

    def per_rank(r, H, g, fab):
        idx = H.l2g(0)
        lu = u.reshape(-1, nc)[idx].ravel()
        lf = f.reshape(-1, nc)[idx].ravel()
        out = {}
        for k, op in ops.items():
            du, df, dr = g.new_vector(0, lu), g.new_vector(0, lf), g.new_vector(0)
            op(g, du, df, dr)
            out[k] = dr.download()
        return out

    got = shard_run(mesh, n, nranks, per_rank, dim)
    for k in ops:
        assert np.array_equal(got[k], want[k]), (k, np.abs(got[k] - want[k]).max())


@pytest.mark.parametrize("nranks", [2, 8])
def test_sharded_exported_ghost_terms(nranks, monkeypatch):
    """Two fused levels in a row (4096 and 512 patches): the ghost terms of the restricted residual are exported by
    the patches that own the face values and gathered into the coarse level's side array (k_fcorr_gather3d); across a
    rank boundary they are formed from the ghost slots. Sharded == single rank == the fix-up pass (TE_NO_FCORR),
    bit for bit."""
    n = 4
    mesh = util.mesh("uniform", 4)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f = util.rand_vec(H1.cells(0), 5)
    nc = n ** 3
    want = {}
    for nofc in (False, True):
        g1.set_option("TE_NO_FCORR", "1" if nofc else None)
        df, du = g1.new_vector(0, f), g1.new_vector(0)
        g1.profile(True)
        g1.profile_reset()
        g1.cycle(g1.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
        rows = g1.profile_rows()
        g1.profile(False)
        want[nofc] = du.download()
        assert rows["rbgs_zero_resid_restrict_faces"]["calls"] + rows.get("rbgs_zero_resid_restrict_faces_fcorr", {"calls": 0})["calls"] == 2
        assert ("fcorr_gather" in rows) == (not nofc) and ("rbgs_resweep_prolong_fcorr" in rows) == (not nofc)
    assert np.array_equal(want[False], want[True])
    g1.set_option("TE_NO_FCORR", None)

    def per_rank(r, H, g, fab):
        idx = H.l2g(0)
        df, du = g.new_vector(0, f.reshape(-1, nc)[idx].ravel()), g.new_vector(0)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
        return {"u": du.download()}

    got = shard_run(mesh, n, nranks, per_rank)
    assert np.array_equal(got["u"], want[False])


@pytest.mark.parametrize("nranks,n", [(2, 64), (4, 64), (4, 16)])
def test_sharded_2d_recomputing_post_sweep(nranks, n, monkeypatch):
    """2D uniform levels cut by rank boundaries take the fuse = 3 path too: the neighbour's rank sends the facing values of
    v + P(coarse) (k_pack_faces_prolong2d: the sum a local neighbour's value would get) and k_rbgs_resweep_prolong2d_lds reads
    them from ghost slots. Sharded == single rank bit for bit, for the stored-iterate path (fuse = 2) as well."""
    monkeypatch.setenv("TE_AGGLOMERATE", "0")  # every level stays spread out: rank boundaries on all of them
    monkeypatch.delenv("TE_2D_NO_MR_FUSE", raising=False)
    mesh = util.mesh("uniform", 3, 2)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f = util.rand_vec(H1.cells(0), 9)
    nc = n * n
    want = {}
    for fuse in (2, 3):
        df, du = g1.new_vector(0, f), g1.new_vector(0)
        g1.cycle(g1.default_opts(smoother=capi.SMOOTH_RBGS, fuse=fuse), df, du)
        want[fuse] = du.download()
    assert np.array_equal(want[2], want[3])

    def per_rank(r, H, g, fab):
        idx = H.l2g(0)
        out = {}
        for fuse in (2, 3):
            df, du = g.new_vector(0, f.reshape(-1, nc)[idx].ravel()), g.new_vector(0)
            g.profile(True)
            g.profile_reset()
            g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS, fuse=fuse), df, du)
            rows = g.profile_rows()
            g.profile(False)
            out[f"u{fuse}"] = du.download()
            out[f"resweep{fuse}"] = rows.get("rbgs_resweep_prolong", {"calls": 0})["calls"]
            out[f"fused_prolong{fuse}"] = rows.get("stencil_rbgs_prolong", {"calls": 0})["calls"]
        return out

    got = shard_run(mesh, n, nranks, per_rank, dim=2)
    assert np.array_equal(got["u3"], want[3]) and np.array_equal(got["u2"], want[2])
    # level 0 (64 patches) and level 1 (16) are cut by rank boundaries on every rank that owns patches of them
    assert all(c >= 2 for c in got["resweep3"]), got["resweep3"]
    assert all(c >= 2 for c in got["fused_prolong2"]), got["fused_prolong2"]


def test_sharded_reference_smoother_face_layers(monkeypatch):
    """The reference smoother's pre-sweep that stores face layers only (k_ps_sym<false, FACES>), cut by a rank boundary: the
    neighbour's rank receives the layers packed from the face buffer. 512 patches of 32^3 on 2 ranks == single rank."""
    monkeypatch.setenv("TE_AGGLOMERATE", "16")
    n = 32
    mesh = util.mesh("uniform", 3)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f = util.rand_vec(H1.cells(0), 11)
    nc = n ** 3
    df, du = g1.new_vector(0, f), g1.new_vector(0)
    g1.cycle(g1.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE), df, du)
    want = du.download()
    del df, du

    def per_rank(r, H, g, fab):
        idx = H.l2g(0)
        df, du = g.new_vector(0, f.reshape(-1, nc)[idx].ravel()), g.new_vector(0)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE), df, du)
        return {"u": du.download()}

    got = shard_run(mesh, n, 2, per_rank)
    assert np.array_equal(got["u"], want)


@pytest.mark.parametrize("nranks", [2, 8])
def test_sharded_bicgstab(nranks):
    """te_bicgstab on a sharded hierarchy (scalars summed over the ranks through the registered all-reduce, all ranks
    take the same branches) against the single-rank solve and against the statement-by-statement host mirror
    solver.bicgstab_host (one reduction per scalar, the reference's call pattern BiCGStab.h:71-97)."""
    n = 8
    mesh = util.mesh("uniform", 2)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f, exact = problems.init_dirichlet(H1.tables(0), n)
    x1 = g1.new_vector(0)
    its1, rr1 = g1.bicgstab(x1, g1.new_vector(0, f), g1.default_opts(smoother=capi.SMOOTH_RBGS))
    want = x1.download()
    nc = n ** 3

    def per_rank(r, H, g, fab):
        idx = H.l2g(0)
        b = g.new_vector(0, f.reshape(-1, nc)[idx].ravel())
        x = g.new_vector(0)
        its, rr = solver.bicgstab(g, x, b, g.default_opts(smoother=capi.SMOOTH_RBGS))  # native, all-reduce of fab.attach
        xh = g.new_vector(0)
        itsh, rrh = solver.bicgstab_host(g, xh, b, g.default_opts(smoother=capi.SMOOTH_RBGS), allreduce=fab.allreduce(r))
        return {"x": x.download(), "its": its, "rr": rr, "xh": xh.download(), "itsh": itsh, "rrh": rrh}

    got = shard_run(mesh, n, nranks, per_rank)
    assert len(set(got["its"])) == 1 and len(set(got["rr"])) == 1  # every rank saw the same scalars
    assert all(abs(i - its1) <= 1 for i in got["its"]) and max(got["rr"]) <= 1e-12
    assert np.linalg.norm(got["x"] - want) <= 1e-9 * np.linalg.norm(want)
    assert all(abs(i - j) <= 1 for i, j in zip(got["its"], got["itsh"])) and max(got["rrh"]) <= 1e-12
    assert np.linalg.norm(got["x"] - got["xh"]) <= 1e-9 * np.linalg.norm(want)


@pytest.mark.parametrize("nranks", [2, 8])
def test_autotune_chooses_alike_on_every_rank_and_changes_no_result(nranks, monkeypatch):
    """te_gmg_autotune on virtual ranks: every rank is handed the same candidate times (maximum over the ranks) and makes the
    same choice; whatever it chose, the cycle afterwards equals the single-rank cycle bit for bit -- also with each of the
    three forms forced on every sharded level (what a different machine might choose)."""
    monkeypatch.delenv("TE_OVERLAP_MIN", raising=False)
    n = 8
    mesh = util.mesh("uniform", 3)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f = util.rand_vec(H1.cells(0), 5)
    d1 = g1.new_vector(0)
    g1.cycle(g1.default_opts(smoother=capi.SMOOTH_RBGS), g1.new_vector(0, f), d1)
    want = d1.download()
    nc = n ** 3

    def per_rank(r, H, g, fab):
        o = g.default_opts(smoother=capi.SMOOTH_RBGS)
        ms, rep = g.autotune(o, reps=3)
        df, du = g.new_vector(0, f.reshape(-1, nc)[H.l2g(0)].ravel()), g.new_vector(0)
        g.cycle(o, df, du)
        return {"u": du.download(), "ms": ms, "rep": rep}

    got = shard_run(mesh, n, nranks, per_rank)
    assert len(set(got["rep"])) == 1 and len(set(got["ms"])) == 1, got["rep"]
    assert "serial=" in got["rep"][0] and "->" in got["rep"][0]
    assert np.array_equal(got["u"], want)


def test_placement_mismatch_is_reported_by_name():
    """Ranks that built their hierarchies with different placements of the small levels (different environments) are told so,
    by name, before the first cycle -- also with TE_NO_VERIFY (the schedule check would have caught it; without it the run
    used to hang until the watchdog fired)."""
    n = 8
    mesh = util.mesh("uniform", 2)
    fab = tedist.LocalFabric(2)
    hs = [capi.Hierarchy(mesh, n, rank=r, nranks=2, placement=(16, 64, 1 if r == 0 else 0)) for r in range(2)]
    assert hs[0].placement() == (16.0, 64, 1) and hs[1].placement() == (16.0, 64, 0)
    gs = [capi.GMG(h) for h in hs]
    for r, g in enumerate(gs):
        fab.attach(g, r)
        g.set_option("TE_NO_VERIFY", "1")

    def per_rank(r):
        g = gs[r]
        try:
            g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), g.new_vector(0), g.new_vector(0))
        except capi.TeError as e:
            return str(e)
        return "no error"

    msgs = fab.run(per_rank)
    assert all("replicate (TE_REPLICATE)" in m and "different hierarchies" in m for m in msgs), msgs


def test_sums_over_a_replicated_level_count_it_once():
    """A level that lives on every rank (TE_REPLICATE): te_integrate, te_volume, te_vec_two_norm_sq and te_vec_dot return the
    level's value on rank 0 and zero elsewhere, so that a host that adds the ranks (as it does on every other level) gets
    the level's integral -- not nranks times it."""
    n = 8
    nranks = 4
    mesh = util.mesh("uniform", 2)  # 64 + 8 + 1 patches: the two coarse levels are gathered on every rank
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    lvl = 1
    v = util.rand_vec(H1.cells(lvl), 9)
    d1 = g1.new_vector(lvl, v)
    want = (g1.integrate(d1, level=lvl), g1.volume(level=lvl), d1.twoNormSqLocal(), d1.dot(d1))

    def per_rank(r, H, g, fab):
        assert H.replicated(lvl) and H.sizes(lvl)[0] == H.sizes(lvl)[1]
        d = g.new_vector(lvl, v)
        return {"vals": (g.integrate(d, level=lvl), g.volume(level=lvl), d.twoNormSqLocal(), d.dot(d)), "inf": d.infNorm()}

    fab = tedist.LocalFabric(nranks)
    hs = [capi.Hierarchy(mesh, n, rank=r, nranks=nranks, placement=(16, 64, 1)) for r in range(nranks)]
    gs = [capi.GMG(h) for h in hs]
    for r, g in enumerate(gs):
        fab.attach(g, r)
    outs = fab.run(lambda r: per_rank(r, hs[r], gs[r], fab))
    sums = [sum(o["vals"][k] for o in outs) for k in range(4)]
    assert sums[0] == pytest.approx(want[0], rel=1e-13) and sums[1] == pytest.approx(want[1], rel=1e-13)
    assert sums[2] == want[2] and sums[3] == want[3]
    assert all(o["vals"] == (0.0, 0.0, 0.0, 0.0) for o in outs[1:])
    assert all(o["inf"] == outs[0]["inf"] for o in outs)


def test_schedule_check_catches_diverging_ranks():
    """A rank that would issue a different exchange sequence (here: rank 1 is handed two pre-sweeps) is an error on
    EVERY rank before anything is exchanged (te_gmg_verify_schedule: a collective the host calls at setup, and
    that the first te_vcycle with new options runs by itself) instead of a hang in the middle of the cycle."""
    n, nranks = 8, 2
    mesh = util.mesh("uniform", 2)
    fab = tedist.LocalFabric(nranks)
    fab.timeout = 30.0
    hs = [capi.Hierarchy(mesh, n, rank=r, nranks=nranks) for r in range(nranks)]
    gs = [capi.GMG(h) for h in hs]
    for r, g in enumerate(gs):
        fab.attach(g, r)

    def body(r):
        g = gs[r]
        f, u = g.new_vector(0), g.new_vector(0)
        f.set(1.0)
        good = g.default_opts(smoother=capi.SMOOTH_RBGS)
        g.cycle(good, f, u)  # new options: checked by te_vcycle itself, passes
        g.verify_schedule(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE, cycle_type=1))  # explicit: passes
        bad = g.default_opts(smoother=capi.SMOOTH_RBGS, pre_sweeps=2 if r == 1 else 1)
        try:
            g.verify_schedule(bad)
        except capi.TeError as e:
            return (e.code, str(e))
        return (0, "")

    out = fab.run(body)
    assert all(code == capi.TE_ESTATE for code, _ in out), out
    assert all("different exchange sequences" in msg for _, msg in out)


def test_watchdog_ends_a_rank_whose_peer_never_answers():
    """Two processes; rank 1 never calls the cycle. Rank 0's first exchange cannot complete: its watchdog thread must
    end the process with status 86 within TE_EXCHANGE_TIMEOUT instead of hanging (the launcher then takes the job down)."""
    import socket
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, TE_EXCHANGE_TIMEOUT="5", TE_NO_VERIFY="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(root, "tests", "mr_worker.py"), "--backend", "gloo",
           "--hang-rank", "1"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "te_hip watchdog" in r.stderr and "MR_WORKER_OK" not in r.stdout, r.stderr[-3000:]
    assert time.time() - t0 < 120


def test_watchdog_stays_quiet_while_the_host_runs_ahead():
    """A sync-free loop of exchanges that lasts three times TE_EXCHANGE_TIMEOUT: the host is always ahead of the GPU, so the
    newest exchange is never complete when the watchdog polls. Its deadline belongs to the oldest OUTSTANDING exchange
    (a ring of events), so the healthy run must survive (exit 0, not 86). Native RCCL back-end, the rank as its own peer."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from pressurepoissonsolver_amd import capi, dist as tedist\n"
            "H = capi.Hierarchy(capi.Mesh.uniform(3, 2), 32)\n"
            "g = capi.GMG(H)\n"
            "tedist.attach_rccl(g, None, 0, 1)\n"
            "n = capi.lib().te_gmg_watchdog_selftest(g.h, 6.0)\n"
            "assert n > 1000, n\n"  # (far more than the watchdog's ring of 64 slots: the host waits for room, nothing watched is dropped)
            "print('WD_OK', n)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TE_EXCHANGE_TIMEOUT="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WD_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("overlap_min", ["0", "0:2", None], ids=["overlapped", "interior-on-2nd-stream", "default"])
def test_two_processes_gloo(overlap_min):
    """Real processes + torch.distributed (gloo staging through host) on the single GPU: one attached callback
    serves the level-0 face exchange on the communication stream (overlapped apply, TE_OVERLAP_MIN=0) and on the
    solver stream (fused cycle) in the sequence apply, cycle, BiCGStab, apply; results equal the single-rank run."""
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.pop("TE_OVERLAP_MIN", None)
    env.pop("TE_OVERLAP_MODE", None)
    if overlap_min is not None:
        env["TE_OVERLAP_MIN"] = overlap_min.split(":")[0]
        if ":" in overlap_min:
            env["TE_OVERLAP_MODE"] = overlap_min.split(":")[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(root, "tests", "mr_worker.py"), "--backend", "gloo"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MR_WORKER_OK" in r.stdout


def test_two_processes_gloo_2d_reference_smoother():
    """the fused 2D block-Jacobi cycle between real processes (gloo): 64 patches of 64^2 on two ranks -- the post-sweep's neighbours
    send u + P e (k_pack_faces_prolong2d), the interface residual reads the neighbours' new edges; apply / cycle / BiCGStab / apply
    equal the single-rank run (mr_worker.py)"""
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(root, "tests", "mr_worker.py"), "--backend", "gloo", "--dim", "2", "--cells", "64", "--divides", "3", "--smoother", "patch_solve"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MR_WORKER_OK" in r.stdout


@pytest.mark.parametrize("n,divides,nproc,dim", [(8, 2, 2, 3), (32, 3, 2, 3), (32, 3, 4, 3), (64, 4, 2, 2)],
                         ids=["64x8^3", "512x32^3-fused-path", "512x32^3-4-processes", "2D-256x64^2"])
def test_two_processes_direct_store_transport(n, divides, nproc, dim):
    """te_gmg_use_push between real PROCESSES (two, and four: more peers per exchange, the blocks of a gather from three ranks;
    all on the one GPU of the box: hipIpcGetMemHandle / hipIpcOpenMemHandle
    mappings, fine-grained flags, the bounded wait kernel): the face exchanges and the in-place exchange of restricted blocks go
    by direct stores, everything else through the attached gloo back-end. te_gmg_autotune's own check (result identical to the
    other transport's after a cycle on different data) must pass, and apply / cycle / BiCGStab / apply equal the single-rank
    run as in test_two_processes_gloo. 32^3 patches: the fused default path (face layers sent from where they lie, no
    post-sweep exchange above the replicated level)."""
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, TE_PUSH_TIMEOUT="30")
    for k in ("TE_OVERLAP_MIN", "TE_OVERLAP_MODE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(root, "tests", "mr_worker.py"), "--backend", "gloo", "--push",
           "--cells", str(n), "--divides", str(divides), "--dim", str(dim)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MR_WORKER_OK" in r.stdout and "transport:" in r.stdout


def test_direct_store_transport_that_never_delivers_is_rejected_everywhere():
    """TE_PUSH_FAULT: the direct-store exchanges raise stale flags, i.e. their data never "arrives". The bounded waits give up
    (TE_PUSH_TIMEOUT = 2 s), te_gmg_autotune rejects the transport on both ranks alike, clears its error word -- the watchdog must
    not end a process that has gone back to the other transport -- and apply / cycle / BiCGStab / apply then equal the single-rank
    run through the attached back-end. What protects a job on a node where the mapping or the memory model does not hold."""
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, TE_PUSH_TIMEOUT="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(root, "tests", "mr_worker.py"), "--backend", "gloo", "--push", "--push-fault",
           "--cells", "8", "--divides", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "MR_WORKER_OK" in r.stdout and "REJECTED" in r.stdout


def run_inner(name, timeout_s="30"):
    """one test of this file in a process of its own (TE_DIRECT_STORE_INNER=1 enables it there)"""
    env = dict(os.environ, TE_DIRECT_STORE_INNER="1", TE_PUSH_TIMEOUT=timeout_s)
    env.pop("TE_OVERLAP_MIN", None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", name,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_direct_store_transport_virtual_ranks():
    """the same transport between two virtual ranks of ONE process (raw pointers instead of hipIpc mappings), in a process of its
    own: a process has few hardware queues, and a kernel that waits for a flag must never sit in the queue in front of the kernel
    that raises it -- true for a fresh process with two solvers (two ranks only), not for a test process that has created dozens of
    streams before (there the wait gives up after TE_PUSH_TIMEOUT and the watchdog ends the process, as it should)."""
    run_inner("direct_store_virtual_ranks_body")


def test_direct_store_transport_under_every_overlap_form():
    """(round 5 advisor) the protocol's checks (overrun, peer behind, sequence) rest on per-slot stream order: push e + 1 of a slot
    sits behind wait e. pushExchange runs on the communication stream or on the solver stream depending on how a level meets its
    exchange (in line / exchange under the interior patches / interior patches on the second stream): every form te_gmg_autotune
    can choose, and changes of form from one cycle to the next on the same slots, must leave push_failed() == 0 and the single-rank
    bits"""
    run_inner("direct_store_overlap_forms_body")


def test_direct_store_overrun_is_caught_by_the_receiver():
    """the no-credit argument as a check: a peer whose flag is TWO exchanges ahead (TE_PUSH_FAULT=overrun raises epoch + 2) cannot
    exist under the protocol -- k_push_wait must set the error word to PUSH_ERR_OVERRUN (2) instead of taking it for an arrival"""
    run_inner("direct_store_overrun_body", "5")


def test_direct_store_same_pid_other_process_is_not_taken_for_a_virtual_rank():
    """pids repeat across containers and nodes: a peer that publishes MY pid but another process nonce (TE_PUSH_FAULT=nonce) must
    be reached through hipIpcOpenMemHandle, never through its raw pointer -- here (the peer really is in this process) that either
    fails cleanly on all ranks together or maps the same memory; it must not be treated as an in-process rank by pid alone"""
    run_inner("direct_store_fake_nonce_body", "5")


@pytest.mark.skipif(os.environ.get("TE_DIRECT_STORE_INNER") != "1", reason="runs in a process of its own: test_direct_store_transport_virtual_ranks")
def test_direct_store_virtual_ranks_body():
    """switched on and off between cycles, W-cycle included (two gathers per cycle and level: the parities of the double buffers),
    all equal to the single-rank run bit for bit."""
    n = 8
    mesh = util.mesh("uniform", 3)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f = util.rand_vec(H1.cells(0), 11)
    nc = n ** 3
    want = {}
    for ct in (0, 1):
        d1 = g1.new_vector(0)
        g1.cycle(g1.default_opts(smoother=capi.SMOOTH_RBGS, cycle_type=ct), g1.new_vector(0, f), d1)
        want[ct] = d1.download()

    def per_rank(r, H, g, fab):
        df, du = g.new_vector(0, f.reshape(-1, nc)[H.l2g(0)].ravel()), g.new_vector(0)
        out = {}
        g.use_push(True)
        for rep in range(3):  # (parities 0, 1, 0)
            g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
            out[f"v{rep}"] = du.download()
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS, cycle_type=1), df, du)
        out["w"] = du.download()
        g.use_push(False)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
        out["off"] = du.download()
        g.use_push(True)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
        out["on"] = du.download()
        out["failed"] = g.push_failed()
        return out

    got = shard_run(mesh, n, 2, per_rank)
    assert not any(got["failed"])
    for k in ("v0", "v1", "v2", "off", "on"):
        assert np.array_equal(got[k], want[0]), k
    assert np.array_equal(got["w"], want[1])


@pytest.mark.skipif(os.environ.get("TE_DIRECT_STORE_INNER") != "1", reason="runs in a process of its own: test_direct_store_transport_under_every_overlap_form")
def test_direct_store_overlap_forms_body():
    n = 8
    mesh = util.mesh("uniform", 3)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f = util.rand_vec(H1.cells(0), 12)
    nc = n ** 3
    d1 = g1.new_vector(0)
    g1.cycle(g1.default_opts(smoother=capi.SMOOTH_RBGS), g1.new_vector(0, f), d1)
    want = d1.download()
    # (TE_OVERLAP_MIN, TE_OVERLAP_MODE): in line; exchange on the communication stream under the interior patches; interior patches on
    # the second stream, exchange on the solver stream -- and back and forth between them, so that consecutive exchanges of one slot
    # are issued on different streams
    forms = [("1000000", None), ("0", None), ("0", "2"), ("0", None), ("1000000", None), ("0", "2"), ("0", "2"), ("0", None)]

    def per_rank(r, H, g, fab):
        df, du = g.new_vector(0, f.reshape(-1, nc)[H.l2g(0)].ravel()), g.new_vector(0)
        out = {}
        g.use_push(True)
        for k, (omin, omode) in enumerate(forms):
            g.set_option("TE_OVERLAP_MIN", omin)
            g.set_option("TE_OVERLAP_MODE", omode)
            g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
            out[f"c{k}"] = du.download()
        out["failed"] = g.push_failed()
        g.use_push(False)
        return out

    got = shard_run(mesh, n, 2, per_rank)
    assert not any(got["failed"]), got["failed"]
    for k in range(len(forms)):
        assert np.array_equal(got[f"c{k}"], want), (k, forms[k])


@pytest.mark.skipif(os.environ.get("TE_DIRECT_STORE_INNER") != "1", reason="runs in a process of its own: test_direct_store_overrun_is_caught_by_the_receiver")
def test_direct_store_overrun_body():
    n = 8
    mesh = util.mesh("uniform", 3)

    def per_rank(r, H, g, fab):
        g.set_option("TE_PUSH_NONFATAL", "1")  # (the test reads the error word itself; the watchdog must not end the process)
        df, du = g.new_vector(0, util.rand_vec(H.sizes(0)[0] * n ** 3, 5 + r)), g.new_vector(0)
        g.use_push(True)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
        g.sync()
        ok_before = g.push_failed()
        fab.barrier.wait()
        g.set_option("TE_PUSH_FAULT", "overrun")
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
        g.sync()
        return {"before": ok_before, "after": g.push_failed()}

    got = shard_run(mesh, n, 2, per_rank)
    assert got["before"] == [0, 0]
    assert got["after"] == [2, 2], got  # PUSH_ERR_OVERRUN on both ranks: each saw the other's flag beyond epoch + 1


@pytest.mark.skipif(os.environ.get("TE_DIRECT_STORE_INNER") != "1", reason="runs in a process of its own")
def test_direct_store_fake_nonce_body():
    n = 8
    mesh = util.mesh("uniform", 3)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f = util.rand_vec(H1.cells(0), 11)
    d1 = g1.new_vector(0)
    g1.cycle(g1.default_opts(smoother=capi.SMOOTH_RBGS), g1.new_vector(0, f), d1)
    want = d1.download()
    nc = n ** 3

    def per_rank(r, H, g, fab):
        g.set_option("TE_PUSH_FAULT", "nonce")
        g.set_option("TE_PUSH_NONFATAL", "1")
        df, du = g.new_vector(0, f.reshape(-1, nc)[H.l2g(0)].ravel()), g.new_vector(0)
        try:
            g.use_push(True)
            msg = ""
        except capi.TeError as e:
            msg = str(e)
        g.set_option("TE_PUSH_FAULT", None)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)  # through the mapping if there is one, else through the fabric
        return {"msg": msg, "u": du.download(), "failed": g.push_failed()}

    got = shard_run(mesh, n, 2, per_rank)
    assert (got["msg"][0] == "") == (got["msg"][1] == ""), got["msg"]  # set up on both ranks, or refused on both
    if got["msg"][0]:
        assert all("hipIpcOpenMemHandle" in m or "te_gmg_use_push" in m for m in got["msg"]), got["msg"]  # (went for a mapping, not for the raw pointer)
    assert got["failed"] == [0, 0]
    assert np.array_equal(got["u"], want)


def test_direct_store_setup_failure_is_collective_and_leaves_nothing_behind():
    """one rank's set-up of the direct-store transport fails (TE_PUSH_FAULT=setup:1): BOTH ranks come back with an error -- the
    failing rank still takes part in the directory reductions, so nobody waits for it until a watchdog fires -- and what was
    allocated or mapped is released: the next attempt, without the fault, succeeds and computes the single-rank result"""
    run_inner("direct_store_setup_failure_body", "5")


@pytest.mark.skipif(os.environ.get("TE_DIRECT_STORE_INNER") != "1", reason="runs in a process of its own")
def test_direct_store_setup_failure_body():
    n = 8
    mesh = util.mesh("uniform", 3)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f = util.rand_vec(H1.cells(0), 11)
    d1 = g1.new_vector(0)
    g1.cycle(g1.default_opts(smoother=capi.SMOOTH_RBGS), g1.new_vector(0, f), d1)
    want = d1.download()
    nc = n ** 3

    def per_rank(r, H, g, fab):
        g.set_option("TE_PUSH_FAULT", "setup:1")
        try:
            g.use_push(True)
            msg = ""
        except capi.TeError as e:
            msg = str(e)
        g.set_option("TE_PUSH_FAULT", None)
        fab.barrier.wait()
        g.use_push(True)  # a clean second attempt
        df, du = g.new_vector(0, f.reshape(-1, nc)[H.l2g(0)].ravel()), g.new_vector(0)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
        return {"msg": msg, "u": du.download(), "failed": g.push_failed()}

    got = shard_run(mesh, n, 2, per_rank)
    assert "injected set-up failure" in got["msg"][1] and "rank 1 could not set up" in got["msg"][0], got["msg"]
    assert got["failed"] == [0, 0]
    assert np.array_equal(got["u"], want)


def test_checksum_is_independent_of_order_and_partition():
    """te_vec_checksum = the sum modulo 2^64 of the values' bit patterns: equal to numpy's on the downloaded vector, unchanged by
    a permutation of the patches, and the sum of the ranks' parts of a sharded cycle equals the single-rank cycle's (the equality
    bench.py prints as u_checksum_after_timed_region for N = 1, 2, 4, 8)"""
    n = 8
    mesh = util.mesh("uniform", 3)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f = util.rand_vec(H1.cells(0), 21)
    f[7] = -0.0  # (the checksum sees bits: -0.0 != +0.0)
    nc = n ** 3
    M = (1 << 64) - 1

    def np_sum(a):
        return int(np.add.reduce(np.ascontiguousarray(a).view(np.uint64), dtype=np.uint64))

    df, du = g1.new_vector(0, f), g1.new_vector(0)
    assert df.checksumLocal() == np_sum(f)
    perm = np.random.default_rng(3).permutation(H1.sizes(0)[0])
    assert g1.new_vector(0, f.reshape(-1, nc)[perm].ravel()).checksumLocal() == np_sum(f)
    g1.cycle(g1.default_opts(smoother=capi.SMOOTH_RBGS), df, du)
    want = du.checksumLocal()
    assert want == np_sum(du.download())

    def per_rank(r, H, g, fab):
        lf, lu = g.new_vector(0, f.reshape(-1, nc)[H.l2g(0)].ravel()), g.new_vector(0)
        g.cycle(g.default_opts(smoother=capi.SMOOTH_RBGS), lf, lu)
        coarse = g.new_vector(H.num_levels - 1)  # a level that lives on every rank counts once
        coarse.set(1.5)
        return {"c": lu.checksumLocal(), "coarse": coarse.checksumLocal(), "repl": H.replicated(H.num_levels - 1)}

    for nranks in (2, 4):
        got = shard_run(mesh, n, nranks, per_rank)
        assert sum(got["c"]) & M == want, nranks
        if all(got["repl"]):
            assert [c != 0 for c in got["coarse"]] == [True] + [False] * (nranks - 1)


def test_native_rccl_backend_selftest():
    """The library's own RCCL back-end (dlopen of the process's librccl.so, ncclCommInitRank, one
    ncclGroup of ncclRecv + ncclSend on the solver stream) moving data with this rank as its own peer:
    everything except a second GPU."""
    import ctypes as C
    H = capi.Hierarchy(util.mesh("uniform", 1), 8)
    g = capi.GMG(H)
    tedist.attach_rccl(g, None, 0, 1)
    capi.check(capi.lib().te_gmg_exchange_selftest(g.h, 1000))


@pytest.mark.parametrize("nranks", [2, 4])
def test_sharded_2d_fused_block_jacobi_cycle(nranks, monkeypatch):
    """the reference smoother's fused 2D cycle cut by rank boundaries (64 patches of 64^2): the post-sweep's neighbours on other ranks
    send their facing values of u + P e, the interface residual reads the neighbours' new edges from ghost slots: sharded == single
    rank bit for bit for every fuse setting"""
    monkeypatch.setenv("TE_AGGLOMERATE", "4")
    n = 64
    mesh = util.mesh("uniform", 3, 2)
    H1 = capi.Hierarchy(mesh, n)
    g1 = capi.GMG(H1)
    f = util.rand_vec(H1.cells(0), 13)
    nc = n * n
    want = {}
    for fuse in (1, 3):
        df, du = g1.new_vector(0, f), g1.new_vector(0)
        g1.cycle(g1.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE, fuse=fuse), df, du)
        want[fuse] = du.download()

    def per_rank(r, H, g, fab):
        idx = H.l2g(0)
        out = {}
        for fuse in (1, 3):
            df, du = g.new_vector(0, f.reshape(-1, nc)[idx].ravel()), g.new_vector(0)
            g.cycle(g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE, fuse=fuse), df, du)
            out[f"u{fuse}"] = du.download()
        return out

    got = shard_run(mesh, n, nranks, per_rank, dim=2)
    assert np.array_equal(got["u1"], want[1]) and np.array_equal(got["u3"], want[3])
