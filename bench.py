#!/usr/bin/env python3
"""V-cycle lattice-site updates per second on the 512^3 uniform 3D Poisson problem
(BASELINE.json metric), MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one GMG V(1,1) cycle (GMG/Cycle.h:116-126 semantics: zero u, pre-smooth, residual,
restrict, recurse, prolong-add, post-smooth; default options: te_cycle_opts_default, fuse = 3) over the whole 512^3 grid, 16^3 patches of 32^3
(apps/3d/steady -n 32 --mesh 4uni.bin --divide 1). Inputs are resident in HBM before the timed
region. N > 1: one rank per GPU (torch.distributed, backend nccl == RCCL); the same 512^3
problem is sharded by contiguous Morton ranges of patches (strong scaling).

The JSON line carries `roofline` for the dominant kernel (HIP-event timed on the solver stream
inside the timed region) and `cpu_baseline` (the CPU restatement of the reference algorithm,
oracle/, timed on this host's cores on the same workload, bounded to ~30 s; rank 0, N = 1 only).
`roofline.traffic` (HBM bytes per launch from rocprofv3 PMC passes) is only reported when
profiles/traffic.json was measured on exactly the kernel sources this run compiled (a hash of
csrc/ travels with it); otherwise it is null.
"""
import argparse
import hashlib
import json
import math
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable

# algorithmic (compulsory) HBM bytes per lattice site per launch, fp64 (DESIGN.md "Kernels")
ALG_BYTES = {
    "stencil_rbgs": 24.0,    # read u, f; write u
    "stencil_jacobi": 24.0,
    "stencil_resid": 24.0,   # read u, f; write r
    "stencil_apply": 16.0,
    "restrict": 9.0,         # read fine (8), write coarse (8/8)
    "prolong_add": 17.0,     # read+write fine (16), read coarse (8/8)
    "vecop": 8.0,
    "patch_rhs": 24.0,
    "dst_axis": 16.0,
    "resid_restrict": 17.0,      # read u, f; write coarse f (8/8): residual never stored
    "stencil_rbgs_zero": 16.0,   # first sweep from a zero guess: read f, write u
    "stencil_rbgs_prolong": 25.0,  # post-sweep on u + P(coarse): read u, f, coarse (8/8); write u
    # single-pass exact patch solve. Inside the default cycle (fuse = 3) the launches of this class are the post-sweeps:
    # read f (8) and the interface terms on the six face layers (1.5), write u (8) = 17.5; a zero-guess sweep that stores u
    # moves 16. The pre-sweep that stores face layers only is a class of its own:
    "patch_solve_mfma": 17.5,
    "patch_solve_mfma_faces": 9.5,   # read f (8), write the six face layers (6/32 of a vector: 1.5)
    "rbgs_zero_resid_restrict": 17.0,  # fuse = 2: sweep from zero + residual + restriction: read f, write u and 1/8
    "restrict_fixup": 12.0,      # per face cell: read the neighbour's value (8), update a coarse cell per 2x2 (16/4)
    # fuse = 3 (default): the iterate between the two sweeps exists only as its six face layers (6/32 of a vector)
    "rbgs_zero_resid_restrict_faces": 10.5,  # read f (8); write the face layers (1.5) and coarse f (1)
    "rbgs_resweep_prolong": 18.5,            # read f (8), coarse (1), neighbours' face layers (1.5); write u (8)
    # the same two kernels on a level whose right-hand side carries exported x-face ghost terms (4/32 of a vector: + 1)
    "rbgs_zero_resid_restrict_faces_fcorr": 11.5,
    "rbgs_resweep_prolong_fcorr": 19.5,
    "fcorr_gather": 16.0,                    # per entry of the 2x2 face sums: read 8, write 8 (cells = entries)
    # levels with few patches run other instantiations, timed as classes of their own (one class = one kernel
    # symbol for the large levels); they only become "dominant" for small problems (--size 256):
    "stencil_rbgs_slabs": 20.5,  # z-slab RB-GS: a V(1,1) launches as many zero-guess (16) as fused-prolong (25) sweeps
    "stencil_slabs": 17.0,       # z-slab stencil kernel: residual+restrict is its only use in the fused cycle
    "patch_solve_3pass": 16.0,   # three-pass patch solve: read 8 + write 8 per launch
    # the small classes (face layers, blocks between ranks): per cell they touch -- read 8, write 8
    "cf_ghost": 16.0, "pack": 16.0, "reduce": 8.0,
    # te_bicgstab's own passes (BiCGStab.h:71-104 between the operator applications and the cycles)
    "bicg_update": 72.0,         # read x, resid, M p, M s, A p, A s, rhat; write x, resid; two dot products on the way
    "bicg_s": 24.0, "bicg_p": 32.0,   # the stand-alone s and p statements (paths where no cycle kernel forms them)
    # TE_SMOOTH_PATCH_BCGS (2D): compute-resident Krylov solve per patch; HBM sees the right-hand side and u in, u out, once per sweep
    "patch_bcgs": 24.0,
    "stencil_apply_dot": 24.0,   # f = A u with one or two dot products summed while f is in registers: read u, the operand; write f
    "exchange": 0.0,             # RCCL / callback time on the solver stream: no HBM pass of this library's own
}
# fp64 matrix-core work of the exact patch solve per lattice site (DESIGN.md "The reference smoother"): k_ps_sym runs 3072
# v_mfma_f64_16x16x4_f64 per 32^3 patch (half-size transforms), 2048 flop each
MFMA_FLOPS_PER_SITE = {"patch_solve_mfma": 3072 * 2048 / 32 ** 3, "patch_solve_mfma_faces": 3072 * 2048 / 32 ** 3}
MFMA_F64_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: dense fp64 matrix peak


def vcycle_alg_bytes_per_finest_cell(levels_cells, fused=False):
    """per level V(1,1) 3D unfused = 2*24 + 24 + 9 + 1 + 17 = 99 B/cell,
    coarsest level 24 B/cell; summed over levels per finest cell."""
    total = 0.0
    for i, c in enumerate(levels_cells):
        total += (24.0 if i == len(levels_cells) - 1 else 99.0) * c
    return total / levels_cells[0]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=512, help="cells per axis of the uniform grid (multiple of 32)")
    ap.add_argument("--smoother", default="rbgs", choices=["rbgs", "jacobi", "patch_solve"])
    ap.add_argument("--dim", type=int, default=3, choices=[2, 3], help="2: the 2D twin (config C5: --dim 2 --size 4096 --patch 64)")
    ap.add_argument("--patch", type=int, default=32, help="cells per axis per patch")
    ap.add_argument("--mesh", default=None, help="octree file instead of the uniform grid (config C4: tests/golden/2refine.bin "
                                                 "--divide 2 or 3); --size is ignored")
    ap.add_argument("--divide", type=int, default=0, help="refineLeaves passes over --mesh (apps/3d/steady --divide)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the reference-smoother block (secondary.reference_smoother)")
    ap.add_argument("--cpu-size", type=int, default=0, help="cells per axis of the CPU baseline (default: --size)")
    return ap.parse_args()


def workload_key(a):
    """what profiles/traffic.json keys a PMC figure by, next to the kernel class and the rank count: a figure is only ever
    quoted for the workload it was measured on"""
    grid = f"{os.path.basename(a.mesh)}+{a.divide}" if a.mesh else f"u{a.size}"
    return f"{a.dim}d:{grid}:p{a.patch}:{a.smoother}"


def kernel_sources_sha():
    """hash of the kernel sources this run compiled: profiles/traffic.json carries the same, so a PMC figure is
    never quoted for kernels it was not measured on"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "pressurepoissonsolver_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(size, dim, n, gpu_smoother):
    """CPU restatement of the reference V-cycle (oracle/, pinned to the reference's golden vectors) timed on this
    host beside the GPU number, on the SAME workload: `size`^dim uniform, V(1,1), both smoothers (the reference's
    block-Jacobi patch solves and the patch-local RB-GS the GPU line uses), with all the cores this job may use
    (the reference's `mpirun -np cores` analogue: OpenMP over patches) and with 1 thread (= one reference MPI rank;
    on a 1/8-size sample, 2x smaller per axis, because one 512^3 cycle takes ~25 s on one core). ~30 s in total.
    `value` is the all-cores row of the smoother the GPU line ran; the reference-smoother row is quoted beside it."""
    from oracle import oracle as orc
    from pressurepoissonsolver_amd import capi, problems
    ncpu = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = ncpu
    # cores this job may really use: the cgroup CPU quota when there is one (the GPU boxes of this pool grant a share
    # of the host per GPU), else the affinity mask, capped at the physical cores (SMT siblings do not help a
    # bandwidth-bound stencil); TE_CPU_THREADS overrides
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    phys = max(1, ncpu // 2)
    threads_all = int(os.environ.get("TE_CPU_THREADS", min(avail, quota or avail, phys)))
    names = {0: "patch_solve", 2: "rbgs"}

    def run(sz, threads, smoother, budget, fft=False):
        m = capi.Mesh.uniform(dim, int(round(np.log2(sz // n))))
        H = capi.Hierarchy(m, n)
        levels = orc.levels_from_hierarchy(H)
        f = problems.random_rhs(H.tables(0)["id"], n ** dim)
        o = orc.cycle_opts(smoother=smoother)
        orc.set_threads(threads)
        orc.set_fast_transforms(fft)
        try:
            t0 = time.time()
            orc.cycle(levels, o, f)  # warm (page faults of the level scratch)
            warm = time.time() - t0
            ts = []
            while len(ts) < 2 or (sum(ts) + warm < budget and len(ts) < 20):
                t0 = time.time()
                orc.cycle(levels, o, f)
                ts.append(time.time() - t0)
        finally:
            orc.set_fast_transforms(False)
        dt = float(np.median(ts))
        return {"size": f"{sz}^{dim}", "threads": threads, "smoother": names[smoother] + ("_fft" if fft else ""),
                "ms_per_cycle": dt * 1e3, "updates_per_s": levels[0].size / dt, "cycles": len(ts)}

    small = max(size // 2, 2 * n)
    rows = [run(size, threads_all, 0, 8.0), run(size, threads_all, 2, 6.0), run(small, 1, 0, 6.0), run(small, 1, 2, 5.0),
            run(size, threads_all, 0, 8.0, fft=True)]
    # the reference has two exact patch solvers: the dense transforms of DftPatchSolver.h:295-347 (row `patch_solve`: the parity
    # oracle) and FFTW's O(n log n) r2r transforms, its DEFAULT (--patch_solver fftw, apps/3d/steady.cpp:126, FftwPatchSolver.h:93-206;
    # row `patch_solve_fft`: the oracle's own radix-2 FFT in their place): the reference-smoother figure is the FASTER of the two
    ref_row = min((rows[0], rows[4]), key=lambda r_: r_["ms_per_cycle"])
    match = {"patch_solve": ref_row, "rbgs": rows[1]}.get(gpu_smoother, ref_row)
    label = {"patch_solve": "the reference's block-Jacobi patch-solve smoother (dense transforms, DftPatchSolver's form)",
             "patch_solve_fft": "the reference's block-Jacobi patch-solve smoother (O(n log n) transforms, FftwPatchSolver's form: the reference's default)",
             "rbgs": "the patch-local RB-GS smoother of the GPU line"}
    return {"value": match["updates_per_s"], "unit": "lattice-site updates/s", "cores": threads_all, "kind": "port",
            "smoother": match["smoother"],
            "sample": f"{size}^{dim} uniform (the benchmarked workload), V(1,1), {label.get(match['smoother'])}, "
                      f"median of {match['cycles']} cycles, {threads_all} OpenMP threads",
            "ms_per_step": match["ms_per_cycle"],
            "reference_smoother_value": ref_row["updates_per_s"], "reference_smoother_ms_per_step": ref_row["ms_per_cycle"],
            "reference_smoother_row": ref_row["smoother"],
            "cpu_model": cpu_model(), "nproc": ncpu, "cpus_available_to_job": avail, "cgroup_cpu_quota": quota,
            "runs": rows,
            "note": "CPU restatement of the reference algorithm (oracle/te_oracle.cpp), not the reference binary "
                    "(PETSc/FFTW/Zoltan are absent); `value` = the row whose smoother matches the GPU line, "
                    "`reference_smoother_value` = the reference's own smoother (what secondary.reference_smoother runs on the GPU), the faster of "
                    "its two patch solvers' restatements: runs[smoother = patch_solve] = dense transforms (PatchSolvers/DftPatchSolver.h:295-347, one "
                    "product per line; the parity oracle), runs[smoother = patch_solve_fft] = O(n log n) transforms (PatchSolvers/FftwPatchSolver.h:93-206, "
                    "the reference's default --patch_solver fftw; here the oracle's own radix-2 FFT, not FFTW's codelets); "
                    "1-thread rows = one reference MPI rank, on a 2x-per-axis smaller grid"}


def self_launch(ngpus, json_out):
    """`python bench.py --gpus N` with no launcher environment (the reference's drivers are self-contained under
    mpirun, apps/3d/steady.cpp:74-78): start one rank per GPU through torch.distributed.run as a CHILD process -- never an
    exec, and before anything in this process has initialised the GPU -- pass rank 0's single JSON line through and return
    the launcher's exit status (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    print("+ " + " ".join(cmd), file=sys.stderr, flush=True)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)  # (stderr is inherited)
    lines = [ln for ln in p.stdout.splitlines() if ln.lstrip().startswith("{")]
    for ln in p.stdout.splitlines():
        if ln not in lines and ln.strip():
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], file=json_out, flush=True)
    if p.returncode == 0 and not lines:
        print("bench.py: the ranks exited without a JSON line", file=sys.stderr)
        return 1
    return p.returncode


PROVISIONAL = "#PROVISIONAL "  # worker -> supervisor: the line so far (the headline is measured; a risky secondary block follows)
# the line as far as it is built (rank 0, once the headline is measured): whatever ends run() afterwards -- ANY exception, not only the
# library's -- main() prints THIS with an `error` entry, never a value-null line in its place
HEADLINE = {"out": None, "block": None}
DIRECT_STORE_BLOCK = "secondary.direct_store (the direct-store transport between the ranks' GPUs)"


def metric_name(a):
    if a.size == 512 and a.dim == 3 and not a.mesh and a.patch == 32:
        return "V-cycle lattice-site updates/sec, 512^3 3D Poisson"
    if a.mesh:
        return f"V-cycle lattice-site updates/sec, {os.path.basename(a.mesh)} --divide {a.divide}, {a.dim}D Poisson"
    return f"V-cycle lattice-site updates/sec, {a.size}^{a.dim} {a.dim}D Poisson"


def error_line(a, world, msg):
    """the one JSON line of a run that produced no number: same keys, value null, the reason in `error`"""
    return json.dumps({"metric": metric_name(a), "value": None, "unit": "lattice-site updates/s", "n_gpus": world, "steps": a.steps,
                       "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                       "dtype": "f64", "data": "synthetic", "config": {"workload_key": workload_key(a)}, "error": msg})


def supervise(a, json_out):
    """N > 1 under a launcher (torchrun sets WORLD_SIZE): the process the launcher started never touches the GPU -- it starts the
    real rank as a CHILD (same arguments, TE_BENCH_WORKER=1), passes its stderr through and reads its stdout. Whatever ends the
    worker -- an exception, the exchange watchdog's exit 86, a GPU fault's abort, the launcher's SIGTERM after another rank died --
    rank 0's supervisor still prints ONE JSON line: the worker's final line; or the line the worker had finished before its last
    optional block (PROVISIONAL) with `error` naming how the worker ended; or an error line with value null. The exit status is
    the worker's."""
    import signal
    import subprocess
    import threading
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    env = dict(os.environ, TE_BENCH_WORKER="1")
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE, text=True)
    ended_by = []

    def forward(sig, _frm):
        ended_by.append(signal.Signals(sig).name)
        try:
            p.send_signal(signal.SIGTERM)
        except ProcessLookupError:
            return
        threading.Timer(10.0, lambda: p.poll() is None and p.kill()).start()  # (a worker stuck in a device call)

    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, forward)
    final, prov = None, None
    for ln in p.stdout:
        t = ln.strip()
        if t.startswith(PROVISIONAL):
            prov = t[len(PROVISIONAL):]
        elif t.startswith("{"):
            final = t
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = p.wait()
    how = (f"the launcher sent {ended_by[0]} (another rank failed)" if ended_by else
           "the exchange watchdog ended the worker (exit 86: an exchange never completed)" if rc == 86 else
           f"worker killed by signal {-rc}" if rc < 0 else f"worker exit status {rc}")
    if rank == 0:
        final_null = False
        if final is not None and prov is not None:
            try:  # (a worker whose last words are a value-null error line, behind a provisional line that carries the measured headline)
                fd = json.loads(final)
                final_null = fd.get("value") is None
                if final_null:
                    how = f"{fd.get('error', how)} ({how})"
            except ValueError:
                final_null = False
        if final is not None and not final_null:
            print(final, file=json_out, flush=True)
        elif prov is not None:
            d = json.loads(prov)
            d["error"] = f"the headline was measured; then, inside {d.pop('next_block', 'an optional block')}: {how}"
            print(json.dumps(d), file=json_out, flush=True)
        else:
            print(error_line(a, world, how + "; see stderr"), file=json_out, flush=True)
    return rc


# classes whose launches on a REFINED level move more than on a uniform one: a patch that does not coarsen (AvgRstr.h:103-107,
# DrctIntp.h:107-111 copy it through) reads its correction, resp. writes its residual, at full size -- 8 B per site where an
# octant child moves 1 (its eighth of a coarse patch): + 7 B per site of such patches
COPY_THROUGH_CLASSES = ("rbgs_resweep_prolong", "rbgs_zero_resid_restrict_faces", "stencil_rbgs_prolong", "rbgs_zero_resid_restrict",
                        "resid_restrict", "restrict", "prolong_add")


def roofline_of(name, st, tkey=None, world=1, copy_sites_per_cycle=0, cycles=1):
    """the contract's `roofline` object for one kernel class from its HIP-event rows of the timed region.
    copy_sites_per_cycle: lattice sites of this rank's patches that copy through to the next level on the levels this class
    runs on (refined meshes; 0 on a uniform grid), `cycles`: cycles the rows cover"""
    avg_ms = st["ms"] / st["calls"]
    sites = st["cells"] / st["calls"]
    alg = ALG_BYTES.get(name, 24.0)
    if name in COPY_THROUGH_CLASSES and copy_sites_per_cycle > 0 and st["cells"] > 0:
        alg = alg + 7.0 * copy_sites_per_cycle * cycles / st["cells"]  # the average over the class's launches
    achieved = alg * sites / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if tkey is not None and os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            if tj.get("kernel_sources_sha") == kernel_sources_sha():  # measured on exactly these kernels ...
                traffic = tj.get(f"{name}:{tkey}:{world}")              # ... and on exactly this workload
                traffic_src = tj.get("commit") if traffic is not None else None
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_measured_at_commit": traffic_src,
            "avg_launch_ms": avg_ms, "launches": st["calls"], "alg_bytes_per_site": alg,
            "alg_bytes_per_site_uniform_level": ALG_BYTES.get(name, 24.0)}


def roofline_mfma_of(name, st, dim=3, n=32):
    """second roof of the exact patch solve ('report it against both roofs'): its transform flops on the fp64
    matrix cores: the flops the kernel EXECUTES. 2D, n = 64: four half-size products per transform pair (k_patch_solve2d_sym:
    128 v_mfma_f64_16x16x4 per patch and stage pair... 4 * n flops per site; patches with a mixed Dirichlet / Neumann axis run the
    full products, twice that, and are counted at the half-size figure), 8 * n with TE_2D_NO_SYM (k_patch_solve2d_mfma)"""
    if name not in MFMA_FLOPS_PER_SITE:
        return None
    per_site = MFMA_FLOPS_PER_SITE[name] if dim == 3 else ((8.0 if os.environ.get("TE_2D_NO_SYM") else 4.0) * n)
    avg_ms = st["ms"] / st["calls"]
    tf = per_site * st["cells"] / st["calls"] / (avg_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": name, "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / MFMA_F64_PEAK_TFLOPS, "avg_launch_ms": avg_ms, "launches": st["calls"],
            "flops_per_site": per_site}


def main():
    a = parse()
    # stdout carries exactly one JSON line: everything else that libraries print there (e.g. Gloo's connection
    # banner from C++) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: this process has not touched the GPU (torch is not imported yet); it starts
        # the N ranks as children and relays rank 0's JSON line and the launcher's exit status
        raise SystemExit(self_launch(a.gpus, json_out))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was started with WORLD_SIZE={world}: one rank per GPU")
    if world > 1 and os.environ.get("TE_BENCH_WORKER") is None and os.environ.get("TE_BENCH_INPROC") is None:
        raise SystemExit(supervise(a, json_out))  # (this process stays off the GPU; the rank itself runs as its child)
    die = os.environ.get("TE_BENCH_TEST_DIE")  # test hook (tests/test_bench_launch.py): a worker that ends like a watchdog exit
    if die is not None and os.environ.get("TE_BENCH_WORKER") is not None:
        if die in ("after-provisional", "raise-after-provisional") and rank == 0:
            print(PROVISIONAL + json.dumps({"metric": metric_name(a), "value": 1.0, "n_gpus": world, "next_block": "a test block"}), file=json_out, flush=True)
        if die == "killed-in-direct-store":  # (a GPU fault's abort / the kernel's OOM killer inside the one block that talks to other GPUs)
            if rank == 0:
                print(PROVISIONAL + json.dumps({"metric": metric_name(a), "value": 1.0, "n_gpus": world, "next_block": DIRECT_STORE_BLOCK}), file=json_out, flush=True)
            import signal
            os.kill(os.getpid(), signal.SIGKILL)
        if not die.startswith("raise-"):
            os._exit(86)
    try:
        if die == "raise-after-provisional":  # (a worker that ends in an exception which is not the library's, behind its provisional line)
            raise ValueError("max() arg is an empty sequence")
        if die == "raise-in-optional-block":  # (in-process: the headline is built, then something that is not a TeError is raised)
            HEADLINE["out"] = {"metric": metric_name(a), "value": 1.0, "n_gpus": world}
            HEADLINE["block"] = "a test block"
            raise ValueError("max() arg is an empty sequence")
        run(a, json_out, rank, world, local_rank)
    except BaseException as e:  # noqa: BLE001 -- every failure path ends in a JSON line (rank 0; N > 1: the supervisor's when nothing is left to say here)
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        if rank == 0:
            if HEADLINE["out"] is not None:  # the headline was measured: it keeps its line, the failure goes beside it
                d = dict(HEADLINE["out"])
                d["error"] = f"the headline was measured; then, inside {HEADLINE['block'] or 'an optional block'}: {type(e).__name__}: {e}"
                print(json.dumps(d), file=json_out, flush=True)
            else:
                print(error_line(a, world, f"{type(e).__name__}: {e}"), file=json_out, flush=True)
        raise


def run(a, json_out, rank, world, local_rank):
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if world > 1:
        # a wedged exchange must end the run (watchdog: exit 86, the supervisor prints the error line) well inside any driver's
        # time limit -- the library's own default is 300 s
        os.environ.setdefault("TE_EXCHANGE_TIMEOUT", "60")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("TE_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of N > 1 on a 1-GPU box only
    if backend == "gloo":
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from pressurepoissonsolver_amd import build
    if rank == 0 or world == 1:
        build.build_hip()
    if dist is not None:
        dist.barrier()
    from pressurepoissonsolver_amd import capi, problems  # noqa: F401
    from pressurepoissonsolver_amd import dist as tedist

    n = a.patch
    t_setup0 = time.perf_counter()
    if a.mesh:
        mesh = capi.Mesh.read(a.mesh, a.dim)
        for _ in range(a.divide):
            mesh.refine_leaves()
    else:
        assert a.size % n == 0 and (a.size // n) & (a.size // n - 1) == 0, "--size must be patch * 2^k"
        mesh = capi.Mesh.uniform(a.dim, int(round(np.log2(a.size // n))))
    t_setup1 = time.perf_counter()
    setup_ms = {"mesh": (t_setup1 - t_setup0) * 1e3}
    smoothers = {"rbgs": capi.SMOOTH_RBGS, "jacobi": capi.SMOOTH_JACOBI, "patch_solve": capi.SMOOTH_PATCH_SOLVE}

    def make(placement):
        """hierarchy + solver + transport for one placement of the small levels ((agglomerate, agglomerate_max, replicate),
        None = the environment's / the defaults)"""
        t0 = time.perf_counter()
        H = capi.Hierarchy(mesh, n, rank=rank, nranks=world, placement=placement)  # te_hier_build: level extraction + partition
        t1 = time.perf_counter()
        g = capi.GMG(H, device=local_rank)                  # te_gmg_create: tables, plans, scratch (synchronised on return)
        t2 = time.perf_counter()
        setup_ms.update({"te_hier_build": (t1 - t0) * 1e3, "te_gmg_create": (t2 - t1) * 1e3,
                         # (te_gmg_setup_ms; the FIRST solver of a process pays the HIP runtime's start and the code-object load in context_streams)
                         "te_gmg_create_parts": g.setup_ms()})
        name = "none"
        if world > 1:
            # RCCL point-to-point issued by the native library itself (no Python per exchange); the
            # torch.distributed callback is the fallback (and the only choice for the gloo rehearsal)
            name = "torch.distributed"
            want = os.environ.get("TE_EXCHANGE", "rccl" if backend == "nccl" else "torch")
            ok = 0
            if want == "rccl":
                try:
                    tedist.attach_rccl(g, dist, rank, world)
                    ok = 1
                except Exception as e:  # noqa: BLE001
                    print(f"[rank {rank}] native RCCL exchange unavailable ({e}); using torch.distributed", file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # all ranks must use the same back-end
            if int(flag.item()) == 1:
                name = "rccl (native ncclSend/ncclRecv groups)"
            else:
                tedist.attach(g, dist)
        return H, g, name

    # N > 1: the switches that decide a sharded cycle's fixed cost -- where the small levels live (te_hier_build_placed) and
    # how the sweeps meet their face exchanges (te_gmg_autotune) -- are chosen HERE, on the communicator the job runs on:
    # every candidate is timed for a few cycles, the maximum over the ranks decides, the fastest is kept. All candidates give
    # the same results (bit for bit; tests/test_gpu_multirank*.py), so the choice changes no number but the time.
    autotune = None
    if world > 1 and os.environ.get("TE_BENCH_NO_AUTOTUNE") is None:
        pinned = any(os.environ.get(k) is not None for k in ("TE_AGGLOMERATE", "TE_AGGLOMERATE_MAX", "TE_REPLICATE"))
        cands = [("environment", None)] if pinned else (
            [("gathered<=64-on-every-rank", (64, 64, 1)), ("gathered<16/rank-on-every-rank", (16, 64, 1)), ("gathered<4/rank-on-every-rank", (4, 64, 1)),
             ("gathered<=64-on-rank-0", (64, 64, 0)), ("never-gathered", (0, 64, 0))] if a.dim == 3
            else [("gathered<=64-on-every-rank", (64, 64, 1)), ("gathered<16/rank-on-every-rank", (16, 64, 1)),
                  ("gathered<=64-on-rank-0", (64, 64, 0)), ("gathered<16/rank-on-rank-0", (16, 64, 0)), ("never-gathered", (0, 64, 0))])
        tried, best = [], None
        # (the headline runs on the transport `north_star` names -- RCCL point-to-point groups issued by the library; the
        # direct-store transport is measured afterwards, as secondary.direct_store, never inside the headline's choice)
        for cname, pl in cands:
            Hc, gc, bname = make(pl)
            ms, rep = gc.autotune(gc.default_opts(smoother=smoothers[a.smoother]), reps=10)
            tried.append({"placement": cname, "agglomerate/max/replicate": list(Hc.placement()), "ms_per_cycle_max_over_ranks": ms, "overlap": rep})
            if best is None or ms < best[0]:
                best = (ms, Hc, gc, bname, cname)
            else:
                del gc, Hc
        _, H, g, exchange_backend, chosen = best
        autotune = {"chosen_placement": chosen, "candidates": tried, "reps": 10,
                    "note": "timed on this run's communicator before the measured region; maximum over the ranks; identical results for every candidate"}
    else:
        H, g, exchange_backend = make(None)
    opts = g.default_opts(smoother=smoothers[a.smoother])

    f = g.new_vector(0)
    g.init_problem(f, None, problem=capi.PROBLEM_RANDOM)  # U(-1,1) splitmix64(0x5EED + patch id), generated on the device
    u = g.new_vector(0)
    g0, f0, u0 = g, f, u
    cells_global = [H.sizes(l)[1] * n ** a.dim for l in range(H.num_levels)]
    # this rank's sites in patches that copy through to the next level, summed over the levels (refined meshes only)
    copy_sites = 0
    if a.mesh:
        for l in range(H.num_levels - 1):
            t_l = H.tables(l)
            copy_sites += int(np.count_nonzero((t_l["orth_on_parent"] < 0) & (t_l["rank"] == rank))) * n ** a.dim
    # patches per rank and level; a gathered level that every rank holds and computes itself (TE_REPLICATE) is "replicated"
    placement = [("replicated" if H.replicated(l) else [int(c) for c in np.bincount(H.tables(l)["rank"], minlength=world)])
                 for l in range(H.num_levels)] if world > 1 else None
    red_dev = "cuda" if backend == "nccl" else "cpu"

    def barrier():
        g.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def measure(o, steps, warmup, g=None, f=None, u=None):
        """W untimed warm-up cycles with every kernel class timed (per-kernel table; names the dominant class), then
        exactly K cycles between two barriers with HIP events on the dominant class only; max over ranks
        (g, f, u: another solver and its vectors -- the problem-size block; default: the headline's)"""
        g, f, u = (g0 if g is None else g), (f0 if f is None else f), (u0 if u is None else u)
        g.profile(True)
        g.profile_select(None)
        # the very first step pays one-time costs (code load, lazy attributes): its events are dropped, so the table
        # covers `profiled_warm` cycles -- ONE variable for the reset rule and for every per-cycle figure derived from it
        first_profiled = 1 if warmup > 1 else 0
        profiled_warm = max(warmup - first_profiled, 0)
        for i in range(warmup):
            if i == first_profiled:
                g.profile_reset()
            g.cycle(o, f, u)
        barrier()
        rows_all = g.profile_rows()
        cand = {k: v for k, v in rows_all.items() if k in ALG_BYTES and ALG_BYTES[k] > 0}
        dom_name = max(cand.items(), key=lambda kv: kv[1]["ms"])[0] if cand else None
        if dist is not None and dom_name is not None:  # same class on every rank (rank 0 decides)
            names = [dom_name]
            dist.broadcast_object_list(names, src=0)
            dom_name = names[0]
        # timed region: HIP events around the dominant class only (an event pair per launch costs a few
        # microseconds of stream time; a V-cycle is a dozen launches)
        stride = 1
        if os.environ.get("TE_BENCH_NOPROFILE") is not None:  # tooling: wall time only
            g.profile(False)
        else:
            g.profile_select(dom_name)  # (--warmup 0: no candidate yet, every class is timed)
            # an event pair on a dispatch costs microseconds of stream time on this runtime (profiles/r06_event_cost.txt: 24-28 us of a
            # 210-us 4096^2 cycle with its six timed launches, 0-20 us of a 512^3 cycle with its one): inside the timed
            # region every FOURTH launch of the dominant class carries one (a stride without a common factor with the class's launches
            # per cycle, so that the sample walks over all its levels; TE_BENCH_EVENT_STRIDE=1: every launch, as before round 6)
            if dom_name is not None and steps >= 8 and profiled_warm > 0:
                per_cycle = max(1, round(rows_all[dom_name]["calls"] / profiled_warm)) if dom_name in rows_all else 1
                want = int(os.environ.get("TE_BENCH_EVENT_STRIDE", "4"))
                stride = next((c for c in (want, want + 1, want - 1, want + 3) if c >= 1 and math.gcd(c, per_cycle) == 1), 1)
            g.profile_stride(stride)
        g.profile_reset()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.cycle(o, f, u)
        barrier()
        dt = time.perf_counter() - t0
        rows = g.profile_rows()
        g.profile(False)
        g.profile_select(None)
        g.profile_stride(1)
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return {"dt": dt, "rows_all": rows_all, "rows": rows, "profiled_warm": profiled_warm, "event_stride": stride}

    def median_cycle_ms(o, steps, g=None, f=None, u=None):
        """median of per-cycle times: an event pair per cycle on the solver stream"""
        g, f, u = (g0 if g is None else g), (f0 if f is None else f), (u0 if u is None else u)
        ext = torch.cuda.ExternalStream(int(g.stream()))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()
        with torch.cuda.stream(ext):
            for e0, e1 in evs:
                e0.record()
                g.cycle(o, f, u)
                e1.record()
        barrier()
        return float(np.median([e0.elapsed_time(e1) for e0, e1 in evs]))

    def reduction_per_cycle(r):
        """sanity: the cycle must actually reduce the residual (guards against timing a no-op)"""
        g.residual(u, f, r)
        rn, fn = r.twoNormSqLocal(), f.twoNormSqLocal()
        if dist is not None:
            tt = torch.tensor([rn, fn], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt)
            rn, fn = tt.tolist()
        return float(np.sqrt(rn / fn))

    def checksum(v):
        """te_vec_checksum over all ranks: the sum modulo 2^64 of the bit patterns of v's values -- independent of the order of the
        patches and of how they are cut over ranks, so the lines of N = 1, 2, 4, 8 must print the SAME number for every vector
        whose computation is order-independent (a cycle's result is; a Krylov solve's is not: its dot products are summed rank
        by rank). The ranks' parts are added as two 32-bit halves in int64 (no overflow below 2^31 ranks)."""
        c = v.checksumLocal()
        if dist is not None:
            tt = torch.tensor([c & 0xFFFFFFFF, c >> 32], dtype=torch.int64, device=red_dev)
            dist.all_reduce(tt)
            lo, hi = (int(x) for x in tt.tolist())
            c = (lo + (hi << 32)) & 0xFFFFFFFFFFFFFFFF
        return f"{c:016x}"

    m = measure(opts, a.steps, a.warmup)
    dt, rows_all, rows = m["dt"], m["rows_all"], m["rows"]
    u_checksum = checksum(u)  # (every cycle of the timed region starts from u = 0 on the same f: this is the result of ONE cycle)
    ms_per_step = dt / a.steps * 1e3
    value = cells_global[0] / (dt / a.steps)
    # median of per-cycle times: a second, untimed-by-the-contract pass with an event pair per cycle on
    # the solver stream (`value` above stays the contract's K steps between two synchronisations)
    ms_median = median_cycle_ms(opts, a.steps)

    # N > 1: what makes a SCALE record diagnosable -- per rank and cycle, the time inside the exchanges as the solver stream
    # sees it (RCCL's kernels plus the wait for the peers), the pack / unpack launches, the kernels, and the host time to
    # enqueue one cycle; minimum and maximum over the ranks. And the rank count RCCL itself reports for the communicator.
    sharded = None
    if world > 1:
        tq = []
        for _ in range(10):
            g.sync()
            t0h = time.perf_counter()
            g.cycle(opts, f, u)
            tq.append(time.perf_counter() - t0h)
        g.sync()
        pw = max(1, m["profiled_warm"])
        per = lambda k: rows_all.get(k, {"ms": 0.0})["ms"] / pw * 1e3  # noqa: E731
        kern = sum(v["ms"] for k, v in rows_all.items() if k not in ("exchange", "pack")) / pw * 1e3
        mine = [per("exchange"), per("pack"), kern, float(np.median(tq)) * 1e6,
                rows_all.get("exchange", {"calls": 0})["calls"] / pw, sum(v["calls"] for v in rows_all.values()) / pw]
        lo = torch.tensor(mine, dtype=torch.float64, device=red_dev)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        names_ = ["exchange_us", "pack_us", "kernels_us", "host_enqueue_us", "exchanges", "launches_plus_exchanges"]
        rn, rr_ = g.comm_info()
        sharded = {"rccl_nranks": rn, "rccl_rank_of_rank0": rr_, "per_cycle_min_max_over_ranks": {k: [float(lo[i]), float(hi[i])] for i, k in enumerate(names_)},
                   "note": "from the warm-up cycles, every class timed (event pairs cost a few us per launch: the sum exceeds ms_per_step); "
                           "exchange_us includes waiting for the peers"}

    # measured ceiling taken in the same run: a bare 2-read + 1-write fp64 stream over the
    # finest-level vectors, one 16-B element per thread (te_vec_scale_then_add_scaled: y = a y + b x)
    r = g.new_vector(0)
    r.set(1.0)
    for _ in range(3):
        r.scaleThenAddScaled(0.5, 0.25, f)
    g.sync()
    t1 = time.perf_counter()
    for _ in range(20):
        r.scaleThenAddScaled(0.5, 0.25, f)
    g.sync()
    triad_gbs = 20 * 24.0 * (H.sizes(0)[0] * n ** a.dim) / (time.perf_counter() - t1) / 1e9
    reduction = reduction_per_cycle(r)

    def fused_hbm(rows_w, profiled_warm, local_sites, sec_per_step, copy_sites_=0):
        """whole cycle against the bytes its fused kernels must move (a roofline fraction): the kernels' own algorithmic bytes from
        the warm-up table, every class that has an ALG_BYTES entry; the others are listed, not silently counted as zero"""
        counted = sorted(k for k in rows_w if k in ALG_BYTES)
        uncounted = sorted(k for k in rows_w if k not in ALG_BYTES)
        fb = sum(ALG_BYTES[k] * rows_w[k]["cells"] for k in counted) / max(1, profiled_warm)
        fb += 14.0 * copy_sites_  # (refined meshes: the residual down and the correction up of a patch that copies through, at full size)
        return {"fused_alg_bytes_per_finest_site": fb / local_sites, "fused_achieved_GBs": fb / sec_per_step / 1e9,
                "fused_frac_of_peak": fb / sec_per_step / 1e9 / HBM_PEAK_GBS, "classes_counted": counted, "classes_without_bytes": uncounted}

    if rank == 0 and not rows:  # TE_BENCH_NOPROFILE=1 (tooling): wall time only
        print(json.dumps({"ms_per_step": ms_per_step, "value": value}), file=json_out, flush=True)
        out = None
    elif rank == 0:
        name, st = max(((k, v) for k, v in rows.items() if k in ALG_BYTES), key=lambda kv: kv[1]["ms"])
        roof = roofline_of(name, st, workload_key(a), world, copy_sites, a.steps / m["event_stride"])
        # (every event_stride-th launch of the class inside the timed region carried an event pair: `launches` of them)
        roof["event_stride"] = m["event_stride"]
        roof["measured_triad_GBs"] = triad_gbs
        roof["frac_of_measured_triad"] = roof["achieved"] / triad_gbs
        b_alg = vcycle_alg_bytes_per_finest_cell(cells_global)
        local_sites = H.sizes(0)[0] * n ** a.dim
        out = {
            "metric": metric_name(a),
            "value": value, "unit": "lattice-site updates/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # The result of one cycle of the timed region as te_vec_checksum over all ranks: a sharded cycle is bit-identical to
            # the single-rank cycle by construction, so the lines of N = 1, 2, 4, 8 on the same workload must carry the SAME value --
            # SURVEY 8(e)'s equality test, run by whoever runs the scaling bench
            "u_checksum_after_timed_region": u_checksum,
            "config": {"workload": f"apps/{a.dim}d/steady-equivalent: " + (f"{os.path.basename(a.mesh)} --divide {a.divide}" if a.mesh else f"{a.size}^{a.dim} uniform") + f", {cells_global[0] // n ** a.dim} "
                                   f"patches of {n}^{a.dim}, {H.num_levels} levels, V(1,1), smoother={a.smoother}, "
                                   "Dirichlet, f ~ U(-1,1) splitmix64(0x5EED + patch id)",
                       "workload_key": workload_key(a),
                       # what carried the exchanges of the timed region (N > 1)
                       "parallelism": f"patch-sharded x{world} (Morton ranges)", "exchange": exchange_backend,
                       "exchange_timeout_s": float(os.environ.get("TE_EXCHANGE_TIMEOUT", "300")) if world > 1 else None,
                       "levels": H.num_levels,
                       "patches_per_level": [c // n ** a.dim for c in cells_global],
                       # where every level lives (patches per rank): coarse levels gathered on rank 0 show up here
                       "placement_patches_per_rank": placement,
                       "agglomerate/max/replicate": list(H.placement()) if world > 1 else None,
                       "autotune": autotune, "sharded": sharded,
                       "smoother": a.smoother, "residual_reduction_per_cycle": reduction},
            "roofline": roof,
            # whole cycle: (i) against the bytes its fused kernels must move (a roofline fraction); (ii) SURVEY 8(d)'s
            # UNFUSED definition (113 B per finest site) divided by the fused cycle's time -- a work-equivalent rate that
            # can exceed the HBM peak because the fused cycle moves about a third of those bytes; not a roofline fraction
            "vcycle_hbm": {**fused_hbm(rows_all, m["profiled_warm"], local_sites, dt / a.steps, copy_sites),
                           "unfused_definition_bytes_per_finest_site": b_alg,
                           "unfused_definition_equivalent_GBs": b_alg * cells_global[0] / (dt / a.steps) / 1e9,
                           "unfused_definition_equivalent_over_peak": b_alg * cells_global[0] / (dt / a.steps) / 1e9 / (HBM_PEAK_GBS * world)},
            "ms_per_step_median": ms_median,
            # host wall time of the set-up calls on this rank (SURVEY (f)3: 'removes the host setup bottleneck')
            "setup_ms": setup_ms,
            # per-kernel table from the warm-up steps (every class timed there; the timed region times only `kernel`)
            "kernels_warmup": {k: {"calls": v["calls"], "ms": round(v["ms"], 4),
                                   "GBs": (ALG_BYTES.get(k, 0) * v["cells"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None}
                               for k, v in rows_all.items()},
        }
        rm = roofline_mfma_of(name, st, a.dim, n)
        if rm:
            out["roofline_mfma"] = rm
    else:
        out = None

    def provisional(next_block):
        """the line as it stands, before a block that may end the process: kept for main()'s handler (any exception from here on
        prints it with `error`), and under the supervisor (N > 1) handed over as well (a worker that dies without a word)"""
        HEADLINE["out"], HEADLINE["block"] = out, next_block
        if out is not None and os.environ.get("TE_BENCH_WORKER") is not None:
            print(PROVISIONAL + json.dumps({**out, "next_block": next_block}), file=json_out, flush=True)

    def optional_failed(name, e):
        """an optional block raised: its error under its own name. A library error (TeError) comes out of reductions -- the same on every
        rank, the run goes on; anything else at N > 1 may be this rank's alone, and the collectives of the blocks that follow would
        wait for it: the line is finished as it stands and the exception ends the run (main() prints the line with `error`)"""
        if out is not None:
            out.setdefault("secondary", {})[name] = {"error": f"{type(e).__name__}: {e}"}
        if world > 1 and not isinstance(e, capi.TeError):
            HEADLINE["out"], HEADLINE["block"] = out, f"secondary.{name}"
            raise e

    # from here on every block is optional: the line as it stands is handed to the supervisor (N > 1), and a block that fails
    # leaves an `error` entry under its own name instead of taking the headline with it
    provisional("the secondary blocks")


    # (f)1 beside the headline: the reference's own smoother (FFTBlockJacobiSmoother.h:55-58, the one cycle with per-V-cycle
    # parity against the reference) timed by the same driver run, a dozen cycles, both roofs
    if a.smoother == "rbgs" and not a.no_secondary and a.dim == 3 and n == 32 and os.environ.get("TE_BENCH_NOPROFILE") is None:
      HEADLINE["block"] = "secondary.reference_smoother"
      try:
        if os.environ.get("TE_BENCH_TEST_RAISE") == "reference_smoother":  # test hook: an exception that is not the library's
            raise ValueError("max() arg is an empty sequence")
        o2 = g.default_opts(smoother=capi.SMOOTH_PATCH_SOLVE)
        k2, w2 = 20, 5  # (an MFMA-bound kernel's clock settles over the first cycles: the quoted figure is a MEDIAN of 20 behind 5)
        m2 = measure(o2, k2, w2)
        med2 = median_cycle_ms(o2, k2)
        cs2 = checksum(u)
        red2 = reduction_per_cycle(r)
        if rank == 0 and m2["rows"]:
            name2, st2 = max(((k, v) for k, v in m2["rows"].items() if k in ALG_BYTES), key=lambda kv: kv[1]["ms"])
            ps_key = workload_key(argparse.Namespace(**{**vars(a), "smoother": "patch_solve"}))
            out.setdefault("secondary", {})["reference_smoother"] = {
                "what": "the same workload with the reference's block-Jacobi smoother (exact patch solves on the fp64 matrix cores), "
                        "default options (fuse = 3)",
                "steps": k2, "warmup": w2, "ms_per_step": m2["dt"] / k2 * 1e3, "ms_per_step_median": med2,
                "value": cells_global[0] / (m2["dt"] / k2),
                "unit": "lattice-site updates/s", "residual_reduction_per_cycle": red2, "u_checksum_after_timed_region": cs2,
                "roofline": roofline_of(name2, st2, ps_key, world), "roofline_mfma": roofline_mfma_of(name2, st2, a.dim, n),
                "kernels_warmup": {k: {"calls": v["calls"], "ms": round(v["ms"], 4)} for k, v in m2["rows_all"].items()}}
      except Exception as e:  # noqa: BLE001 -- an optional block never costs the headline its line
        optional_failed("reference_smoother", e)

    # (f)2 in the driver's run: time to solution of the call a user makes (apps/3d/steady.cpp:519-524): BiCGStab (BiCGStab.h:45-106,
    # tolerance 1e-12) preconditioned with one V-cycle, the drivers' trig problem on the same grid, both smoothers. A first solve
    # warms up (work vectors, code), the second is timed between two synchronisations, a third runs with every kernel class
    # timed and gives the algorithmic bytes per site and iteration.
    if not a.no_secondary and not a.mesh and os.environ.get("TE_BENCH_NOPROFILE") is None:
      HEADLINE["block"] = "secondary.solve"
      try:
        solve = {}
        sites_local = H.sizes(0)[0] * n ** a.dim
        bb, xx = g.new_vector(0), g.new_vector(0)
        g.init_problem(bb, None, problem=capi.PROBLEM_TRIG)
        for sname in ("rbgs", "patch_solve"):
            o3 = g.default_opts(smoother=smoothers[sname])
            xx.set(0.0)
            its, rr = g.bicgstab(xx, bb, o3)
            xx.set(0.0)
            barrier()
            t0s = time.perf_counter()
            its, rr = g.bicgstab(xx, bb, o3)
            barrier()
            dts = time.perf_counter() - t0s
            x_cs = checksum(xx)
            xx.set(0.0)
            g.profile(True)
            g.profile_select(None)
            g.profile_reset()
            g.bicgstab(xx, bb, o3)
            prow = g.profile_rows()
            g.profile(False)
            if dist is not None:
                tt = torch.tensor([dts], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dts = float(tt.item())
            cyc = {k for k in prow if k.startswith(("rbgs_", "stencil_rbgs", "stencil_slabs", "patch_solve", "restrict", "fcorr", "resid_restrict", "prolong", "patch_rhs", "dst_axis", "cf_ghost"))}
            krylov = sum(ALG_BYTES.get(k, 8.0) * v["cells"] for k, v in prow.items() if k not in cyc and k not in ("exchange", "pack")) / max(its, 1) / max(sites_local, 1)
            allb = sum(ALG_BYTES.get(k, 8.0) * v["cells"] for k, v in prow.items()) / max(its, 1) / max(sites_local, 1)
            solve[sname] = {"iterations": its, "ms": dts * 1e3, "rel_resid": rr, "ms_per_iteration": dts * 1e3 / max(its, 1),
                            # (equal across N only where the dot products are summed in the same order: informative, not an invariant)
                            "x_checksum": x_cs,
                            "alg_bytes_per_site_per_iteration": {"outside_the_two_cycles": krylov, "all_kernels": allb}}
        del bb, xx
        g.release_workspace()
        if rank == 0:
            out.setdefault("secondary", {})["solve"] = {"what": f"te_bicgstab + one V(1,1) cycle as preconditioner to 1e-12, trig problem ({'apps/3d/steady.cpp:253-265' if a.dim == 3 else 'apps/2d/steady.cpp:314-318'}) on the benchmarked grid, "
                                          "second of two solves, wall time between two synchronisations (max over ranks)", **solve}
      except Exception as e:  # noqa: BLE001
        optional_failed("solve", e)

    # N > 1: the OTHER transport. One more cycle on the headline's transport is the reference; then the direct-store transport
    # (te_gmg_use_push: hipIpc-mapped peer buffers, flags, bounded waits) is prepared, must reproduce that result BIT FOR BIT on
    # every rank -- after a cycle on other data, so that a stale ghost plane would show; three times -- with no check of its
    # protocol tripped (te_gmg_push_failed), and only then is timed like the headline: secondary.direct_store, a second figure,
    # never `value`. config.sharded.verified_bit_identical says whether the two transports agreed.
    if world > 1 and a.dim == 3 and not a.no_secondary and os.environ.get("TE_BENCH_PUSH", "1") != "0" and os.environ.get("TE_BENCH_NOPROFILE") is None:
        provisional(DIRECT_STORE_BLOCK)
        ds = {"what": "the same cycles with the face exchanges and the gathers of restricted blocks as direct stores into the peers' "
                      "hipIpc-mapped buffers (te_gmg_use_push) instead of " + exchange_backend}
        verified = None
        try:
            g.set_option("TE_PUSH_NONFATAL", "1")  # (a wait that gives up is reported below, not turned into exit 86)
            g.set_option("TE_PUSH_TIMEOUT", os.environ.get("TE_PUSH_TIMEOUT", "10"))
            g.cycle(opts, f, u)
            uref = g.new_vector(0)
            uref.copy(u)
            ref_cs = checksum(uref)
            g.use_push(True)  # (collective: succeeds or fails on all ranks together)
            f2, d = g.new_vector(0), g.new_vector(0)
            f2.copy(f)
            f2.scale(-0.625)
            bad = 0
            for _trial in range(3):
                g.cycle(opts, f2, u)
                g.cycle(opts, f, u)
                d.copy(u)
                d.addScaled(-1.0, uref)
                bad = max(bad, 9 if d.infNorm() != 0.0 else 0, g.push_failed())
            tt = torch.tensor([bad], dtype=torch.int64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            bad = int(tt.item())
            verified = (bad == 0)
            ds.update({"verified_bit_identical": verified, "u_checksum": checksum(u), "u_checksum_headline_transport": ref_cs,
                       "failure_code_max_over_ranks": bad,
                       "failure_codes": "0 none; 1 a wait gave up; 2 a peer's flag two exchanges ahead; 3 a peer behind when its buffer "
                                        "was overwritten; 4 epochs out of sequence; 9 result differs"})
            if verified:
                m3 = measure(opts, a.steps, a.warmup)
                ds.update({"steps": a.steps, "warmup": a.warmup, "ms_per_step": m3["dt"] / a.steps * 1e3, "ms_per_step_median": median_cycle_ms(opts, a.steps),
                           "value": cells_global[0] / (m3["dt"] / a.steps), "unit": "lattice-site updates/s",
                           "u_checksum_after_timed_region": checksum(u), "push_failed_after_timed_region": g.push_failed()})
            g.use_push(False)
            del f2, d, uref
        except capi.TeError as e:  # (set-up failures come out of reductions: the same on every rank)
            ds["error"] = str(e)
            try:
                g.use_push(False)  # (collective, like the failure that brought every rank here)
            except capi.TeError:
                pass
        except Exception as e:  # noqa: BLE001 -- (this rank's alone, possibly: the line as it stands, then the end of the run)
            optional_failed("direct_store", e)
        finally:
            try:
                g.set_option("TE_PUSH_NONFATAL", None)  # (the headline solver goes back to a watchdog that ends the process on a failed exchange)
            except Exception:  # noqa: BLE001
                pass
        if out is not None:
            out.setdefault("secondary", {})["direct_store"] = ds
            if out["config"]["sharded"] is not None:
                out["config"]["sharded"]["verified_bit_identical"] = {
                    "headline_transport": exchange_backend, "other_transport": "direct stores (te_gmg_use_push)", "identical": verified,
                    "note": "one cycle on the same right-hand side through each transport, compared bit for bit on every rank (maximum over the "
                            "ranks of the differences' infinity norm == 0), three times, each after a cycle on other data; null: the other "
                            "transport could not be set up here (secondary.direct_store.error)"}

    # the reference's "GMG Setup" timer (apps/3d/steady.cpp:480-484) is the cost of building a cycle in a process that already runs:
    # a SECOND solver on the same hierarchy, created and destroyed here (rank-local; no collective inside te_gmg_create)
    if out is not None and world == 1 and not a.no_secondary and os.environ.get("TE_BENCH_NOPROFILE") is None:
        HEADLINE["block"] = "setup_ms.te_gmg_create_second"
        try:
            t0c = time.perf_counter()
            gs = capi.GMG(H, device=local_rank)
            t1c = time.perf_counter()
            out["setup_ms"]["te_gmg_create_second"] = (t1c - t0c) * 1e3
            out["setup_ms"]["te_gmg_create_second_parts"] = gs.setup_ms()
            del gs
        except Exception as e:  # noqa: BLE001
            out["setup_ms"]["te_gmg_create_second"] = {"error": f"{type(e).__name__}: {e}"}

    # N = 1: a problem-size axis (apps/3d/steady.cpp:95 --divide, OctTree.h:119-179): the same cycle at 1024^3 -- 32 768 patches,
    # 8 GiB per vector, the shape at which eight GPUs would hold 512^3 each (`--size 1024 --gpus 8` is that weak-scaling twin)
    if world == 1 and a.dim == 3 and a.size == 512 and not a.mesh and n == 32 and not a.no_secondary and os.environ.get("TE_BENCH_NOPROFILE") is None \
            and os.environ.get("TE_BENCH_NO_1024") is None:
        HEADLINE["block"] = "secondary.size_1024"
        try:
            del r
            mesh2 = capi.Mesh.uniform(3, 5)
            H2 = capi.Hierarchy(mesh2, n)
            g2 = capi.GMG(H2, device=local_rank)
            fb, ub = g2.new_vector(0), g2.new_vector(0)
            g2.init_problem(fb, None, problem=capi.PROBLEM_RANDOM)
            o4 = g2.default_opts(smoother=smoothers[a.smoother])
            k4, w4 = 10, 3
            m4 = measure(o4, k4, w4, g2, fb, ub)
            med4 = median_cycle_ms(o4, k4, g2, fb, ub)
            cs4 = f"{ub.checksumLocal():016x}"
            rb = g2.new_vector(0)
            g2.residual(ub, fb, rb)
            red4 = float(np.sqrt(rb.twoNormSqLocal() / fb.twoNormSqLocal()))
            sites4 = H2.sizes(0)[1] * n ** 3
            name4, st4 = max(((k, v) for k, v in m4["rows"].items() if k in ALG_BYTES), key=lambda kv: kv[1]["ms"])
            a4 = argparse.Namespace(**{**vars(a), "size": 1024})
            out["secondary"] = out.get("secondary") or {}
            out["secondary"]["size_1024"] = {
                "what": "the headline's cycle on 1024^3 (32 768 patches of 32^3, 6 levels) on this one GPU: 8 GiB per vector; "
                        "`bench.py --size 1024 --gpus 8` is the weak-scaling twin of the headline (512^3 per GPU)",
                "steps": k4, "warmup": w4, "ms_per_step": m4["dt"] / k4 * 1e3, "ms_per_step_median": med4, "value": sites4 / (m4["dt"] / k4),
                "unit": "lattice-site updates/s", "residual_reduction_per_cycle": red4, "u_checksum_after_timed_region": cs4,
                "patches_per_level": [H2.sizes(l)[1] for l in range(H2.num_levels)],
                "roofline": roofline_of(name4, st4, workload_key(a4), world),
                "vcycle_hbm": fused_hbm(m4["rows_all"], m4["profiled_warm"], sites4, m4["dt"] / k4),
                "kernels_warmup": {k: {"calls": v["calls"], "ms": round(v["ms"], 4)} for k, v in m4["rows_all"].items()}}
            del rb, fb, ub, g2, H2
        except Exception as e:  # noqa: BLE001 -- an optional block never costs the headline its line
            out.setdefault("secondary", {})["size_1024"] = {"error": f"{type(e).__name__}: {e}"}

    if out is not None:
        if world == 1 and not a.no_cpu_baseline:
            HEADLINE["block"] = "cpu_baseline"
            try:
                from oracle import build as obuild
                obuild.build_oracle()
                out["cpu_baseline"] = cpu_baseline(a.cpu_size or a.size, a.dim, n, a.smoother)
            except Exception as e:  # noqa: BLE001 -- the reported baseline never costs the headline its line
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        HEADLINE["out"] = None  # (printed here)
        print(json.dumps(out), file=json_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
