"""TEST INFRASTRUCTURE — writes tests/golden/c3_solve_<smoother>.npz: the CPU oracle's BiCGStab + GMG solve of the
headline configuration (C3: 512^3 uniform, 4096 patches of 32^3, trig problem, tolerance 1e-12), the call of
apps/3d/steady.cpp:519-524 over BiCGStab.h:45-106, run ONCE here (build container, minutes of host time per
smoother) so that the GPU tests can compare at full size without repeating it:

    its, rr                 iteration count and final relative residual
    x_norm2, err_rel        ||x||_2 and ||x - exact||_2 / ||exact||_2 (exact = the analytic solution at cell centres)
    patch_id, cell, value   4096 sampled entries of x, keyed by tree node id of the patch and cell index inside it
                            (this build's patch order is its own Morton order: a consumer looks patches up by id)
    checksum                wrap-around uint64 sum of the bit patterns of x (order-independent)

Smoothers: 0 = the reference's block-Jacobi patch solve (FFTBlockJacobiSmoother.h:55-58), 2 = patch-local red-black
Gauss-Seidel (the headline smoother; the builder's restatement). Usage: python oracle/gen_c3_solve.py [div] [n] [mesh] [tag]
(defaults 4, 32 = C3; `3 32` writes the C2 twin used to cross-check the fixture format against a live run; round 6:
`2 32 2refine.bin c4` = C4 at the parity tests' size (960 patches) and `4 16 uniform d16` = 256^3 in 16^3 patches, whose live
oracle solves were 45 s of every GPU test run -- for these two the oracle's level tables come from oracle/levels_bfs.py, not from
the product's hierarchy)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from pressurepoissonsolver_amd import capi, problems  # noqa: E402  (host mesh tables + the numpy twin of Init.cpp only)

SAMPLES = 4096


def sample_index(P, nc, seed=0xC3):
    rng = np.random.default_rng(seed)
    return rng.integers(0, P, SAMPLES), rng.integers(0, nc, SAMPLES)


def main():
    div = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    mesh = sys.argv[3] if len(sys.argv) > 3 else "uniform"
    m = capi.Mesh.unit_root(3) if mesh == "uniform" else capi.Mesh.read(os.path.join(ROOT, "tests", "golden", mesh), 3)
    for _ in range(div):
        m.refine_leaves()
    H = capi.Hierarchy(m, n)
    if len(sys.argv) > 4:  # (the fixtures of round 6: level tables independent of the product's)
        from oracle import levels_bfs
        nodes = m.nodes()
        tabs = levels_bfs.tables_in_order(levels_bfs.extract_levels(nodes, 3), nodes, 3, [H.tables(l)["id"] for l in range(H.num_levels)])
        levels = [orc.Level.from_tables(t, 3, n, False) for t in tabs]
    else:
        levels = orc.levels_from_hierarchy(H)
    t0 = H.tables(0)
    ids = np.asarray(t0["id"])
    orc.set_threads(os.cpu_count() or 1)
    b, exact = problems.init_dirichlet(t0, n)
    nc = n ** 3
    pp, cc = sample_index(levels[0].P, nc)
    tag = sys.argv[4] if len(sys.argv) > 4 else {4: "c3", 3: "c2"}.get(div, f"div{div}") + ("" if n == 32 else f"_n{n}")
    for sm, name in ((0, "patch_solve"), (2, "rbgs")):
        t = time.time()
        x, its, rr = orc.bicgstab(levels, orc.cycle_opts(smoother=sm), b)
        out = os.path.join(ROOT, "tests", "golden", f"{tag}_solve_{name}.npz")
        np.savez(out, div=div, n=n, smoother=sm, its=its, rr=rr, x_norm2=np.linalg.norm(x),
                 err_rel=np.linalg.norm(x - exact) / np.linalg.norm(exact), patch_id=ids[pp].astype(np.int64),
                 cell=cc.astype(np.int64), value=x.reshape(-1, nc)[pp, cc],
                 checksum=np.add.reduce(x.view(np.uint64), dtype=np.uint64))
        print(f"{out}: its={its} rr={rr:.3e} ({time.time() - t:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
