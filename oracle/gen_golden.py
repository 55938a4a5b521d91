#!/usr/bin/env python3
"""TEST INFRASTRUCTURE. Generates tests/golden/ref_*.npz from the reference's own compiled code
(oracle/_ref/libte_ref.so, built by oracle/Makefile.ref from /root/reference sources).

Each fixture holds the flat level tables, seeded inputs and the reference's outputs for
  a3  StarPatchOp::applyWithInterface      a4  StarPatchOp::apply
  a5  StarPatchOp::addInterfaceToRHS       a6  TriLinInterp / BilinearInterpolator::interpolate
  a10 Vector<D> BLAS-1 virtuals            caller: BiCGStab<D>::solve (unpreconditioned)
plus Tree<D> node tables after refineLeaves. Run in the build container only (the GPU box has no
reference); the .npz files are committed, this script is how they were made:

    make -C oracle -f Makefile.ref
    LD_LIBRARY_PATH=/usr/lib/x86_64-linux-gnu:/opt/conda/lib python oracle/gen_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from oracle import refslice  # noqa: E402
from pressurepoissonsolver_amd import capi  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def level_case(mesh_file, dim, n, neumann, seed):
    REF = refslice.lib()
    p = refslice.p
    m = capi.Mesh.read(os.path.join(GOLDEN, mesh_file), dim)
    H = capi.Hierarchy(m, n, neumann=neumann)
    t = H.tables(0)
    L = orc.Level.from_tables(t, dim, n, neumann)
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1, 1, L.size)
    f = rng.uniform(-1, 1, L.size)
    nif = REF.ref_num_ifaces(C.byref(L.c))
    gamma_in = rng.uniform(-1, 1, nif * L.nf)
    out = dict(dim=dim, n=n, neumann=int(neumann), mesh=mesh_file, u=u, f=f, gamma_in=gamma_in, num_ifaces=nif)
    for k, v in L.a.items():
        out["t_" + k] = v
    ii = np.zeros((L.P, 2 * dim), np.int32)
    REF.ref_iface_index(C.byref(L.c), p(ii))
    out["iface_index"] = ii
    out["gamma"] = refslice.interp(L, u)
    out["apply_with_gamma"] = refslice.apply_with_gamma(L, u, gamma_in)
    if dim == 3:
        au7 = np.zeros(L.size)
        REF.ref_apply_with_gamma_7pt(C.byref(L.c), p(u), p(gamma_in), p(au7))
        assert np.array_equal(out["apply_with_gamma"], au7), "reference StarPatchOp<3> and SevenPtPatchOperator disagree"
    # A u with gamma built from u itself == SchurHelper::apply on one rank
    out["apply"] = refslice.apply_with_gamma(L, u, out["gamma"])
    out["patch_apply"] = refslice.patch_apply(L, u)
    out["add_iface_rhs"] = refslice.add_iface_rhs(L, gamma_in, f)
    if not neumann and "multi_refine_8" not in mesh_file:  # (the unpreconditioned solve does not converge in 1000 iterations on the 9-level trees)
        x, its = refslice.bicgstab(L, f)
        out["bicg_its"] = its
        out["bicg_x"] = x
    return out


def bcgs_case(mesh_file, dim, n, neumann, seed, tol=1e-12, max_it=1000):
    """one block-Jacobi sweep with the reference's Krylov patch solver (refslice.smooth_bcgs: BiCGStab<D>::solve per patch on
    StarPatchOp<D>::apply, the calls of PatchSolvers/BiCGStabSolver.h:114-132) from a seeded iterate"""
    m = capi.Mesh.read(os.path.join(GOLDEN, mesh_file), dim)
    H = capi.Hierarchy(m, n, neumann=neumann)
    L = orc.Level.from_tables(H.tables(0), dim, n, neumann)
    rng = np.random.default_rng(seed)
    u, f = rng.uniform(-1, 1, L.size), rng.uniform(-1, 1, L.size)
    out = dict(dim=dim, n=n, neumann=int(neumann), mesh=mesh_file, u=u, f=f, tol=tol, max_it=max_it)
    for k, v in L.a.items():
        out["t_" + k] = v
    out["u_out"], out["its"] = refslice.smooth_bcgs(L, f, u, tol, max_it)
    return out


def vecop_case(seed):
    REF = refslice.lib()
    p = refslice.p
    n, P = 4, 3
    rng = np.random.default_rng(seed)
    v0, a, b = (rng.uniform(-1, 1, P * n ** 3) for _ in range(3))
    alpha, beta, gamma = 0.75, -1.25, 2.5
    out = dict(n=n, P=P, v0=v0, a=a, b=b, alpha=alpha, beta=beta, gamma=gamma)
    for op in range(10):
        v = v0.copy()
        REF.ref_vecop(op, n, P, p(v), p(a), p(b), alpha, beta, gamma)
        out[f"op{op}"] = v
    return out


def main():
    cases = [("2uni.bin", 3, 4, False), ("2uni.bin", 3, 8, False), ("2refine.bin", 3, 4, False),
             ("2refine.bin", 3, 8, False), ("2refine.bin", 3, 4, True), ("1uni.bin", 3, 8, True),
             ("2d2uni.bin", 2, 8, False), ("2d2ref.bin", 2, 8, False), ("2d2ref.bin", 2, 4, True)]
    # round 3: deep adaptively refined trees (apps/3d/meshes/multi_refine{,_8}.bin, apps/2d/meshes/multi_refine_8.bin)
    cases += [("multi_refine.bin", 3, 4, False), ("multi_refine.bin", 3, 4, True), ("multi_refine_8.bin", 3, 4, False),
              ("2d_multi_refine_8.bin", 2, 4, False), ("2d_multi_refine_8.bin", 2, 8, True)]
    only = sys.argv[1] if len(sys.argv) > 1 else None  # e.g. "multi_refine": (re)generate the fixtures of matching meshes only
    for i, (mf, dim, n, neu) in enumerate(cases):
        if only and only not in mf:
            continue
        d = level_case(mf, dim, n, neu, 1000 + i)
        name = f"ref_{mf.split('.')[0]}_n{n}{'_neumann' if neu else ''}.npz"
        np.savez_compressed(os.path.join(GOLDEN, name), **d)
        print(name, "P", len(d["t_id"]), "ifaces", d["num_ifaces"], "bicg its", d.get("bicg_its"))
    if not only or only == "bcgs":
        for j, (mf, dim, n, neu) in enumerate([("2d2uni.bin", 2, 8, False), ("2d2ref.bin", 2, 8, False), ("2d2ref.bin", 2, 4, True),
                                               ("2refine.bin", 3, 4, False)]):
            d = bcgs_case(mf, dim, n, neu, 2000 + j)
            name = f"bcgs_ref_{mf.split('.')[0]}_n{n}{'_neumann' if neu else ''}.npz"
            np.savez_compressed(os.path.join(GOLDEN, name), **d)
            print(name, "P", len(d["t_id"]), "its", d["its"].min(), "..", d["its"].max())
    if not only:
        np.savez_compressed(os.path.join(GOLDEN, "ref_vecops.npz"), **vecop_case(7))
    for mf, dim, div in [("2uni.bin", 3, 1), ("2refine.bin", 3, 1), ("2refine.bin", 3, 2), ("2d2ref.bin", 2, 2),
                         ("multi_refine.bin", 3, 0), ("multi_refine.bin", 3, 1), ("2d_multi_refine_8.bin", 2, 1)]:
        if only and only not in mf:
            continue
        d = refslice.tree_nodes(os.path.join(GOLDEN, mf), dim, div)
        d.update(mesh=mf, dim=dim, divides=div)
        np.savez_compressed(os.path.join(GOLDEN, f"ref_tree_{mf.split('.')[0]}_div{div}.npz"), **d)
        print("tree", mf, div, len(d["ilp"]), "nodes")


if __name__ == "__main__":
    main()
