"""TEST INFRASTRUCTURE. ctypes binding of oracle/_ref/libte_ref.so: the reference's own code for
the PETSc-free slice of the hot path (see oracle/ref_driver.cpp, oracle/Makefile.ref). Present
only where /root/reference was available at build time; `available()` says so."""
import ctypes as C
import os

import numpy as np

from oracle import oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libte_ref.so")
V = C.c_void_p
_lib = None


def available():
    if not os.path.exists(LIB_PATH):
        return False
    try:
        lib()
        return True
    except OSError:
        return False


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        PL = C.POINTER(orc.OrcLevel)  # identical layout to ref_level
        for name, res, args in [
                ("ref_num_ifaces", C.c_int, [PL]), ("ref_iface_index", None, [PL, V]),
                ("ref_interp", None, [PL, V, V]), ("ref_apply_with_gamma", None, [PL, V, V, V]),
                ("ref_apply_with_gamma_7pt", None, [PL, V, V, V]), ("ref_patch_apply", None, [PL, V, V]),
                ("ref_add_iface_rhs", None, [PL, V, V]), ("ref_bicgstab", C.c_int, [PL, V, V, C.c_int, C.c_double]),
                ("ref_smooth_bcgs", None, [PL, V, V, C.c_double, C.c_int, V]),
                ("ref_vecop", None, [C.c_int, C.c_int, C.c_int, V, V, V, C.c_double, C.c_double, C.c_double]),
                ("ref_tree_nodes", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, V, V, V, V, V, V]),
                ("ref_init", C.c_int, [])]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.ref_init()
        _lib = L
    return _lib


def p(a):
    return None if a is None else a.ctypes.data_as(V)


def _v(a):
    return np.ascontiguousarray(a, np.float64)


def interp(L, u):
    nif = lib().ref_num_ifaces(C.byref(L.c))
    g = np.zeros(nif * L.nf)
    lib().ref_interp(C.byref(L.c), p(_v(u)), p(g))
    return g


def apply_with_gamma(L, u, gamma):
    f = np.zeros(L.size)
    lib().ref_apply_with_gamma(C.byref(L.c), p(_v(u)), p(_v(gamma)), p(f))
    return f


def apply(L, u):
    return apply_with_gamma(L, u, interp(L, u))


def patch_apply(L, u):
    f = np.zeros(L.size)
    lib().ref_patch_apply(C.byref(L.c), p(_v(u)), p(f))
    return f


def add_iface_rhs(L, gamma, f):
    f = _v(f).copy()
    lib().ref_add_iface_rhs(C.byref(L.c), p(_v(gamma)), p(f))
    return f


def bicgstab(L, b, max_it=1000, tol=1e-12):
    x = np.zeros(L.size)
    its = lib().ref_bicgstab(C.byref(L.c), p(_v(b)), p(x), max_it, tol)
    return x, its


def smooth_bcgs(L, f, u, tol=1e-12, max_it=1000):
    """one sweep of the reference's own BiCGStab<D>::solve per patch on StarPatchOp<D>::apply (the body of
    PatchSolvers/BiCGStabSolver.h:114-132) -> (u, iterations per patch)"""
    u = _v(u).copy()
    its = np.zeros(max(L.c.P, 1), np.int32)
    lib().ref_smooth_bcgs(C.byref(L.c), p(_v(f)), p(u), tol, max_it, p(its))
    return u, its[:L.c.P]


def tree_nodes(path, dim, divides):
    mx = 200000
    ilp = np.zeros((mx, 3), np.int32)
    le, st = np.zeros((mx, dim)), np.zeros((mx, dim))
    nb, ch = np.zeros((mx, 2 * dim), np.int32), np.zeros((mx, 1 << dim), np.int32)
    nl = np.zeros(1, np.int32)
    cnt = lib().ref_tree_nodes(os.fsencode(path), dim, divides, mx, p(ilp), p(le), p(st), p(nb), p(ch), p(nl))
    if cnt <= 0:
        raise RuntimeError("ref_tree_nodes failed")
    return dict(num_levels=int(nl[0]), ilp=ilp[:cnt], lengths=le[:cnt], starts=st[:cnt], nbr=nb[:cnt], child=ch[:cnt])
