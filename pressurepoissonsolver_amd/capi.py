"""ctypes binding of include/te_hip.h (libte_hip.so) — plumbing only.

The shared library is the product; this module adds nothing numerical. It fails loudly when the
library is missing (`TeLibraryMissing`): there is no Python / torch / CPU fallback for any
operation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TE_HIP_LIB_PATH: tooling -- another build of the library, build.build_variant, for same-box A/B runs)
LIB_PATH = os.environ.get("TE_HIP_LIB_PATH") or os.path.join(_HERE, "libte_hip.so")


class TeError(RuntimeError):
    """Non-zero status from the C ABI (the reference's convention is `throw 3;`)."""

    def __init__(self, code, msg):
        super().__init__(f"te_hip error {code}: {msg}")
        self.code = code


class TeLibraryMissing(ImportError):
    pass


TE_OK, TE_EINVAL, TE_EHIP, TE_ESTATE, TE_EIO, TE_EUNSUPPORTED, TE_ENOMEM = 0, -1, -2, -3, -4, -5, -6
SMOOTH_PATCH_SOLVE, SMOOTH_JACOBI, SMOOTH_RBGS, SMOOTH_PATCH_BCGS = 0, 1, 2, 3
PROBLEM_TRIG, PROBLEM_GAUSS, PROBLEM_RANDOM = 0, 1, 2


class CycleOpts(C.Structure):
    """GMG/CycleOpts.h:51-79 + the smoother selection of this build."""

    _fields_ = [("pre_sweeps", C.c_int32), ("post_sweeps", C.c_int32), ("coarse_sweeps", C.c_int32),
                ("mid_sweeps", C.c_int32), ("cycle_type", C.c_int32), ("smoother", C.c_int32),
                ("omega", C.c_double), ("exact_coarse", C.c_int32), ("fuse", C.c_int32)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                          C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                          C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p)

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)

# every symbol include/te_hip.h declares: (restype, argtypes)
_P = C.c_void_p
_D = C.c_double
_I = C.c_int
_PI32 = C.POINTER(C.c_int32)
_PD = C.POINTER(C.c_double)
SYMBOLS = {
    "te_last_error": (C.c_char_p, []),
    "te_version": (C.c_char_p, []),
    "te_mesh_read": (_I, [C.c_char_p, _I, C.POINTER(_P)]),
    "te_mesh_unit_root": (_I, [_I, C.POINTER(_P)]),
    "te_mesh_refine_leaves": (_I, [_P]),
    "te_mesh_num_nodes": (_I, [_P]),
    "te_mesh_num_levels": (_I, [_P]),
    "te_mesh_dim": (_I, [_P]),
    "te_mesh_get_nodes": (_I, [_P, _P, _P, _P, _P, _P]),
    "te_mesh_destroy": (None, [_P]),
    "te_hier_build": (_I, [_P, _I, _I, _I, _D, _I, _I, C.POINTER(_P)]),
    "te_hier_build_placed": (_I, [_P, _I, _I, _I, _D, _I, _I, _D, _I, _I, C.POINTER(_P)]),
    "te_hier_placement": (_I, [_P, _PD, C.POINTER(_I), C.POINTER(_I)]),
    "te_hier_num_levels": (_I, [_P]),
    "te_hier_dim": (_I, [_P]),
    "te_hier_n": (_I, [_P]),
    "te_hier_level_sizes": (_I, [_P, _I, C.POINTER(_I), C.POINTER(_I)]),
    "te_hier_level_replicated": (_I, [_P, _I]),
    "te_hier_level_tables": (_I, [_P, _I] + [_P] * 10),
    "te_hier_level_l2g": (_I, [_P, _I, _P]),
    "te_hier_destroy": (None, [_P]),
    "te_cycle_opts_default": (None, [C.POINTER(CycleOpts)]),
    "te_gmg_create": (_I, [_P, _I, C.POINTER(_P)]),
    "te_gmg_destroy": (None, [_P]),
    "te_gmg_num_levels": (_I, [_P]),
    "te_gmg_sync": (_I, [_P]),
    "te_gmg_stream": (_P, [_P]),
    "te_vec_create": (_I, [_P, _I, C.POINTER(_P)]),
    "te_vec_destroy": (None, [_P]),
    "te_vec_size": (C.c_size_t, [_P]),
    "te_vec_upload": (_I, [_P, _P]),
    "te_vec_download": (_I, [_P, _P]),
    "te_vec_upload_patches": (_I, [_P, _I, _I, _P]),
    "te_vec_download_patches": (_I, [_P, _I, _I, _P]),
    "te_vec_device_ptr": (_P, [_P]),
    "te_vec_set": (_I, [_P, _D]),
    "te_vec_scale": (_I, [_P, _D]),
    "te_vec_shift": (_I, [_P, _D]),
    "te_vec_copy": (_I, [_P, _P]),
    "te_vec_add": (_I, [_P, _P]),
    "te_vec_add_scaled": (_I, [_P, _D, _P]),
    "te_vec_add_scaled2": (_I, [_P, _D, _P, _D, _P]),
    "te_vec_scale_then_add": (_I, [_P, _D, _P]),
    "te_vec_scale_then_add_scaled": (_I, [_P, _D, _D, _P]),
    "te_vec_scale_then_add_scaled2": (_I, [_P, _D, _D, _P, _D, _P]),
    "te_vec_two_norm_sq": (_I, [_P, _PD]),
    "te_vec_inf_norm": (_I, [_P, _PD]),
    "te_vec_dot": (_I, [_P, _P, _PD]),
    "te_vec_checksum": (_I, [_P, C.POINTER(C.c_uint64)]),
    "te_apply": (_I, [_P, _I, _P, _P]),
    "te_patch_apply": (_I, [_P, _I, _P, _P]),
    "te_residual": (_I, [_P, _I, _P, _P, _P]),
    "te_residual_norm_sq": (_I, [_P, _I, _P, _P, _P, _PD]),
    "te_smooth": (_I, [_P, _I, _P, _P, _I, _D, _I]),
    "te_restrict": (_I, [_P, _I, _P, _P]),
    "te_prolong_add": (_I, [_P, _I, _P, _P]),
    "te_vcycle": (_I, [_P, C.POINTER(CycleOpts), _P, _P]),
    "te_bicgstab": (_I, [_P, C.POINTER(CycleOpts), _P, _P, _I, _D, C.POINTER(_I), _PD]),
    "te_gmg_release_workspace": (_I, [_P]),
    "te_gmg_set_option": (_I, [_P, C.c_char_p, C.c_char_p]),
    "te_gmg_set_exchange": (_I, [_P, EXCHANGE_FN, _P]),
    "te_rccl_unique_id": (_I, [C.c_char_p, C.c_char_p]),
    "te_gmg_use_rccl": (_I, [_P, C.c_char_p, C.c_char_p, _I, _I]),
    "te_gmg_set_allreduce": (_I, [_P, ALLREDUCE_FN, _P]),
    "te_gmg_verify_schedule": (_I, [_P, C.POINTER(CycleOpts)]),
    "te_gmg_autotune": (_I, [_P, C.POINTER(CycleOpts), _I, _PD, C.c_char_p, _I]),
    "te_gmg_comm_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "te_gmg_use_push": (_I, [_P, _I]),
    "te_gmg_push_failed": (_I, [_P]),
    "te_gmg_set_patch_bcgs": (_I, [_P, _D, _I]),
    "te_gmg_patch_bcgs_iterations": (_I, [_P, _I, _P]),
    "te_gmg_exchange_selftest": (_I, [_P, _I]),
    "te_gmg_watchdog_selftest": (_I, [_P, _D]),
    "te_gmg_setup_ms": (_I, [_P, _PD, _I]),
    "te_gmg_profile": (_I, [_P, _I]),
    "te_gmg_profile_rows": (_I, [_P, _I, _P, _P, _P, _P]),
    "te_gmg_profile_reset": (_I, [_P]),
    "te_gmg_profile_select": (_I, [_P, C.c_char_p]),
    "te_gmg_profile_stride": (_I, [_P, _I]),
    "te_init_problem": (_I, [_P, _I, _I, _I, _P, _P]),
    "te_integrate": (_I, [_P, _I, _P, _P]),
    "te_volume": (_I, [_P, _I, _P]),
}

_lib = None


def lib():
    """Load libte_hip.so (once). Raises TeLibraryMissing if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TeLibraryMissing(
                f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no fallback implementation.")
        # One HIP runtime per process: PyTorch's ROCm wheel bundles its own libamdhip64.so.7 (same soname as
        # /opt/rocm's). If torch is loaded first the dynamic loader hands that copy to libte_hip.so as well;
        # the other order leaves two runtimes that cannot see each other's allocations (torch.distributed /
        # RCCL would then reject our device pointers). A pure C++ host simply gets /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != TE_OK:
        raise TeError(rc, lib().te_last_error().decode())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Mesh:
    """Tree<D> (src/Thunderegg/OctTree.h:34)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def read(cls, path, dim=3):
        h = C.c_void_p()
        check(lib().te_mesh_read(os.fsencode(path), dim, C.byref(h)))
        return cls(h)

    @classmethod
    def unit_root(cls, dim=3):
        h = C.c_void_p()
        check(lib().te_mesh_unit_root(dim, C.byref(h)))
        return cls(h)

    @classmethod
    def uniform(cls, dim, divides):
        """1-node tree refined `divides` times: 2^(dim*divides) leaves."""
        m = cls.unit_root(dim)
        for _ in range(divides):
            m.refine_leaves()
        return m

    def refine_leaves(self):
        check(lib().te_mesh_refine_leaves(self.h))

    @property
    def dim(self):
        return lib().te_mesh_dim(self.h)

    @property
    def num_nodes(self):
        return lib().te_mesh_num_nodes(self.h)

    @property
    def num_levels(self):
        return lib().te_mesh_num_levels(self.h)

    def nodes(self):
        n, d = self.num_nodes, self.dim
        out = dict(ilp=np.zeros((n, 3), np.int32), lengths=np.zeros((n, d)), starts=np.zeros((n, d)),
                   nbr=np.zeros((n, 2 * d), np.int32), child=np.zeros((n, 1 << d), np.int32))
        check(lib().te_mesh_get_nodes(self.h, _ptr(out["ilp"]), _ptr(out["lengths"]), _ptr(out["starts"]),
                                      _ptr(out["nbr"]), _ptr(out["child"])))
        return out

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.te_mesh_destroy(self.h)
            self.h = None


class Hierarchy:
    """DomainGenerator + the CycleFactory level loop (ThundereggDomGen.h, CycleFactory3d.cpp:98-127)."""

    def __init__(self, mesh, n, neumann=False, max_levels=0, patches_per_proc=0.0, rank=0, nranks=1, placement=None):
        """placement = (agglomerate, agglomerate_max, replicate) spelled out (a negative entry = the default), or None: the
        environment's TE_AGGLOMERATE / TE_AGGLOMERATE_MAX / TE_REPLICATE, read once by te_hier_build"""
        self.h = C.c_void_p()
        if placement is None:
            check(lib().te_hier_build(mesh.h, n, int(neumann), max_levels, float(patches_per_proc), rank, nranks,
                                      C.byref(self.h)))
        else:
            agg, cap, rep = placement
            check(lib().te_hier_build_placed(mesh.h, n, int(neumann), max_levels, float(patches_per_proc), rank, nranks,
                                             float(agg), int(cap), int(rep), C.byref(self.h)))
        self.n = n
        self.neumann = bool(neumann)
        self.dim = lib().te_hier_dim(self.h)
        self.num_levels = lib().te_hier_num_levels(self.h)
        self.rank, self.nranks = rank, nranks

    def sizes(self, level):
        a, b = C.c_int(), C.c_int()
        check(lib().te_hier_level_sizes(self.h, level, C.byref(a), C.byref(b)))
        return a.value, b.value

    def placement(self):
        """(agglomerate, agglomerate_max, replicate) this hierarchy was built with"""
        a, b, c = C.c_double(), C.c_int(), C.c_int()
        check(lib().te_hier_placement(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def replicated(self, level):
        """the level lives on every rank (a gathered coarse level, TE_REPLICATE): each rank holds and computes all of it"""
        r = lib().te_hier_level_replicated(self.h, level)
        if r < 0:
            check(r)
        return bool(r)

    def tables(self, level):
        P = self.sizes(level)[1]
        d, ns = self.dim, 2 * self.dim
        t = dict(id=np.zeros(P, np.int32), rank=np.zeros(P, np.int32), local=np.zeros(P, np.int32),
                 starts=np.zeros((P, d)), lengths=np.zeros((P, d)), nbr_kind=np.zeros((P, ns), np.int32),
                 nbr=np.zeros((P, ns, 4), np.int32), nbr_orth=np.zeros((P, ns), np.int32),
                 parent=np.zeros(P, np.int32), orth_on_parent=np.zeros(P, np.int32))
        check(lib().te_hier_level_tables(self.h, level, *[_ptr(t[k]) for k in (
            "id", "rank", "local", "starts", "lengths", "nbr_kind", "nbr", "nbr_orth", "parent",
            "orth_on_parent")]))
        return t

    def l2g(self, level):
        out = np.zeros(self.sizes(level)[0], np.int32)
        check(lib().te_hier_level_l2g(self.h, level, _ptr(out)))
        return out

    def cells(self, level):
        return self.sizes(level)[0] * self.n ** self.dim

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.te_hier_destroy(self.h)
            self.h = None


class Vec:
    """Vector<D> on the device (Vector.h:179-321); method names follow the reference."""

    def __init__(self, gmg, level=0, data=None):
        self.gmg, self.level = gmg, level
        self.h = C.c_void_p()
        check(lib().te_vec_create(gmg.h, level, C.byref(self.h)))
        if data is not None:
            self.upload(data)

    @property
    def size(self):
        return lib().te_vec_size(self.h)

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).ravel()
        if a.size != self.size:
            raise ValueError(f"upload: {a.size} values for a vector of {self.size}")
        check(lib().te_vec_upload(self.h, _ptr(a)))

    def download(self):
        out = np.empty(self.size, np.float64)
        check(lib().te_vec_download(self.h, _ptr(out)))
        return out

    def upload_patches(self, first, a):
        """Vector<D>::getLocalData(i) write path for a run of patches"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        nc = self.gmg.hier.n ** self.gmg.hier.dim
        if a.size % nc:
            raise ValueError("upload_patches: not a whole number of patches")
        check(lib().te_vec_upload_patches(self.h, first, a.size // nc, _ptr(a.ravel())))

    def download_patches(self, first, count):
        out = np.empty(count * self.gmg.hier.n ** self.gmg.hier.dim, np.float64)
        check(lib().te_vec_download_patches(self.h, first, count, _ptr(out)))
        return out

    def device_ptr(self):
        return lib().te_vec_device_ptr(self.h)

    def set(self, alpha): check(lib().te_vec_set(self.h, alpha))
    def scale(self, alpha): check(lib().te_vec_scale(self.h, alpha))
    def shift(self, delta): check(lib().te_vec_shift(self.h, delta))
    def copy(self, b): check(lib().te_vec_copy(self.h, b.h))
    def add(self, b): check(lib().te_vec_add(self.h, b.h))

    def addScaled(self, alpha, a, beta=None, b=None):
        if b is None:
            check(lib().te_vec_add_scaled(self.h, alpha, a.h))
        else:
            check(lib().te_vec_add_scaled2(self.h, alpha, a.h, beta, b.h))

    def scaleThenAdd(self, alpha, b): check(lib().te_vec_scale_then_add(self.h, alpha, b.h))

    def scaleThenAddScaled(self, alpha, beta, b, gamma=None, c=None):
        if c is None:
            check(lib().te_vec_scale_then_add_scaled(self.h, alpha, beta, b.h))
        else:
            check(lib().te_vec_scale_then_add_scaled2(self.h, alpha, beta, b.h, gamma, c.h))

    def twoNormSqLocal(self):
        out = C.c_double()
        check(lib().te_vec_two_norm_sq(self.h, C.byref(out)))
        return out.value

    def twoNorm(self):
        return float(np.sqrt(self.twoNormSqLocal()))

    def infNorm(self):
        out = C.c_double()
        check(lib().te_vec_inf_norm(self.h, C.byref(out)))
        return out.value

    def dot(self, b):
        out = C.c_double()
        check(lib().te_vec_dot(self.h, b.h, C.byref(out)))
        return out.value

    def checksumLocal(self):
        """this rank's part of te_vec_checksum (sum modulo 2^64 of the values' bit patterns); see combine_checksums"""
        out = C.c_uint64()
        check(lib().te_vec_checksum(self.h, C.byref(out)))
        return int(out.value)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None and getattr(self.gmg, "h", None):
            _lib.te_vec_destroy(self.h)
            self.h = None


class GMG:
    """Device-resident level stack: the product of CycleFactory3d::getCycle (CycleFactory3d.cpp:69-134).

    apply / smooth / restrict / interpolate / cycle carry the reference plugin names
    (Operator::apply, Smoother::smooth, Restrictor::restrict, Interpolator::interpolate,
    Cycle::apply)."""

    def __init__(self, hier, device=-1):
        self.hier = hier
        self.h = C.c_void_p()
        check(lib().te_gmg_create(hier.h, device, C.byref(self.h)))
        self.num_levels = lib().te_gmg_num_levels(self.h)
        self._cb = None

    @staticmethod
    def default_opts(**kw):
        o = CycleOpts()
        lib().te_cycle_opts_default(C.byref(o))
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        return o

    def new_vector(self, level=0, data=None):
        return Vec(self, level, data)

    def sync(self):
        check(lib().te_gmg_sync(self.h))

    def stream(self):
        return lib().te_gmg_stream(self.h)

    def apply(self, u, f, level=0): check(lib().te_apply(self.h, level, u.h, f.h))
    def residual(self, u, f, r, level=0): check(lib().te_residual(self.h, level, u.h, f.h, r.h))
    def residual_norm_sq(self, u, f, r, level=0):
        """r = f - A u and this rank's part of ||r||^2, formed by the residual kernel itself"""
        out = C.c_double()
        check(lib().te_residual_norm_sq(self.h, level, u.h, f.h, r.h, C.byref(out)))
        return out.value

    def patch_apply(self, u, f, level=0): check(lib().te_patch_apply(self.h, level, u.h, f.h))
    def verify_schedule(self, opts): check(lib().te_gmg_verify_schedule(self.h, C.byref(opts)))

    def autotune(self, opts, reps=10):
        """te_gmg_autotune (collective): -> (ms per cycle of the chosen form, report line)"""
        ms, buf = C.c_double(), C.create_string_buffer(1024)
        check(lib().te_gmg_autotune(self.h, C.byref(opts), reps, C.byref(ms), buf, 1024))
        return ms.value, buf.value.decode()

    def use_push(self, enable=True):
        """the direct-store transport (te_gmg_use_push): collective"""
        check(lib().te_gmg_use_push(self.h, int(bool(enable))))

    def push_failed(self):
        """0, or the first failure's code of the direct-store transport (1 a wait gave up, 2 a peer's flag two exchanges
        ahead, 3 a peer behind when its buffer was overwritten, 4 epochs out of sequence)"""
        return int(lib().te_gmg_push_failed(self.h))

    def set_patch_bcgs(self, tol=1e-12, max_it=1000):
        """BiCGStabSolver(op, tol, max_it), PatchSolvers/BiCGStabSolver.h:103-108"""
        check(lib().te_gmg_set_patch_bcgs(self.h, float(tol), int(max_it)))

    def patch_bcgs_iterations(self, level, n_local):
        import numpy as np
        its = np.zeros(max(int(n_local), 1), dtype=np.int32)
        check(lib().te_gmg_patch_bcgs_iterations(self.h, level, its.ctypes.data_as(_P)))
        return its[:n_local]

    def comm_info(self):
        """(ranks, rank) of the native RCCL communicator, (0, -1) without one"""
        a, b = C.c_int(), C.c_int()
        check(lib().te_gmg_comm_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_option(self, name, value="1"):
        """one TE_* switch of this solver (they are read from the environment once, at creation); None clears it"""
        check(lib().te_gmg_set_option(self.h, name.encode(), None if value is None else str(value).encode()))

    def release_workspace(self):
        """hand te_bicgstab's work vectors back (8 x a level-0 vector)"""
        check(lib().te_gmg_release_workspace(self.h))

    def set_allreduce(self, fn):
        """fn(list of floats, op) -> list of floats (op 0 sum, 1 max), the same on every rank"""
        def cb(user, vals, n, op):
            try:
                out = fn([vals[i] for i in range(n)], op)
                for i in range(n):
                    vals[i] = out[i]
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._ar = ALLREDUCE_FN(cb)
        check(lib().te_gmg_set_allreduce(self.h, self._ar, None))

    def smooth(self, f, u, level=0, smoother=SMOOTH_PATCH_SOLVE, omega=6.0 / 7.0, sweeps=1):
        check(lib().te_smooth(self.h, level, f.h, u.h, smoother, omega, sweeps))

    def restrict(self, coarse, fine, fine_level=0): check(lib().te_restrict(self.h, fine_level, fine.h, coarse.h))
    def interpolate(self, coarse, fine, fine_level=0): check(lib().te_prolong_add(self.h, fine_level, coarse.h, fine.h))
    def cycle(self, opts, f, u): check(lib().te_vcycle(self.h, C.byref(opts), f.h, u.h))

    def bicgstab(self, x, b, opts=None, max_it=1000, tol=1e-12):
        its, rr = C.c_int(), C.c_double()
        check(lib().te_bicgstab(self.h, C.byref(opts) if opts is not None else None, x.h, b.h, max_it, tol,
                                C.byref(its), C.byref(rr)))
        return its.value, rr.value

    def init_problem(self, f, exact=None, problem=0, neumann=False, level=0):
        """Init::initDirichlet / initNeumann for the canned problems (0 trig, 1 gauss, 2 random rhs), on the device"""
        check(lib().te_init_problem(self.h, level, problem, int(neumann), f.h, exact.h if exact is not None else None))

    def integrate(self, v, level=0):
        """Domain<D>::integrate (Domain.h:258-278), this rank's part"""
        out = C.c_double()
        check(lib().te_integrate(self.h, level, v.h, C.byref(out)))
        return out.value

    def volume(self, level=0):
        """Domain<D>::volume (Domain.h:237-251), this rank's part"""
        out = C.c_double()
        check(lib().te_volume(self.h, level, C.byref(out)))
        return out.value

    def setup_ms(self):
        """where te_gmg_create spent its time (include/te_hip.h te_gmg_setup_ms), by name"""
        v = (C.c_double * 8)()
        check(lib().te_gmg_setup_ms(self.h, v, 8))
        names = ("context_streams", "host_tables", "device_allocations", "device_allocation_count", "table_uploads", "work_vectors", "final_sync", "total")
        return {k: (int(x) if k.endswith("count") else round(float(x), 3)) for k, x in zip(names, v)}

    def profile(self, enable=True): check(lib().te_gmg_profile(self.h, int(enable)))
    def profile_reset(self): check(lib().te_gmg_profile_reset(self.h))
    def profile_select(self, name=None): check(lib().te_gmg_profile_select(self.h, (name or "").encode()))
    def profile_stride(self, stride=1): check(lib().te_gmg_profile_stride(self.h, int(stride)))

    def profile_rows(self):
        names = (C.c_char * 64 * 64)()
        calls = (C.c_int64 * 64)()
        ms = (C.c_double * 64)()
        cells = (C.c_int64 * 64)()
        n = lib().te_gmg_profile_rows(self.h, 64, names, calls, ms, cells)
        if n < 0:
            check(n)
        return {bytes(names[i]).split(b"\0")[0].decode(): dict(calls=calls[i], ms=ms[i], cells=cells[i])
                for i in range(n)}

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.te_gmg_destroy(self.h)
            self.h = None
