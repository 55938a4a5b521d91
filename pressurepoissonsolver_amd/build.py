"""Builds the product library in-tree: libte_hip.so from pressurepoissonsolver_amd/csrc with
hipcc --offload-arch=gfx950. The output is git-ignored but travels with the gpurun snapshot.
(The test-only checker libraries are built by the checker's own build script, not from here.)
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pressurepoissonsolver_amd", "csrc")
LIB_HIP = os.path.join(ROOT, "pressurepoissonsolver_amd", "libte_hip.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the product library cannot be built")


# translation units of libte_hip.so (gmg_internal.hpp says what lives where); compiled in parallel, one object file each
UNITS = ("gmg_core.hip", "gmg_transport.hip", "gmg_launch3d.hip", "gmg_fused3d.hip", "gmg_patchsolve.hip", "gmg_launch2d.hip", "gmg_cycle.hip", "gmg_krylov.hip", "capi_mesh.cpp", "mesh.cpp")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-Wno-unused-function"]


def _compile_link(out, defines=()):
    """every unit to an object file under csrc/../build/<library name>/ (as many at once as there are CPUs), then one link"""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(ROOT, "pressurepoissonsolver_amd", "build", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()

    def one(unit):
        obj = os.path.join(objdir, os.path.splitext(unit)[0] + ".o")
        cmd = [cc] + CFLAGS + list(defines) + ["-c", "-x", "hip", os.path.join(CSRC, unit), "-o", obj]
        print("+", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{unit}: hipcc failed\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, UNITS))
    # linked under another name and moved into place: nobody ever dlopens a half-written library
    tmp = f"{out}.tmp.{os.getpid()}"
    _run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-soname," + os.path.basename(out)] + objs + ["-o", tmp])
    os.replace(tmp, out)
    return out


class _BuildLock:
    """one builder at a time per checkout (the ranks of a torchrun job all import capi at once; the object files have fixed
    names): an exclusive flock on build/.lock around `is it stale? -> compile -> link`; the others wait and then find it fresh"""

    def __enter__(self):
        import fcntl
        d = os.path.join(ROOT, "pressurepoissonsolver_amd", "build")
        os.makedirs(d, exist_ok=True)
        self.f = open(os.path.join(d, ".lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def build_hip(force=False):
    srcs = [os.path.join(CSRC, f) for f in UNITS]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    deps.append(os.path.join(ROOT, "include", "te_hip.h"))
    if not force and not _newer(LIB_HIP, deps):
        return LIB_HIP
    with _BuildLock():
        if not force and not _newer(LIB_HIP, deps):  # (another process built it while this one waited)
            return LIB_HIP
        return _compile_link(LIB_HIP)


def build_variant(name, defines):
    """tooling: a second library with other compile-time switches (e.g. build_variant("noskew", ["-DTE_LDS_SKEW=0"])) for
    same-box A/B runs: TE_HIP_LIB_PATH=<returned path> python tools/variant_bench.py ..."""
    with _BuildLock():
        return _compile_link(os.path.join(ROOT, "pressurepoissonsolver_amd", f"libte_hip_{name}.so"), defines)


def build_all(force=False):
    return {"hip": build_hip(force)}


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
