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


def build_hip(force=False):
    srcs = [os.path.join(CSRC, f) for f in ("gmg.hip", "capi_mesh.cpp", "mesh.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    deps.append(os.path.join(ROOT, "include", "te_hip.h"))
    if not force and not _newer(LIB_HIP, deps):
        return LIB_HIP
    _run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
          "-Wno-unused-result", "-Wl,-soname,libte_hip.so", "-x", "hip", srcs[0], "-x", "hip", srcs[1], "-x", "hip", srcs[2],
          "-o", LIB_HIP])
    return LIB_HIP


def build_variant(name, defines):
    """tooling: a second library with other compile-time switches (e.g. build_variant("noskew", ["-DTE_LDS_SKEW=0"])) for
    same-box A/B runs: TE_HIP_LIB_PATH=<returned path> python tools/variant_bench.py ..."""
    out = os.path.join(ROOT, "pressurepoissonsolver_amd", f"libte_hip_{name}.so")
    srcs = [os.path.join(CSRC, f) for f in ("gmg.hip", "capi_mesh.cpp", "mesh.cpp")]
    _run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-result"] + list(defines)
         + ["-Wl,-soname," + os.path.basename(out), "-x", "hip", srcs[0], "-x", "hip", srcs[1], "-x", "hip", srcs[2], "-o", out])
    return out


def build_all(force=False):
    return {"hip": build_hip(force)}


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
