"""Builds the in-tree native libraries.

  libte_hip.so   (pressurepoissonsolver_amd/csrc)  hipcc --offload-arch=gfx950: the product
  libte_oracle.so (oracle/)                         g++: CPU restatement, test infrastructure
  oracle/_ref/libte_ref.so                          g++ over /root/reference sources, only when
                                                    that tree is present (never on the GPU box)

Outputs are git-ignored but travel with the gpurun snapshot.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pressurepoissonsolver_amd", "csrc")
LIB_HIP = os.path.join(ROOT, "pressurepoissonsolver_amd", "libte_hip.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_ORACLE = os.path.join(ORACLE_DIR, "libte_oracle.so")
REF_ROOT = os.environ.get("THUNDEREGG_REF", "/root/reference")
LIB_REF = os.path.join(ORACLE_DIR, "_ref", "libte_ref.so")


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
          "-Wno-unused-result", "-x", "hip", srcs[0], "-x", "hip", srcs[1], "-x", "hip", srcs[2],
          "-o", LIB_HIP])
    return LIB_HIP


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, "te_oracle.cpp")]
    deps = srcs + [os.path.join(ORACLE_DIR, "te_oracle.h")]
    if not force and not _newer(LIB_ORACLE, deps):
        return LIB_ORACLE
    _run(["g++", "-O3", "-march=x86-64-v3", "-std=c++14", "-fPIC", "-shared", "-fopenmp", "-Wall",
          srcs[0], "-o", LIB_ORACLE])
    return LIB_ORACLE


def build_ref(force=False):
    """Compile the PETSc-free slice of the reference from where it lies (oracle/Makefile.ref)."""
    if not os.path.isdir(os.path.join(REF_ROOT, "src", "Thunderegg")):
        return None
    mk = os.path.join(ORACLE_DIR, "Makefile.ref")
    if not os.path.exists(mk):
        return None
    _run(["make", "-s", "-C", ORACLE_DIR, "-f", "Makefile.ref", "THUNDEREGG_REF=" + REF_ROOT]
         + (["-B"] if force else []))
    return LIB_REF


def build_all(force=False):
    out = {"hip": build_hip(force), "oracle": build_oracle(force)}
    try:
        out["ref"] = build_ref(force)
    except subprocess.CalledProcessError as e:  # reference slice is optional test tooling
        print("warning: oracle/_ref build failed:", e, file=sys.stderr)
        out["ref"] = None
    return out


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
