"""TEST INFRASTRUCTURE. Builds the checker libraries:

  oracle/libte_oracle.so     g++: CPU restatement of the reference algorithm (te_oracle.cpp)
  oracle/_ref/libte_ref.so   g++ over /root/reference sources (Makefile.ref), only when that tree
                             is present (never on the GPU box, which uses the prebuilt file)
"""
import os
import subprocess
import sys

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
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
        return LIB_REF if os.path.exists(LIB_REF) else None
    try:
        _run(["make", "-s", "-C", ORACLE_DIR, "-f", "Makefile.ref", "THUNDEREGG_REF=" + REF_ROOT]
             + (["-B"] if force else []))
    except subprocess.CalledProcessError as e:  # optional tooling
        print("warning: oracle/_ref build failed:", e, file=sys.stderr)
        return None
    return LIB_REF


if __name__ == "__main__":
    print(build_oracle("--force" in sys.argv), build_ref("--force" in sys.argv))
