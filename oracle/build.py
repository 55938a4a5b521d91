"""TEST INFRASTRUCTURE. Builds the checker libraries:

  oracle/libte_oracle.so     g++: CPU restatement of the reference algorithm (te_oracle.cpp)
  oracle/_ref/libte_ref.so   g++ over /root/reference sources (Makefile.ref), only when that tree
                             is present (never on the GPU box, which uses the prebuilt file)
  oracle/_ref/dropin_run     the reference's BiCGStab<3> + Vector<3> over the product's C++ adaptors, linked
                             against libte_hip.so (tests/dropin_run.cpp), same condition
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


def build_dropin(force=False):
    """tests/dropin_run.cpp: the reference's own BiCGStab<3> (BiCGStab.h) and Vector<3> (Vector.cpp), compiled from where
    they lie, over the product's C++ adaptors and linked against libte_hip.so -> oracle/_ref/dropin_run (git-ignored,
    travels with the snapshot; run by tests/test_gpu_dropin.py on the GPU box). Only where the reference tree is."""
    root = os.path.dirname(ORACLE_DIR)
    out = os.path.join(ORACLE_DIR, "_ref", "dropin_run")
    src = os.path.join(REF_ROOT, "src")
    if not os.path.isdir(os.path.join(src, "Thunderegg")):
        return out if os.path.exists(out) else None
    hip = os.path.join(root, "pressurepoissonsolver_amd", "libte_hip.so")
    adapt = os.path.join(root, "pressurepoissonsolver_amd", "thunderegg")
    deps = [os.path.join(root, "tests", "dropin_run.cpp"), os.path.join(adapt, "HipGMG.h"), os.path.join(adapt, "HipInit.h"),
            os.path.join(root, "include", "te_hip.h"), hip]
    if not os.path.exists(hip):
        return None
    if not force and not _newer(out, deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    mpi = os.environ.get("MPI_PREFIX", "/opt/conda")
    try:
        # (libraries by path: -L<conda>/lib would put conda's older libstdc++ in front of the system's)
        _run(["g++", "-std=c++11", "-O1", "-w", "-I" + src, "-I" + os.path.join(mpi, "include"),
              "-I" + os.path.join(root, "include"), "-I" + adapt, deps[0], os.path.join(src, "Thunderegg", "Vector.cpp"),
              hip, os.path.join(mpi, "lib", "libmpi.so"), "-Wl,-rpath,/usr/lib/x86_64-linux-gnu",
              "-Wl,-rpath,$ORIGIN/../../pressurepoissonsolver_amd", "-Wl,-rpath,/opt/rocm/lib",
              "-Wl,-rpath," + os.path.join(mpi, "lib"), "-Wl,-rpath-link,/opt/rocm/lib", "-o", out])
    except subprocess.CalledProcessError as e:  # optional tooling
        print("warning: drop-in program build failed:", e, file=sys.stderr)
        return None
    return out


if __name__ == "__main__":
    print(build_oracle("--force" in sys.argv), build_ref("--force" in sys.argv), build_dropin("--force" in sys.argv))
