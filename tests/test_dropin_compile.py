"""CPU, build container only: the C++ adaptors (pressurepoissonsolver_amd/thunderegg/HipGMG.h) compile
against the reference's own headers and satisfy its plugin interfaces; the reference's BiCGStab<3>
instantiates over them (syntax + semantic check, nothing is run)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("THUNDEREGG_REF", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "Thunderegg")), reason="reference tree not present")
def test_adaptors_compile_against_reference_headers():
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-w", "-I" + os.path.join(REF, "src"), "-I/opt/conda/include",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pressurepoissonsolver_amd", "thunderegg"),
           os.path.join(ROOT, "tests", "dropin_compile.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "Thunderegg")), reason="reference tree not present")
def test_dropin_program_links_and_fails_loudly_without_a_gpu():
    """The drop-in test program (tests/dropin_run.cpp: the reference's BiCGStab<3> + Vector.cpp over the adaptors) is
    really linked against libte_hip.so; without a GPU the adaptors turn TE_EHIP into the reference's `throw 3`."""
    import torch
    from oracle import build as ob
    exe = ob.build_dropin()
    assert exe and os.path.exists(exe)
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_gpu_dropin.py runs the program for real")
    env = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu")
    r = subprocess.run([exe, "uniform", "1", "8", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 1 and "reference-style exception 3" in r.stderr and "no HIP device" in r.stderr
