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
