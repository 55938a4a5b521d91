"""-m gpu: the drop-in layer EXECUTES. oracle/_ref/dropin_run (tests/dropin_run.cpp, built in the build container
by oracle/build.py: build_dropin against /root/reference/src + libte_hip.so; only the binary travels) runs the
reference's own BiCGStab<3>::solve (BiCGStab.h:45-106; the call of apps/3d/steady.cpp:519-524) over
HipVG / HipOperator / HipCycle with the right-hand side filled through Vector<3>::getLocalData as
Init::initDirichlet does (Init.cpp:152-245), a level-by-level V-cycle through the four plugin interfaces in the
order of GMG/Cycle.h:56-90, and the getLocalData aliasing rules. Fresh child process per case."""
import os
import subprocess

import pytest

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "dropin_run")

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/dropin_run was not built (needs the reference tree at build time)")
@pytest.mark.parametrize("mesh,div,n,smoother", [("uniform", 2, 8, 2), ("uniform", 2, 8, 0), ("2refine.bin", 1, 8, 0),
                                                 ("uniform", 2, 32, 2), ("uniform", 3, 32, 0), ("2refine.bin", 0, 32, 2)])
def test_reference_bicgstab_runs_over_the_adaptors(mesh, div, n, smoother):
    path = mesh if mesh == "uniform" else os.path.join(util.GOLDEN, mesh)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join(p for p in ("/usr/lib/x86_64-linux-gnu", env.get("LD_LIBRARY_PATH", "")) if p)
    r = subprocess.run([BIN, path, str(div), str(n), str(smoother)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/dropin_run was not built (needs the reference tree at build time)")
@pytest.mark.parametrize("mesh,div,n,smoother,its_expected", [
    ("uniform", 0, 256, 0, 1),      # config C1: one 256^2 patch, the exact patch solve as smoother: the cycle IS the solve, 1 iteration
    ("2d2ref.bin", 0, 8, 0, None),  # refined quadtree (apps/2d/meshes/2d2ref.bin): coarse/fine edges, BilinearInterpolator weights
    ("2d2ref.bin", 1, 8, 2, None),
    ("uniform", 3, 64, 2, None),    # 64 patches of 64^2 (config C5's patch size), RB-GS: the fused LDS kernels behind HipCycle<2>
    ("uniform", 3, 64, 0, None),    # ... and the reference's block-Jacobi smoother on the matrix cores
    ("2d2ref.bin", 1, 8, 3, None),  # --patch_solver bcgs (apps/2d/steady.cpp:326-327): HipSmoother<2>(TE_SMOOTH_PATCH_BCGS)
    ("uniform", 2, 64, 3, None)])
def test_reference_bicgstab_runs_over_the_2d_adaptors(mesh, div, n, smoother, its_expected):
    """the D = 2 half of the boundary (apps/2d/steady.cpp:322-331, 494, 523, 563-568): HipVector<2> / HipVG<2> / HipOperator<2> /
    HipCycle<2> / HipSmoother<2> / HipRestrictor<2> / HipInterpolator<2> under the reference's own BiCGStab<2>::solve, the
    right-hand side through Init::initDirichlet2d's twin (HipInit.h) and Vector<2>::getLocalData; iteration count equal to
    te_bicgstab's (checked by the program), level-by-level cycle == te_vcycle(fuse = 0) bit for bit."""
    path = mesh if mesh == "uniform" else os.path.join(util.GOLDEN, mesh)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join(p for p in ("/usr/lib/x86_64-linux-gnu", env.get("LD_LIBRARY_PATH", "")) if p)
    r = subprocess.run([BIN, path, str(div), str(n), str(smoother), "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DROPIN_OK dim=2" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    if its_expected is not None:
        assert f"its={its_expected} native_its={its_expected}" in r.stdout, r.stdout[-1000:]
