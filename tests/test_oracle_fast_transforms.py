"""CPU: the oracle's O(n log n) transforms (own radix-2 FFT; what the reference's default patch solver, FftwPatchSolver.h:93-206, does
through FFTW) against its dense products (DftPatchSolver.h:295-347, the parity oracle): the exact patch solve and a whole V-cycle agree
to 1e-12 for every transform type -- DST-II/III (Dirichlet axes), DCT-II/III (Neumann axes), DST-IV / DCT-IV (a patch with a physical
Neumann face on one side of an axis and a neighbour or Dirichlet face on the other) -- and every patch size the kernels exist for.
The fast path is used by bench.py's `cpu_baseline.runs[smoother = patch_solve_fft]` only."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import util

CASES = [("2uni.bin", 4, 0, False, 3), ("2uni.bin", 8, 0, True, 3), ("2refine.bin", 8, 0, True, 3), ("2uni.bin", 16, 0, True, 3),
         ("uniform", 32, 1, False, 3), ("uniform", 32, 1, True, 3), ("2refine.bin", 32, 0, True, 3),
         ("2d2ref.bin", 8, 0, True, 2), ("uniform", 64, 1, True, 2), ("uniform", 64, 1, False, 2), ("uniform", 256, 0, False, 2)]


@pytest.mark.parametrize("name,n,div,neumann,dim", CASES, ids=lambda c: str(c))
def test_fast_transforms_equal_dense_products(name, n, div, neumann, dim):
    m, H, levels = util.setup(name, n, div, neumann=neumann, dim=dim)
    L = levels[0]
    u = util.rand_vec(L.size, 7)
    f = util.rand_vec(L.size, 8) / L.a["h"].min() ** 2
    try:
        orc.set_fast_transforms(False)
        dense = orc.smooth(L, f, u)
        cyc_dense = orc.cycle(levels, orc.cycle_opts(smoother=0), f) if len(levels) > 1 else None
        orc.set_fast_transforms(True)
        fast = orc.smooth(L, f, u)
        cyc_fast = orc.cycle(levels, orc.cycle_opts(smoother=0), f) if len(levels) > 1 else None
    finally:
        orc.set_fast_transforms(False)
    assert np.linalg.norm(fast - dense) <= 1e-12 * np.linalg.norm(dense)
    if cyc_dense is not None:
        assert np.linalg.norm(cyc_fast - cyc_dense) <= 1e-12 * np.linalg.norm(cyc_dense)
