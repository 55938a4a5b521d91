"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

from oracle import oracle as orc
from pressurepoissonsolver_amd import capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPS = np.finfo(np.float64).eps


def mesh(name, divides=0, dim=3):
    """name: a fixture file in tests/golden (copied data files of the reference's test/ and
    apps/*/meshes), or 'uniform' for a synthesised 1-node tree."""
    if name == "uniform":
        m = capi.Mesh.unit_root(dim)
    else:
        m = capi.Mesh.read(os.path.join(GOLDEN, name), dim)
    for _ in range(divides):
        m.refine_leaves()
    return m


def setup(name, n, divides=0, neumann=False, dim=3, **kw):
    m = mesh(name, divides, dim)
    H = capi.Hierarchy(m, n, neumann=neumann, **kw)
    return m, H, orc.levels_from_hierarchy(H)


def rand_vec(size, seed):
    return np.random.default_rng(seed).uniform(-1, 1, size)


def op_tol(level, u):
    """Backward-error bound for one operator application: a few ulps of sum|coef||u| =
    (4*dim / h_min^2) * max|u|."""
    hmin = level.a["h"].min()
    return 32 * EPS * 4 * level.dim / hmin ** 2 * max(np.abs(u).max(), 1e-300)
