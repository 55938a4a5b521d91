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


def independent_levels(m, H):
    """The oracle's level tables WITHOUT the product's: the tree's node table (pinned node for node to the reference's Tree<D> by
    tests/golden/ref_tree_*.npz) through oracle/levels_bfs.py, the breadth-first walk of ThundereggDomGen.h:127-222 restated in
    Python. Only the ORDER of the patches inside a level is taken from the hierarchy (the implementation's choice, not semantics): a
    neighbour, parent or orthant the product's csrc/mesh.cpp got wrong no longer cancels between the two sides of a parity test."""
    from oracle import levels_bfs
    nodes = m.nodes()
    nl = H.num_levels
    tabs = levels_bfs.tables_in_order(levels_bfs.extract_levels(nodes, H.dim)[:nl], nodes, H.dim, [H.tables(l)["id"] for l in range(nl)])
    return [orc.Level.from_tables(t, H.dim, H.n, H.neumann) for t in tabs]


def setup(name, n, divides=0, neumann=False, dim=3, independent=True, **kw):
    """mesh, hierarchy and the oracle's levels. independent (default): the levels come from oracle/levels_bfs.py, not from the
    hierarchy under test (round 5 review, parity 1c); False: orc.levels_from_hierarchy (hierarchies the walk does not model:
    patches_per_proc cut-offs)"""
    m = mesh(name, divides, dim)
    H = capi.Hierarchy(m, n, neumann=neumann, **kw)
    if independent and not kw.get("patches_per_proc"):
        return m, H, independent_levels(m, H)
    return m, H, orc.levels_from_hierarchy(H)


def rand_vec(size, seed):
    return np.random.default_rng(seed).uniform(-1, 1, size)


def op_tol(level, u):
    """Backward-error bound for one operator application: a few ulps of sum|coef||u| =
    (4*dim / h_min^2) * max|u|."""
    hmin = level.a["h"].min()
    return 32 * EPS * 4 * level.dim / hmin ** 2 * max(np.abs(u).max(), 1e-300)
