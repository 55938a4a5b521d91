"""CPU: host mesh model (octree reader, refineLeaves, level extraction, Morton partition) through
the C ABI — against the reference's Tree<D> node tables (golden, from oracle/gen_golden.py), the
leaf counts of SURVEY Appendix B, and a brute-force geometric adjacency check."""
import glob
import itertools
import os

import numpy as np
import pytest

from pressurepoissonsolver_amd import capi
from tests import util


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(util.GOLDEN, "ref_tree_*.npz"))),
                         ids=lambda p: os.path.basename(p)[9:-4])
def test_tree_matches_reference(path):
    d = np.load(path)
    m = util.mesh(str(d["mesh"]), int(d["divides"]), int(d["dim"]))
    nodes = m.nodes()
    assert m.num_levels == int(d["num_levels"])
    for k in ("ilp", "lengths", "starts", "nbr", "child"):
        assert np.array_equal(nodes[k], d[k]), k  # ids, links and geometry identical, node for node


@pytest.mark.parametrize("name,dim,counts", [("1uni.bin", 3, [1]), ("2uni.bin", 3, [8, 1]), ("3uni.bin", 3, [64, 8, 1]),
                                             ("2refine.bin", 3, [15, 8, 1]), ("2d2uni.bin", 2, [4, 1]),
                                             ("2d2ref.bin", 2, [7, 4, 1])])
def test_level_patch_counts(name, dim, counts):
    m, H, _ = util.setup(name, 4, dim=dim)
    assert [H.sizes(l)[1] for l in range(H.num_levels)] == counts


def test_divide_and_truncation():
    m = util.mesh("uniform", 3)
    H = capi.Hierarchy(m, 8)
    assert [H.sizes(l)[1] for l in range(H.num_levels)] == [512, 64, 8, 1]
    assert capi.Hierarchy(m, 8, max_levels=2).num_levels == 2  # CycleOpts.h:55 max_levels
    # CycleFactory3d.cpp:104: stop before a level with fewer than patches_per_proc patches per rank
    H4 = capi.Hierarchy(m, 8, patches_per_proc=2, nranks=4)
    assert [H4.sizes(l)[1] for l in range(H4.num_levels)] == [512, 64, 8]


def _touching(t, p, q, dim):
    """side of p on which q touches it with positive (dim-1)-area, else None"""
    s0, l0, s1, l1 = t["starts"][p], t["lengths"][p], t["starts"][q], t["lengths"][q]
    for ax in range(dim):
        for up in (0, 1):
            face = s0[ax] + (l0[ax] if up else 0.0)
            other = s1[ax] + (0.0 if up else l1[ax])
            if abs(face - other) > 1e-12:
                continue
            ok = all(min(s0[a] + l0[a], s1[a] + l1[a]) - max(s0[a], s1[a]) > 1e-12 for a in range(dim) if a != ax)
            if ok:
                return 2 * ax + up
    return None


@pytest.mark.parametrize("name,dim,div", [("2refine.bin", 3, 0), ("2refine.bin", 3, 1), ("3uni.bin", 3, 0),
                                          ("2d2ref.bin", 2, 1)])
def test_neighbour_tables_match_geometry(name, dim, div):
    m, H, _ = util.setup(name, 4, div, dim=dim)
    for lvl in range(H.num_levels):
        t = H.tables(lvl)
        P = len(t["id"])
        want = [[set() for _ in range(2 * dim)] for _ in range(P)]
        for p, q in itertools.permutations(range(P), 2):
            s = _touching(t, p, q, dim)
            if s is not None:
                want[p][s].add(q)
        for p in range(P):
            for s in range(2 * dim):
                kind = t["nbr_kind"][p, s]
                got = set(int(x) for x in t["nbr"][p, s] if x >= 0)
                assert got == want[p][s], (lvl, p, s)
                if not got:
                    assert kind == 0
                    continue
                q = next(iter(got))
                ratio = t["lengths"][q, 0] / t["lengths"][p, 0]
                assert kind == (1 if ratio == 1 else 2 if ratio == 2 else 3)
                if kind == 2:  # quadrant of p on the coarse neighbour's face
                    axes = [a for a in range(dim) if a != s // 2]
                    quad = sum(((t["starts"][p, a] - t["starts"][q, a]) > 1e-12) << i for i, a in enumerate(axes))
                    assert t["nbr_orth"][p, s] == quad
                if kind == 3:  # fine neighbours listed in quadrant order
                    axes = [a for a in range(dim) if a != s // 2]
                    for qi, fq in enumerate(t["nbr"][p, s][:1 << (dim - 1)]):
                        quad = sum(((t["starts"][fq, a] - t["starts"][p, a]) > 1e-12) << i for i, a in enumerate(axes))
                        assert quad == qi


@pytest.mark.parametrize("name,dim,div", [("2refine.bin", 3, 1), ("3uni.bin", 3, 0), ("2d2ref.bin", 2, 0)])
def test_parent_links(name, dim, div):
    m, H, _ = util.setup(name, 4, div, dim=dim)
    for lvl in range(H.num_levels - 1):
        f, c = H.tables(lvl), H.tables(lvl + 1)
        for p in range(len(f["id"])):
            par, o = f["parent"][p], f["orth_on_parent"][p]
            if o < 0:
                assert f["id"][p] == c["id"][par] and np.array_equal(f["starts"][p], c["starts"][par])
            else:
                off = [(o >> a) & 1 for a in range(dim)]
                assert np.allclose(f["starts"][p], c["starts"][par] + np.array(off) * f["lengths"][p])
                assert np.allclose(2 * f["lengths"][p], c["lengths"][par])


@pytest.mark.parametrize("replicate", [1, 0], ids=["replicated", "rank0"])
@pytest.mark.parametrize("agg,cap", [(0, 64), (16, 64), (16, 8), (64, 64)])
@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_morton_partition(nranks, agg, cap, replicate, monkeypatch):
    """agg = patches per rank below which a level (and every coarser one) is gathered (TE_AGGLOMERATE, default 64; SURVEY
    8(e), CycleFactory3d.cpp:104 semantics); 0 = never. cap = the largest level (patches in total) that may be the first
    gathered one (TE_AGGLOMERATE_MAX, default 64): the per-rank threshold alone grows with the number of ranks. Gathered =
    on EVERY rank (TE_REPLICATE, the default in 3D: each rank holds and computes the whole level) or on rank 0 alone."""
    monkeypatch.setenv("TE_AGGLOMERATE", str(agg))
    monkeypatch.setenv("TE_AGGLOMERATE_MAX", str(cap))
    monkeypatch.setenv("TE_REPLICATE", str(replicate))
    m = util.mesh("uniform", 3)  # 8^3 patches
    hs = [capi.Hierarchy(m, 4, rank=r, nranks=nranks) for r in range(nranks)]
    gathered = False
    for lvl in range(hs[0].num_levels):
        t = hs[0].tables(lvl)
        P = len(t["id"])
        counts = np.bincount(t["rank"], minlength=nranks)
        gathered = gathered or (lvl > 0 and P < agg * nranks and P <= cap)
        if gathered and replicate:
            for r, h in enumerate(hs):  # the whole level on every rank, in global order, and each rank's tables say "mine"
                assert h.replicated(lvl) and not h.replicated(0)
                assert np.array_equal(h.l2g(lvl), np.arange(P))
                tr = h.tables(lvl)
                assert np.all(tr["rank"] == r) and np.array_equal(tr["local"], np.arange(P))
            continue
        if gathered:
            assert counts[0] == P  # the whole level on rank 0
        elif P >= nranks:
            assert counts.max() - counts.min() <= 0, (lvl, counts)  # uniform tree: perfectly balanced
        owned = np.concatenate([h.l2g(lvl) for h in hs])
        assert sorted(owned) == list(range(P))  # every patch owned exactly once
        for r, h in enumerate(hs):
            assert np.all(t["rank"][h.l2g(lvl)] == r)
            assert np.array_equal(t["local"][h.l2g(lvl)], np.arange(len(h.l2g(lvl))))
        Pn = len(hs[0].tables(lvl + 1)["id"]) if lvl + 1 < hs[0].num_levels else 0
        nxt = lvl + 1 < hs[0].num_levels and not (gathered or (Pn < agg * nranks and Pn <= cap))
        if nxt:  # a coarse patch lives where its orthant-0 child lives (above the gathered levels)
            c = hs[0].tables(lvl + 1)
            for p in range(P):
                if t["orth_on_parent"][p] <= 0:
                    assert c["rank"][t["parent"][p]] == t["rank"][p]
    # 2x2x2 blocks of ranks for 8 ranks: each rank's finest patches form one octant
    if nranks == 8:
        t = hs[0].tables(0)
        for r in range(8):
            s = t["starts"][t["rank"] == r]
            assert np.all(s.max(0) - s.min(0) < 0.5)


DEEP = [("multi_refine.bin", 3, 0), ("multi_refine_8.bin", 3, 0), ("2d_multi_refine_8.bin", 2, 0), ("2refine.bin", 3, 1),
        ("2d2ref.bin", 2, 2), ("multi_refine.bin", 3, 1)]


@pytest.mark.parametrize("name,dim,div", DEEP)
def test_level_tables_equal_independent_breadth_first_extraction(name, dim, div):
    """Every level table of the product (csrc/mesh.cpp: membership rule + Morton sort) against a second, independent
    statement of the reference's level extraction -- the breadth-first walk of ThundereggDomGen.h:127-222 in plain Python
    (oracle/levels_bfs.py) -- table for table: patch set, geometry, neighbour kinds / ids / quadrants, parents and orthants.
    Deep AMR trees included (apps/3d/meshes/multi_refine.bin: 137 nodes, 5 levels; multi_refine_8.bin: 9 levels; the mesh of
    apps/3d/config/gmg_example.ini)."""
    from oracle import levels_bfs
    m = util.mesh(name, div, dim)
    H = capi.Hierarchy(m, 4)
    nodes = m.nodes()
    levels = levels_bfs.extract_levels(nodes, dim)
    assert len(levels) == H.num_levels
    prod = [H.tables(l) for l in range(H.num_levels)]
    want = levels_bfs.tables_in_order(levels, nodes, dim, [t["id"] for t in prod])
    for l, (t, w) in enumerate(zip(prod, want)):
        for k in ("id", "starts", "lengths", "nbr_kind", "nbr", "nbr_orth", "orth_on_parent"):
            assert np.array_equal(t[k], w[k]), (l, k)
        if l + 1 < H.num_levels:
            assert np.array_equal(t["parent"], w["parent"]), (l, "parent")


@pytest.mark.parametrize("name,dim,div", DEEP)
def test_level_table_properties_deep_trees(name, dim, div):
    """Per level: every coarse patch has all its children (or is a copy of one fine patch), neighbour links are symmetric
    (normal <-> normal, coarse <-> fine with matching quadrant), refinement is 2:1 across every face."""
    m = util.mesh(name, div, dim)
    H = capi.Hierarchy(m, 4)
    for l in range(H.num_levels):
        t = H.tables(l)
        P = len(t["id"])
        for p in range(P):
            for s in range(2 * dim):
                k = t["nbr_kind"][p, s]
                if k == 1:
                    q = t["nbr"][p, s, 0]
                    assert t["nbr_kind"][q, s ^ 1] == 1 and t["nbr"][q, s ^ 1, 0] == p
                    assert np.allclose(t["lengths"][q], t["lengths"][p])
                elif k == 2:  # my neighbour is coarser: exactly twice my size, and it lists me in my quadrant
                    q, quad = t["nbr"][p, s, 0], t["nbr_orth"][p, s]
                    assert np.allclose(t["lengths"][q], 2 * t["lengths"][p])
                    assert t["nbr_kind"][q, s ^ 1] == 3 and t["nbr"][q, s ^ 1, quad] == p
                elif k == 3:
                    for quad in range(1 << (dim - 1)):
                        q = t["nbr"][p, s, quad]
                        assert np.allclose(2 * t["lengths"][q], t["lengths"][p])
                        assert t["nbr_kind"][q, s ^ 1] == 2 and t["nbr"][q, s ^ 1, 0] == p and t["nbr_orth"][q, s ^ 1] == quad
        if l + 1 < H.num_levels:
            c = H.tables(l + 1)
            kids = [[] for _ in range(len(c["id"]))]
            for p in range(P):
                kids[t["parent"][p]].append(t["orth_on_parent"][p])
            for pc, ks in enumerate(kids):
                assert sorted(ks) == list(range(1 << dim)) or ks == [-1], (l, pc, ks)
