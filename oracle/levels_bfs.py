"""TEST INFRASTRUCTURE. Level extraction restated independently of the product's csrc/mesh.cpp: the breadth-first walk
of ThundereggDomGen<D>::extractLevel (src/Thunderegg/ThundereggDomGen.h:127-222) over the tree's node table, in plain
Python. The product collects a level by a membership rule and a Morton sort; this follows the reference's own
traversal -- start at one node of the tree level, enqueue what its neighbour links reach -- so the two are separate
statements of the same level tables. Input: the node table (ids, levels, parents, neighbour and child ids; itself pinned
node for node to the reference's Tree<D> by tests/golden/ref_tree_*.npz). Only tests/ use this file.

  extract_levels(nodes, dim)            -> [ {id: info} per level, finest first ]  (info: parent_id, orth_on_parent,
                                            sides[s] = ("normal", id) | ("coarse", id, quad) | ("fine", ids) | None)
  tables_in_order(levels, nodes, dim, ids_per_level)
                                         -> per level the arrays of capi.Hierarchy.tables() (id, starts, lengths, nbr_kind,
                                            nbr, nbr_orth, parent, orth_on_parent) for a GIVEN patch order per level (the
                                            order of patches inside a level is the implementation's choice, not semantics)
"""
from collections import deque

import numpy as np


def orthants_on_side(dim, s):
    """Orthant<D>::getValuesOnSide (Side.h:346-362): the orthants whose bit `axis` is the side's, ascending"""
    axis, upper = s // 2, s & 1
    return [o for o in range(1 << dim) if ((o >> axis) & 1) == upper]


def extract_levels(nodes, dim):
    ilp, nbr, child = nodes["ilp"], nodes["nbr"], nodes["child"]
    by_id = {int(ilp[i, 0]): i for i in range(len(ilp))}
    level_of = lambda i: int(ilp[by_id[i], 1])  # noqa: E731
    parent_of = lambda i: int(ilp[by_id[i], 2])  # noqa: E731
    nbr_of = lambda i, s: int(nbr[by_id[i], s])  # noqa: E731
    child_of = lambda i, o: int(child[by_id[i], o])  # noqa: E731
    has_children = lambda i: child_of(i, 0) != -1  # noqa: E731
    num_levels = int(ilp[:, 1].max())
    root_level = int(ilp[:, 1].min())
    out = []
    for curr in range(num_levels, root_level - 1, -1):
        start = max(int(i) for i, l in zip(ilp[:, 0], ilp[:, 1]) if l == curr)  # any node of the tree level (Tree::levels)
        q, qed, level = deque([start]), {start}, {}

        def enqueue(i):
            if i not in qed:
                qed.add(i)
                q.append(i)

        while q:
            n = q.popleft()
            info = dict(parent_id=-1, orth_on_parent=-1, sides=[None] * (2 * dim))
            if level_of(n) < curr:
                info["parent_id"] = n  # ThundereggDomGen.h:150-151: the patch stands for itself on the coarser level
            else:
                info["parent_id"] = parent_of(n)
                if parent_of(n) != -1:
                    o = 0
                    while child_of(parent_of(n), o) != n:
                        o += 1
                    info["orth_on_parent"] = o
            for s in range(2 * dim):
                par = parent_of(n)
                if nbr_of(n, s) == -1 and par != -1 and nbr_of(par, s) != -1:  # :166-179 coarser neighbour
                    cn = nbr_of(par, s)
                    octs = orthants_on_side(dim, s)
                    quad = 0
                    while child_of(par, octs[quad]) != n:
                        quad += 1
                    info["sides"][s] = ("coarse", cn, quad)
                    enqueue(cn)
                elif level_of(n) < curr and nbr_of(n, s) != -1 and has_children(nbr_of(n, s)):  # :180-194 finer neighbours
                    ids = [child_of(nbr_of(n, s), o) for o in orthants_on_side(dim, s ^ 1)]
                    for i in ids:
                        enqueue(i)
                    info["sides"][s] = ("fine", ids)
                elif nbr_of(n, s) != -1:  # :195-202
                    info["sides"][s] = ("normal", nbr_of(n, s))
                    enqueue(nbr_of(n, s))
            level[n] = info
        out.append(level)
    return out


def tables_in_order(levels, nodes, dim, ids_per_level):
    ilp = nodes["ilp"]
    by_id = {int(ilp[i, 0]): i for i in range(len(ilp))}
    tabs = []
    for li, (lvl, ids) in enumerate(zip(levels, ids_per_level)):
        ids = [int(i) for i in ids]
        assert sorted(ids) == sorted(lvl), "the level holds other patches than the breadth-first walk reaches"
        index = {i: p for p, i in enumerate(ids)}
        nxt = {int(i): p for p, i in enumerate(ids_per_level[li + 1])} if li + 1 < len(levels) else None
        P, ns = len(ids), 2 * dim
        t = dict(id=np.array(ids, np.int32), starts=np.zeros((P, dim)), lengths=np.zeros((P, dim)),
                 nbr_kind=np.zeros((P, ns), np.int32), nbr=np.full((P, ns, 4), -1, np.int32),
                 nbr_orth=np.full((P, ns), -1, np.int32), parent=np.full(P, -1, np.int32),
                 orth_on_parent=np.full(P, -1, np.int32))
        for p, i in enumerate(ids):
            info = lvl[i]
            t["starts"][p] = nodes["starts"][by_id[i]]
            t["lengths"][p] = nodes["lengths"][by_id[i]]
            t["orth_on_parent"][p] = info["orth_on_parent"]
            if nxt is not None:
                t["parent"][p] = nxt[info["parent_id"]]
            for s, side in enumerate(info["sides"]):
                if side is None:
                    continue
                if side[0] == "normal":
                    t["nbr_kind"][p, s] = 1
                    t["nbr"][p, s, 0] = index[side[1]]
                elif side[0] == "coarse":
                    t["nbr_kind"][p, s] = 2
                    t["nbr"][p, s, 0] = index[side[1]]
                    t["nbr_orth"][p, s] = side[2]
                else:
                    t["nbr_kind"][p, s] = 3
                    for k, fid in enumerate(side[1]):
                        t["nbr"][p, s, k] = index[fid]
        tabs.append(t)
    return tabs
