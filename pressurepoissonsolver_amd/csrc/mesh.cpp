// See mesh.hpp for the reference map. No HIP in this file: it is plain host C++ and is
// exercised by the CPU test-suite through the C ABI (te_mesh_* / te_hier_* in capi.cpp).
#include "mesh.hpp"
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <set>
#include <stdexcept>

namespace te
{
Node::Node()
{
	for (int i = 0; i < MAX_D; i++) {
		lengths[i] = -1;
		starts[i]  = -1;
	}
	for (int i = 0; i < MAX_SIDE; i++) nbr[i] = -1;
	for (int i = 0; i < MAX_ORTH; i++) child[i] = -1;
}

// File layout (little endian), OctTree.h:90-118 / OctNode.h:30-58:
//   int32 num_nodes, num_trees; then per node: int32 id, level, parent;
//   double lengths[D], starts[D]; int32 nbr_id[2D]; int32 child_id[2^D]
Tree Tree::read(const std::string &path, int dim)
{
	if (dim != 2 && dim != 3) throw std::runtime_error("te::Tree::read: dim must be 2 or 3");
	std::ifstream in(path, std::ios_base::binary);
	if (!in) throw std::runtime_error("te::Tree::read: cannot open " + path);
	Tree t;
	t.dim          = dim;
	int32_t hdr[2] = {0, 0};
	in.read(reinterpret_cast<char *>(hdr), 8);
	int num_nodes = hdr[0];
	if (!in || num_nodes <= 0) throw std::runtime_error("te::Tree::read: bad header in " + path);
	const int nsides = 2 * dim, north = 1 << dim;
	for (int i = 0; i < num_nodes; i++) {
		Node    nd;
		int32_t ilp[3];
		in.read(reinterpret_cast<char *>(ilp), 12);
		nd.id     = ilp[0];
		nd.level  = ilp[1];
		nd.parent = ilp[2];
		in.read(reinterpret_cast<char *>(nd.lengths), 8 * dim);
		in.read(reinterpret_cast<char *>(nd.starts), 8 * dim);
		int32_t tmp[8];
		in.read(reinterpret_cast<char *>(tmp), 4 * nsides);
		for (int s = 0; s < nsides; s++) nd.nbr[s] = tmp[s];
		in.read(reinterpret_cast<char *>(tmp), 4 * north);
		for (int o = 0; o < north; o++) nd.child[o] = tmp[o];
		if (!in) throw std::runtime_error("te::Tree::read: truncated file " + path);
		if (i == 0) {
			t.root       = nd.id;
			t.root_level = nd.level;
		}
		t.max_id         = std::max(t.max_id, nd.id);
		t.num_levels     = std::max(t.num_levels, nd.level);
		t.nodes[nd.id]   = nd;
	}
	return t;
}

Tree Tree::unitRoot(int dim)
{
	Tree t;
	t.dim = dim;
	Node nd;
	nd.id     = 0;
	nd.level  = 1;
	nd.parent = -1;
	for (int i = 0; i < dim; i++) {
		nd.lengths[i] = 1.0;
		nd.starts[i]  = 0.0;
	}
	t.nodes[0]   = nd;
	t.root       = 0;
	t.root_level = 1;
	t.num_levels = 1;
	t.max_id     = 0;
	return t;
}

int Tree::depthOf(int id) const
{
	int d = 0;
	for (int p = nodes.at(id).parent; p != -1; p = nodes.at(p).parent) d++;
	return d;
}

// interior / exterior orthant neighbours, Side.h Orthant<D>::getInteriorNbrOnSide /
// getExteriorNbrOnSide: both flip the bit of the side's axis.
static inline int orthFlip(int orth, int side) { return orth ^ (1 << (side / 2)); }
static inline bool orthOnSide(int orth, int side)
{
	return ((orth >> (side / 2)) & 1) == (side & 1);
}

void Tree::refineNode(int id)
{
	const int nsides = 2 * dim, north = 1 << dim;
	Node     &n = nodes.at(id);
	std::vector<Node> kids(north);
	for (int o = 0; o < north; o++) {
		Node &c  = kids[o];
		c.parent = n.id;
		c.level  = n.level + 1;
		for (int i = 0; i < dim; i++) {
			c.lengths[i] = n.lengths[i] / 2;
			c.starts[i]  = ((o >> i) & 1) ? n.starts[i] + c.lengths[i] : n.starts[i];
		}
		c.id       = ++max_id;
		n.child[o] = c.id;
	}
	// siblings
	for (int o = 0; o < north; o++) {
		for (int a = 0; a < dim; a++) {
			int s          = 2 * a + (((o >> a) & 1) ? 0 : 1); // interior side on axis a
			kids[o].nbr[s] = kids[orthFlip(o, s)].id;
		}
	}
	// across the parent's faces, when that neighbour is already split
	for (int s = 0; s < nsides; s++) {
		if (n.nbr[s] == -1) continue;
		Node &nb = nodes.at(n.nbr[s]);
		if (!nb.hasChildren()) continue;
		for (int o = 0; o < north; o++) {
			if (!orthOnSide(o, s)) continue;
			Node &nc       = nodes.at(nb.child[orthFlip(o, s)]);
			kids[o].nbr[s] = nc.id;
			nc.nbr[s ^ 1]  = kids[o].id;
		}
	}
	for (auto &c : kids) nodes[c.id] = c;
}

void Tree::refineLeaves()
{
	std::vector<std::pair<int, int>> leaves; // (depth, id) == the reference's BFS set order
	for (auto &p : nodes) {
		if (!p.second.hasChildren()) leaves.emplace_back(depthOf(p.first), p.first);
	}
	std::sort(leaves.begin(), leaves.end());
	for (auto &l : leaves) refineNode(l.second);
	num_levels++;
}

uint64_t mortonKey(const double *starts, const double *root_starts, const double *root_lengths,
                   int dim, int bits)
{
	uint64_t c[MAX_D] = {0, 0, 0};
	for (int a = 0; a < dim; a++) {
		double rel = (starts[a] - root_starts[a]) / root_lengths[a];
		c[a]       = (uint64_t) std::llround(rel * (double) (1ull << bits));
	}
	uint64_t key = 0;
	for (int b = bits - 1; b >= 0; b--) {
		for (int a = dim - 1; a >= 0; a--) key = (key << 1) | ((c[a] >> b) & 1);
	}
	return key;
}

namespace
{
struct LevelNodes {
	std::vector<int>   ids; // Morton order
	std::map<int, int> index;
};

// Which tree nodes make up the domain of tree level L (ThundereggDomGen.h:127-222): every
// node with level == L, plus every leaf with level < L.
LevelNodes collectLevel(const Tree &t, int L)
{
	const Node &root = t.nodes.at(t.root);
	int         bits = std::max(1, t.num_levels);
	std::vector<std::pair<uint64_t, int>> keyed;
	for (auto &p : t.nodes) {
		const Node &nd = p.second;
		if (nd.level == L || (nd.level < L && !nd.hasChildren())) {
			keyed.emplace_back(mortonKey(nd.starts, root.starts, root.lengths, t.dim, bits), nd.id);
		}
	}
	std::sort(keyed.begin(), keyed.end());
	LevelNodes ln;
	for (auto &k : keyed) {
		ln.index[k.second] = (int) ln.ids.size();
		ln.ids.push_back(k.second);
	}
	return ln;
}
} // namespace

Hierarchy Hierarchy::build(const Tree &t, int n, bool neumann, int max_levels,
                           double patches_per_proc, int rank, int nranks, const Placement &pl)
{
	if (n < 2 || (n & 1)) throw std::runtime_error("te::Hierarchy: n must be even and >= 2");
	if (nranks < 1 || rank < 0 || rank >= nranks) throw std::runtime_error("te::Hierarchy: bad rank");
	Hierarchy h;
	h.dim     = t.dim;
	h.n       = n;
	h.rank    = rank;
	h.nranks  = nranks;
	h.neumann = neumann;
	const int dim = t.dim, nsides = 2 * dim, nq = 1 << (dim - 1);

	std::vector<LevelNodes> lns;
	int                     built = 0;
	for (int L = t.num_levels; L >= t.root_level; L--) {
		if (built > 0) {
			if (max_levels > 0 && built >= max_levels) break;
			LevelNodes probe = collectLevel(t, L);
			if ((probe.ids.size() + 0.0) / nranks < patches_per_proc) break;
			lns.push_back(std::move(probe));
		} else {
			lns.push_back(collectLevel(t, L));
		}
		built++;
	}

	h.levels.resize(lns.size());
	for (size_t li = 0; li < lns.size(); li++) {
		const int   L  = t.num_levels - (int) li;
		LevelNodes &ln = lns[li];
		Level      &lv = h.levels[li];
		lv.dim         = dim;
		lv.n           = n;
		lv.tree_level  = L;
		const int P    = (int) ln.ids.size();
		lv.P_global    = P;
		lv.g_id.assign(P, -1);
		lv.g_rank.assign(P, 0);
		lv.g_local.assign(P, -1);
		lv.g_starts.assign((size_t) P * dim, 0);
		lv.g_lengths.assign((size_t) P * dim, 0);
		lv.g_nbr_kind.assign((size_t) P * nsides, NBR_NONE);
		lv.g_nbr.assign((size_t) P * nsides * 4, -1);
		lv.g_nbr_orth.assign((size_t) P * nsides, -1);
		lv.g_parent.assign(P, -1);
		lv.g_orth_on_parent.assign(P, -1);
		for (int p = 0; p < P; p++) {
			const Node &nd = t.nodes.at(ln.ids[p]);
			lv.g_id[p]     = nd.id;
			for (int a = 0; a < dim; a++) {
				lv.g_starts[(size_t) p * dim + a]  = nd.starts[a];
				lv.g_lengths[(size_t) p * dim + a] = nd.lengths[a];
			}
			for (int s = 0; s < nsides; s++) {
				size_t f = (size_t) p * nsides + s;
				if (nd.nbr[s] == -1 && nd.parent != -1 && t.nodes.at(nd.parent).nbr[s] != -1) {
					// coarser neighbour: the parent's neighbour (a leaf); quadrant = position
					// of this node among the parent's orthants on side s (ascending order)
					const Node &par = t.nodes.at(nd.parent);
					int         q   = 0;
					for (int o = 0; o < (1 << dim); o++) {
						if (!orthOnSide(o, s)) continue;
						if (par.child[o] == nd.id) break;
						q++;
					}
					lv.g_nbr_kind[f] = NBR_COARSE;
					lv.g_nbr[f * 4]  = ln.index.at(par.nbr[s]);
					lv.g_nbr_orth[f] = q;
				} else if (nd.level < L && nd.nbr[s] != -1 && t.nodes.at(nd.nbr[s]).hasChildren()) {
					const Node &nb   = t.nodes.at(nd.nbr[s]);
					lv.g_nbr_kind[f] = NBR_FINE;
					int q            = 0;
					for (int o = 0; o < (1 << dim); o++) {
						if (!orthOnSide(o, s ^ 1)) continue;
						lv.g_nbr[f * 4 + q] = ln.index.at(nb.child[o]);
						q++;
					}
					(void) nq;
				} else if (nd.nbr[s] != -1) {
					lv.g_nbr_kind[f] = NBR_NORMAL;
					lv.g_nbr[f * 4]  = ln.index.at(nd.nbr[s]);
				}
			}
		}
	}
	// parent links (AvgRstr.h:88-107 / InterLevelComm.h:115-160 semantics)
	for (size_t li = 0; li + 1 < lns.size(); li++) {
		const int L  = t.num_levels - (int) li;
		Level    &lv = h.levels[li];
		for (int p = 0; p < lv.P_global; p++) {
			const Node &nd = t.nodes.at(lv.g_id[p]);
			if (nd.level < L) {
				lv.g_parent[p]         = lns[li + 1].index.at(nd.id);
				lv.g_orth_on_parent[p] = -1;
			} else {
				lv.g_parent[p] = lns[li + 1].index.at(nd.parent);
				const Node &par = t.nodes.at(nd.parent);
				int         o   = 0;
				while (par.child[o] != nd.id) o++;
				lv.g_orth_on_parent[p] = o;
			}
		}
	}
	// partition: finest level = equal contiguous Morton ranges; coarser = follow orthant-0 /
	// copy-through child.
	{
		Level &f = h.levels[0];
		for (int p = 0; p < f.P_global; p++) f.g_rank[p] = (int) (((int64_t) p * nranks) / f.P_global);
		for (size_t li = 0; li + 1 < h.levels.size(); li++) {
			Level &fine = h.levels[li], &coarse = h.levels[li + 1];
			std::vector<int> assigned(coarse.P_global, 0);
			for (int p = 0; p < fine.P_global; p++) {
				int o = fine.g_orth_on_parent[p];
				if (o <= 0) {
					coarse.g_rank[fine.g_parent[p]] = fine.g_rank[p];
					assigned[fine.g_parent[p]]      = 1;
				}
			}
			for (int p = 0; p < coarse.P_global; p++) {
				if (!assigned[p]) throw std::runtime_error("te::Hierarchy: coarse patch without child");
			}
		}
		// Agglomeration (the patches_per_proc idea of CycleFactory3d.cpp:104, without cutting the hierarchy short): a
		// level with fewer than `agg` patches per rank, and every level below it, lives on rank 0. Those levels cost
		// microseconds of compute but one latency-bound neighbour exchange per stencil operation when spread out; gathered,
		// the cycle pays one block transfer down and one up (the inter-level blocks that exist anyway) and no exchange at
		// all below. A global fact (patch counts, rank count, TE_AGGLOMERATE): the same on every rank.
		// The per-rank threshold alone grows with the number of ranks (at 64 ranks a 512-patch level of 32^3 patches would land
		// on one GPU): the first gathered level also has at most TE_AGGLOMERATE_MAX patches in total (default 64 -- the size
		// below which one GPU runs a level in the same ~20 us however many patches it has).
		// TE_REPLICATE (default on, 3D): a gathered level is not rank 0's but EVERY rank's -- each rank receives the restricted
		// blocks of all children of the first gathered level (the transfer rank 0 alone received before, now to everybody: the
		// same bytes per link) and runs the small levels itself, redundantly and bit for bit the same; the way back up then needs
		// no transfer at all, every parent being local. One exchange per cycle less on the critical path, and the level above
		// keeps its fused post-sweep (its parents are local). 0: rank 0 alone, as before.
		// (64 per rank since round 4: above a level that lives on every rank the post-sweep needs no exchange at all, so gathering
		// the 64-patch level at 2 and 4 ranks as well takes three exchanges out of a 512^3 cycle: 659 -> 622, 528 -> 475 us per
		// rank in loop-back, profiles/r04_mr8_budget.txt)
		h.agglomerate     = pl.agglomerate >= 0 ? pl.agglomerate : 64.0;
		h.agglomerate_max = pl.agglomerate_max >= 0 ? pl.agglomerate_max : 64;
		h.replicate       = (pl.replicate < 0 || pl.replicate != 0) ? 1 : 0; // (round 5: 2D hierarchies as well)
		if (nranks > 1) {
			const double agg = h.agglomerate;
			const int    cap = h.agglomerate_max;
			const bool   rep = h.replicate != 0;
			bool         gathered = false;
			for (size_t li = 1; li < h.levels.size(); li++) {
				Level &lv = h.levels[li];
				if (!gathered && lv.P_global < agg * nranks && lv.P_global <= cap) gathered = true;
				if (gathered) {
					lv.replicated = rep;
					std::fill(lv.g_rank.begin(), lv.g_rank.end(), rep ? rank : 0);
				}
			}
		}
		for (auto &lv : h.levels) {
			std::vector<int> count(nranks, 0);
			for (int p = 0; p < lv.P_global; p++) {
				lv.g_local[p] = count[lv.g_rank[p]]++;
				if (lv.g_rank[p] == rank) lv.l2g.push_back(p);
			}
			lv.P = (int) lv.l2g.size();
		}
	}
	return h;
}
} // namespace te
