// Host-side mesh model for the MI355X GMG path: octree/quadtree, per-level patch
// extraction, static Morton partition, and the flattened SoA tables the HIP kernels read.
//
// Replaces (behaviour, not code) the reference's
//   Tree<D>            src/Thunderegg/OctTree.h:34-213  (file reader, refineLeaves)
//   Node<D>            src/Thunderegg/OctNode.h:30-130
//   extractLevel       src/Thunderegg/ThundereggDomGen.h:127-222 (which nodes form a level,
//                      Normal / Coarse / Fine neighbour classification, parent + orthant)
//   level truncation   src/Thunderegg/GMG/CycleFactory3d.cpp:101-104 (max_levels,
//                      patches_per_proc)
//   PatchInfo<D>       src/Thunderegg/PatchInfo.h:74-277 (only the fields kernels consume)
// The Zoltan hypergraph partitioner (ThundereggDomGen.h:223-648) is replaced by a
// deterministic Morton-range partition: one contiguous Morton range of finest-level patches
// per rank; a coarser patch lives where its orthant-0 child (or itself, if it does not
// coarsen) lives.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace te
{
constexpr int MAX_D    = 3;
constexpr int MAX_SIDE = 6;
constexpr int MAX_ORTH = 8;

/// Side numbering follows src/Thunderegg/Side.h:51-56: W,E,S,N,B,T = 0..5; axis = side/2.
/// Orthant numbering follows Side.h:183-211: bit0 = x upper, bit1 = y upper, bit2 = z upper.
struct Node {
	int    id     = -1;
	int    level  = -1;
	int    parent = -1;
	double lengths[MAX_D];
	double starts[MAX_D];
	int    nbr[MAX_SIDE];
	int    child[MAX_ORTH];
	Node();
	bool hasChildren() const { return child[0] != -1; }
};

struct Tree {
	int                 dim = 3;
	std::map<int, Node> nodes;
	int                 root       = 0;
	int                 num_levels = 0;
	int                 max_id     = 0;
	/// depth of the root as stored in the file (shipped files use level = 1)
	int root_level = 1;

	static Tree read(const std::string &path, int dim);
	/// single root node on the unit box, level 1 (what a 1-node mesh file holds)
	static Tree unitRoot(int dim);
	/// split every leaf once; new ids are ++max_id in (depth, id) leaf order then orthant
	/// order, as OctTree.h:119-179 + :183-189 assign them.
	void refineLeaves();
	void refineNode(int id);
	int  depthOf(int id) const;
};

enum NbrKind : int32_t { NBR_NONE = 0, NBR_NORMAL = 1, NBR_COARSE = 2, NBR_FINE = 3 };

/// One multigrid level as seen by ONE rank. "Global" arrays cover every patch of the level
/// (small: tens of bytes per patch); local arrays cover this rank's patches only.
struct Level {
	int dim = 3;
	int n   = 0; ///< cells per axis per patch (cubic patches, as the reference's solvers assume)
	int tree_level = 0;

	// ---- global (all ranks' patches), Morton order ------------------------------------
	int                  P_global = 0;
	std::vector<int32_t> g_id;    ///< tree node id
	std::vector<int32_t> g_rank;  ///< owner rank
	std::vector<int32_t> g_local; ///< index inside the owner's local numbering
	std::vector<double>  g_starts;  ///< [P][dim]
	std::vector<double>  g_lengths; ///< [P][dim]
	std::vector<int32_t> g_nbr_kind; ///< [P][2*dim]
	std::vector<int32_t> g_nbr;      ///< [P][2*dim][4] global patch indices, -1 unused
	std::vector<int32_t> g_nbr_orth; ///< [P][2*dim] quadrant on the coarse nbr face (kind COARSE)
	std::vector<int32_t> g_parent;   ///< [P] global patch index in the next coarser level (-1 none)
	std::vector<int32_t> g_orth_on_parent; ///< [P] 0..2^dim-1, or -1 = copies through

	// ---- local --------------------------------------------------------------------------
	int                  P = 0;   ///< patches owned by this rank
	/// every rank holds (and computes) the whole level: g_rank is this rank's number for every patch, P == P_global
	bool                 replicated = false;
	std::vector<int32_t> l2g;     ///< [P] local -> global index
};

/// Where the levels with few patches live when nranks > 1 (the patches_per_proc idea of CycleFactory3d.cpp:104 without
/// cutting the hierarchy): a level with fewer than `agglomerate` patches per rank -- and at most `agglomerate_max` patches
/// in total -- and every level below it is gathered: on every rank (`replicate`, 3D) or on rank 0. Negative = the default
/// (64 / 64 / 1), which the environment may override (TE_AGGLOMERATE, TE_AGGLOMERATE_MAX, TE_REPLICATE: read by
/// te_hier_build only, once per call, and recorded in the hierarchy).
struct Placement {
	double agglomerate     = -1.0;
	int    agglomerate_max = -1, replicate = -1;
};

struct Hierarchy {
	int                dim = 3;
	int                n   = 0;
	int                rank = 0, nranks = 1;
	bool               neumann = false;
	std::vector<Level> levels; ///< [0] = finest
	/// placement of the small levels over the ranks (see Placement), as this hierarchy was built: part of what every rank
	/// must agree on (te_gmg checks it across the ranks before the first cycle)
	double agglomerate = 64.0;
	int    agglomerate_max = 64, replicate = 1;

	/// Build every level the reference's CycleFactory would build
	/// (CycleFactory3d.cpp:69-134): finest first, then coarser tree levels while
	/// (max_levels <= 0 || built < max_levels) and patches/nranks >= patches_per_proc.
	static Hierarchy build(const Tree &t, int n, bool neumann, int max_levels,
	                       double patches_per_proc, int rank, int nranks, const Placement &pl = Placement());
};

uint64_t mortonKey(const double *starts, const double *root_starts, const double *root_lengths,
                   int dim, int bits);
} // namespace te
