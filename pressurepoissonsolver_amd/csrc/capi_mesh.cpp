// C ABI for the host-only half of include/te_hip.h (te_mesh_*, te_hier_*). No HIP calls here.
#include "capi_common.hpp"
#include <cstring>

namespace te
{
std::string &lastError()
{
	static thread_local std::string s;
	return s;
}
int fail(int code, const std::string &msg)
{
	lastError() = msg;
	return code;
}
} // namespace te

template <typename T> static void copyOut(T *dst, const std::vector<T> &src)
{
	if (dst && !src.empty()) memcpy(dst, src.data(), sizeof(T) * src.size());
}

extern "C" {
const char *te_last_error(void) { return te::lastError().c_str(); }
const char *te_version(void) { return "pressurepoissonsolver_amd 0.1 (gfx950)"; }

int te_mesh_read(const char *path, int dim, te_mesh **out)
{
	if (!path || !out) return te::fail(TE_EINVAL, "te_mesh_read: null argument");
	try {
		*out = new te_mesh{te::Tree::read(path, dim)};
		return TE_OK;
	} catch (const std::exception &e) {
		return te::fail(TE_EIO, e.what());
	}
}
int te_mesh_unit_root(int dim, te_mesh **out)
{
	if (!out || (dim != 2 && dim != 3)) return te::fail(TE_EINVAL, "te_mesh_unit_root: bad argument");
	*out = new te_mesh{te::Tree::unitRoot(dim)};
	return TE_OK;
}
int te_mesh_refine_leaves(te_mesh *m)
{
	if (!m) return te::fail(TE_EINVAL, "te_mesh_refine_leaves: null mesh");
	try {
		m->tree.refineLeaves();
		return TE_OK;
	} catch (const std::exception &e) {
		return te::fail(TE_EINVAL, e.what());
	}
}
int te_mesh_num_nodes(const te_mesh *m) { return m ? (int) m->tree.nodes.size() : TE_EINVAL; }
int te_mesh_num_levels(const te_mesh *m) { return m ? m->tree.num_levels : TE_EINVAL; }
int te_mesh_dim(const te_mesh *m) { return m ? m->tree.dim : TE_EINVAL; }
int te_mesh_get_nodes(const te_mesh *m, int32_t *ilp, double *lengths, double *starts, int32_t *nbr,
                      int32_t *child)
{
	if (!m) return te::fail(TE_EINVAL, "te_mesh_get_nodes: null mesh");
	const int dim = m->tree.dim, ns = 2 * dim, no = 1 << dim;
	size_t    i = 0;
	for (auto &p : m->tree.nodes) {
		const te::Node &nd = p.second;
		if (ilp) {
			ilp[3 * i]     = nd.id;
			ilp[3 * i + 1] = nd.level;
			ilp[3 * i + 2] = nd.parent;
		}
		for (int a = 0; a < dim; a++) {
			if (lengths) lengths[i * dim + a] = nd.lengths[a];
			if (starts) starts[i * dim + a] = nd.starts[a];
		}
		if (nbr)
			for (int s = 0; s < ns; s++) nbr[i * ns + s] = nd.nbr[s];
		if (child)
			for (int o = 0; o < no; o++) child[i * no + o] = nd.child[o];
		i++;
	}
	return TE_OK;
}
void te_mesh_destroy(te_mesh *m) { delete m; }

int te_hier_build_placed(const te_mesh *m, int n, int neumann, int max_levels, double patches_per_proc,
                         int rank, int nranks, double agglomerate, int agglomerate_max, int replicate, te_hier **out)
{
	if (!m || !out) return te::fail(TE_EINVAL, "te_hier_build: null argument");
	if (nranks < 1 || rank < 0 || rank >= nranks) return te::fail(TE_EINVAL, "te_hier_build: rank outside [0, nranks)");
	try {
		te::Placement pl;
		pl.agglomerate     = agglomerate;
		pl.agglomerate_max = agglomerate_max;
		pl.replicate       = replicate;
		*out = new te_hier{te::Hierarchy::build(m->tree, n, neumann != 0, max_levels,
		                                        patches_per_proc, rank, nranks, pl)};
		return TE_OK;
	} catch (const std::exception &e) {
		return te::fail(TE_EINVAL, e.what());
	}
}
// the placement of the small levels from the environment (each variable read here, once per call; unset = the default)
int te_hier_build(const te_mesh *m, int n, int neumann, int max_levels, double patches_per_proc,
                  int rank, int nranks, te_hier **out)
{
	const char *e = getenv("TE_AGGLOMERATE"), *em = getenv("TE_AGGLOMERATE_MAX"), *er = getenv("TE_REPLICATE");
	return te_hier_build_placed(m, n, neumann, max_levels, patches_per_proc, rank, nranks, e ? atof(e) : -1.0, em ? atoi(em) : -1,
	                            er ? (atoi(er) != 0) : -1, out);
}
int te_hier_placement(const te_hier *h, double *agglomerate, int *agglomerate_max, int *replicate)
{
	if (!h) return te::fail(TE_EINVAL, "te_hier_placement: null");
	if (agglomerate) *agglomerate = h->h.agglomerate;
	if (agglomerate_max) *agglomerate_max = h->h.agglomerate_max;
	if (replicate) *replicate = h->h.replicate;
	return TE_OK;
}
int te_hier_num_levels(const te_hier *h) { return h ? (int) h->h.levels.size() : TE_EINVAL; }
int te_hier_dim(const te_hier *h) { return h ? h->h.dim : TE_EINVAL; }
int te_hier_n(const te_hier *h) { return h ? h->h.n : TE_EINVAL; }
int te_hier_level_sizes(const te_hier *h, int level, int *P_local, int *P_global)
{
	if (!h || level < 0 || level >= (int) h->h.levels.size())
		return te::fail(TE_EINVAL, "te_hier_level_sizes: bad level");
	if (P_local) *P_local = h->h.levels[level].P;
	if (P_global) *P_global = h->h.levels[level].P_global;
	return TE_OK;
}
int te_hier_level_replicated(const te_hier *h, int level)
{
	if (!h || level < 0 || level >= (int) h->h.levels.size())
		return te::fail(TE_EINVAL, "te_hier_level_replicated: bad level");
	return h->h.levels[level].replicated ? 1 : 0;
}
int te_hier_level_tables(const te_hier *h, int level, int32_t *id, int32_t *rank, int32_t *local,
                         double *starts, double *lengths, int32_t *nbr_kind, int32_t *nbr,
                         int32_t *nbr_orth, int32_t *parent, int32_t *orth_on_parent)
{
	if (!h || level < 0 || level >= (int) h->h.levels.size())
		return te::fail(TE_EINVAL, "te_hier_level_tables: bad level");
	const te::Level &lv = h->h.levels[level];
	copyOut(id, lv.g_id);
	copyOut(rank, lv.g_rank);
	copyOut(local, lv.g_local);
	copyOut(starts, lv.g_starts);
	copyOut(lengths, lv.g_lengths);
	copyOut(nbr_kind, lv.g_nbr_kind);
	copyOut(nbr, lv.g_nbr);
	copyOut(nbr_orth, lv.g_nbr_orth);
	copyOut(parent, lv.g_parent);
	copyOut(orth_on_parent, lv.g_orth_on_parent);
	return TE_OK;
}
int te_hier_level_l2g(const te_hier *h, int level, int32_t *l2g)
{
	if (!h || level < 0 || level >= (int) h->h.levels.size() || (!l2g && !h->h.levels[level].l2g.empty()))
		return te::fail(TE_EINVAL, "te_hier_level_l2g: bad argument"); // (a rank without patches on the level may pass NULL)
	copyOut(l2g, h->h.levels[level].l2g);
	return TE_OK;
}
void te_hier_destroy(te_hier *h) { delete h; }
} // extern "C"
