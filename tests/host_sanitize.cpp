// CPU: the host-only half of the C ABI (mesh reader, refineLeaves, level extraction, Morton partition, agglomeration) under
// AddressSanitizer + UndefinedBehaviorSanitizer, driven through include/te_hip.h exactly as a host would. Built and run by
// tests/test_capi_surface.py::test_host_mesh_code_is_clean_under_sanitizers (GPU sanitizers are not available on this
// pool; the device half has no host-side allocation patterns of its own beyond std::vector uploads).
#include "te_hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

static int fail(const char *what)
{
	fprintf(stderr, "FAILED: %s: %s\n", what, te_last_error());
	return 1;
}

int main(int argc, char **argv)
{
	long patches = 0;
	for (int a = 1; a + 2 < argc; a += 3) { // triples: mesh file, dim, divides
		const int dim = atoi(argv[a + 1]), div = atoi(argv[a + 2]);
		te_mesh  *m = nullptr;
		if (te_mesh_read(argv[a], dim, &m)) return fail("te_mesh_read");
		for (int i = 0; i < div; i++)
			if (te_mesh_refine_leaves(m)) return fail("te_mesh_refine_leaves");
		const int           nn = te_mesh_num_nodes(m);
		std::vector<int32_t> ilp((size_t) nn * 3), nbr((size_t) nn * 2 * dim), child((size_t) nn * (1 << dim));
		std::vector<double>  len((size_t) nn * dim), st((size_t) nn * dim);
		if (te_mesh_get_nodes(m, ilp.data(), len.data(), st.data(), nbr.data(), child.data())) return fail("te_mesh_get_nodes");
		for (int nranks : {1, 2, 3, 8}) {
			for (int rank = 0; rank < nranks; rank += (nranks > 2 ? nranks - 1 : 1)) {
				te_hier *h = nullptr;
				if (te_hier_build(m, 4, rank & 1, 0, 0.0, rank, nranks, &h)) return fail("te_hier_build");
				for (int l = 0; l < te_hier_num_levels(h); l++) {
					int pl = 0, pg = 0;
					if (te_hier_level_sizes(h, l, &pl, &pg)) return fail("te_hier_level_sizes");
					const int            ns = 2 * dim;
					std::vector<int32_t> id(pg), rk(pg), lo(pg), kind((size_t) pg * ns), nb((size_t) pg * ns * 4), orth((size_t) pg * ns), par(pg), oop(pg), l2g(pl);
					std::vector<double>  s((size_t) pg * dim), ln((size_t) pg * dim);
					if (te_hier_level_tables(h, l, id.data(), rk.data(), lo.data(), s.data(), ln.data(), kind.data(), nb.data(), orth.data(), par.data(),
					                         oop.data()))
						return fail("te_hier_level_tables");
					if (te_hier_level_l2g(h, l, l2g.data())) return fail("te_hier_level_l2g");
					patches += pg;
				}
				te_hier_destroy(h);
			}
		}
		// error paths: a bad patch size and a bad rank must come back as codes, not as exceptions or leaks
		te_hier *h = nullptr;
		if (te_hier_build(m, 7, 0, 0, 0.0, 0, 1, &h) == TE_OK) return fail("odd n accepted");
		if (te_hier_build(m, 4, 0, 0, 0.0, 5, 2, &h) == TE_OK) return fail("rank >= nranks accepted");
		te_mesh_destroy(m);
	}
	te_mesh *bad = nullptr;
	if (te_mesh_read("/nonexistent/mesh.bin", 3, &bad) == TE_OK) return fail("missing file accepted");
	printf("SANITIZE_OK %ld\n", patches);
	return 0;
}
