// Init::initDirichlet / Init::initNeumann (apps/shared/Init.cpp:152-245, :57-151) for vectors of ANY Vector<3>
// subclass, filled through getLocalData(i) one patch at a time -- for HipVector<3> that is one n^3 patch up and down
// per patch (te_vec_download_patches / te_vec_upload_patches), never the whole vector.
//
// The reference's Init takes a Domain<3>& and raw PETSc Vec handles (f->vec, apps/3d/steady.cpp:319-321): geometry
// from PatchInfo, data through VecGetArray. Here geometry comes from the te_hier level tables (the same starts /
// lengths / neighbour facts, in this library's patch order) and data goes through the reference's own Vector<3>
// interface. Same arguments otherwise: user callbacks for f, the exact solution and (Neumann) its derivatives;
// cell centres and face centres as Init.cpp:25-50 getXYZ; physical Dirichlet data folded into f as -2 g / h^2
// (Init.cpp:186-240), Neumann data as -/+ g_n / h (Init.cpp:89-146).
//
// For the canned problems of apps/3d/steady.cpp the device kernels behind te_init_problem fill a te_vec without any
// host traffic; this header is the general path (arbitrary std::function callbacks).
#ifndef THUNDEREGG_HIP_INIT_H
#define THUNDEREGG_HIP_INIT_H
#include <Thunderegg/Vector.h>
#include <functional>
#include <memory>
#include <te_hip.h>
#include <vector>

namespace tehip
{
struct LevelGeometry { // this rank's patches of one level, in vector order
	int                  n = 0, P = 0;
	std::vector<double>  starts, lengths; // [P][3]
	std::vector<int32_t> nbr_kind;        // [P][6], 0 = physical boundary
	LevelGeometry(const te_hier *h, int level)
	{
		int Pl = 0, Pg = 0;
		if (te_hier_level_sizes(h, level, &Pl, &Pg) != TE_OK) throw 3;
		n = te_hier_n(h);
		P = Pl;
		std::vector<double>  gs((size_t) Pg * 3), gl((size_t) Pg * 3);
		std::vector<int32_t> gk((size_t) Pg * 6), l2g((size_t) (Pl > 0 ? Pl : 1));
		if (te_hier_level_tables(h, level, nullptr, nullptr, nullptr, gs.data(), gl.data(), gk.data(), nullptr, nullptr, nullptr, nullptr) != TE_OK) throw 3;
		if (te_hier_level_l2g(h, level, l2g.data()) != TE_OK) throw 3;
		starts.resize((size_t) Pl * 3), lengths.resize((size_t) Pl * 3), nbr_kind.resize((size_t) Pl * 6);
		for (int p = 0; p < Pl; p++) {
			for (int a = 0; a < 3; a++) starts[p * 3 + a] = gs[(size_t) l2g[p] * 3 + a], lengths[p * 3 + a] = gl[(size_t) l2g[p] * 3 + a];
			for (int s = 0; s < 6; s++) nbr_kind[p * 6 + s] = gk[(size_t) l2g[p] * 6 + s];
		}
	}
};

namespace detail
{
using Fun3 = std::function<double(double, double, double)>;
// Init.cpp:25-50: index -1 / n = the patch face, else the cell centre
inline double coord(double start, double h, int n, int i) { return i == -1 ? start : (i == n ? start + h * n : start + h / 2.0 + h * i); }
template <class Face> void init(const LevelGeometry &G, std::shared_ptr<Vector<3>> f, std::shared_ptr<Vector<3>> exact, Fun3 ffun, Fun3 efun, Face face)
{
	const int n = G.n;
	for (int p = 0; p < G.P; p++) {
		LocalData<3> fv = f->getLocalData(p), ev = exact->getLocalData(p);
		const double *st = &G.starts[p * 3];
		double        h[3];
		for (int a = 0; a < 3; a++) h[a] = G.lengths[p * 3 + a] / n;
		auto X = [&](int a, int i) { return coord(st[a], h[a], n, i); };
		for (int zi = 0; zi < n; zi++)
			for (int yi = 0; yi < n; yi++)
				for (int xi = 0; xi < n; xi++) {
					const double x = X(0, xi), y = X(1, yi), z = X(2, zi);
					fv[{{xi, yi, zi}}] = ffun(x, y, z);
					ev[{{xi, yi, zi}}] = efun(x, y, z);
				}
		for (int s = 0; s < 6; s++) { // west, east, south, north, bottom, top (Init.cpp:186-240 order)
			if (G.nbr_kind[p * 6 + s] != 0) continue;
			const int ax = s / 2, fixed = (s & 1) ? n - 1 : 0, out = (s & 1) ? n : -1;
			for (int b = 0; b < n; b++)
				for (int a = 0; a < n; a++) {
					int c[3], o[3];
					c[ax] = fixed, o[ax] = out;
					const int a0 = (ax == 0) ? 1 : 0, a1 = (ax == 2) ? 1 : 2; // the two other axes in order
					c[a0] = o[a0] = a;
					c[a1] = o[a1] = b;
					fv[{{c[0], c[1], c[2]}}] += face(s, X(0, o[0]), X(1, o[1]), X(2, o[2]), h[ax]);
				}
		}
	}
}
} // namespace detail

/// Init::initDirichlet(domain, f, exact, ffun, efun), Init.cpp:152-245
inline void initDirichlet(const LevelGeometry &G, std::shared_ptr<Vector<3>> f, std::shared_ptr<Vector<3>> exact, detail::Fun3 ffun, detail::Fun3 efun)
{
	detail::init(G, f, exact, ffun, efun, [&](int, double x, double y, double z, double h) { return -(2 * efun(x, y, z) / (h * h)); });
}
/// Init::initNeumann(domain, f, exact, ffun, efun, nfunx, nfuny, nfunz), Init.cpp:57-151: += n_x / h on the low side, -= on the high side
inline void initNeumann(const LevelGeometry &G, std::shared_ptr<Vector<3>> f, std::shared_ptr<Vector<3>> exact, detail::Fun3 ffun, detail::Fun3 efun,
                        detail::Fun3 nfunx, detail::Fun3 nfuny, detail::Fun3 nfunz)
{
	detail::init(G, f, exact, ffun, efun, [&](int s, double x, double y, double z, double h) {
		const double g = (s / 2 == 0) ? nfunx(x, y, z) : (s / 2 == 1 ? nfuny(x, y, z) : nfunz(x, y, z));
		return (s & 1) ? -(g / h) : g / h;
	});
}
} // namespace tehip
#endif
