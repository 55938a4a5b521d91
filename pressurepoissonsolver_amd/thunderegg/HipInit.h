// Init::initDirichlet / Init::initNeumann (apps/shared/Init.cpp:152-245, :57-151) and their 2D twins Init::initDirichlet2d /
// Init::initNeumann2d (:304-361, :246-303) for vectors of ANY Vector<D> subclass, filled through getLocalData(i) one patch at
// a time -- for HipVector<D> that is one n^D patch up and down per patch (te_vec_download_patches / te_vec_upload_patches),
// never the whole vector.
//
// The reference's Init takes a Domain<D>& and raw PETSc Vec handles (f->vec, apps/3d/steady.cpp:319-321, apps/2d/steady.cpp:
// 383-386): geometry
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
#include <array>
#include <functional>
#include <memory>
#include <te_hip.h>
#include <vector>

namespace tehip
{
struct LevelGeometry { // this rank's patches of one level, in vector order (2D or 3D: te_hier_dim)
	int                  n = 0, P = 0, dim = 3;
	std::vector<double>  starts, lengths; // [P][dim]
	std::vector<int32_t> nbr_kind;        // [P][2 dim], 0 = physical boundary
	LevelGeometry(const te_hier *h, int level)
	{
		int Pl = 0, Pg = 0;
		if (te_hier_level_sizes(h, level, &Pl, &Pg) != TE_OK) throw 3;
		n   = te_hier_n(h);
		dim = te_hier_dim(h);
		P   = Pl;
		const int D = dim, NS = 2 * dim;
		std::vector<double>  gs((size_t) Pg * D), gl((size_t) Pg * D);
		std::vector<int32_t> gk((size_t) Pg * NS), l2g((size_t) (Pl > 0 ? Pl : 1));
		if (te_hier_level_tables(h, level, nullptr, nullptr, nullptr, gs.data(), gl.data(), gk.data(), nullptr, nullptr, nullptr, nullptr) != TE_OK) throw 3;
		if (te_hier_level_l2g(h, level, l2g.data()) != TE_OK) throw 3;
		starts.resize((size_t) Pl * D), lengths.resize((size_t) Pl * D), nbr_kind.resize((size_t) Pl * NS);
		for (int p = 0; p < Pl; p++) {
			for (int a = 0; a < D; a++) starts[p * D + a] = gs[(size_t) l2g[p] * D + a], lengths[p * D + a] = gl[(size_t) l2g[p] * D + a];
			for (int s = 0; s < NS; s++) nbr_kind[p * NS + s] = gk[(size_t) l2g[p] * NS + s];
		}
	}
};

namespace detail
{
using Fun3 = std::function<double(double, double, double)>;
using Fun2 = std::function<double(double, double)>;
// Init.cpp:25-50: index -1 / n = the patch face, else the cell centre
inline double coord(double start, double h, int n, int i) { return i == -1 ? start : (i == n ? start + h * n : start + h / 2.0 + h * i); }
// the common loop: F(x[D]) / E(x[D]) at the cell centres, then on every physical face (west, east, south, north[, bottom,
// top]: Init.cpp:186-240 / :329-358 order) f += face(side, x[D] on the face, h of the face's axis) for the cells along it
template <size_t D, class FF, class EF, class Face>
void init(const LevelGeometry &G, std::shared_ptr<Vector<D>> f, std::shared_ptr<Vector<D>> exact, FF ffun, EF efun, Face face)
{
	if (G.dim != (int) D) throw 3;
	const int n = G.n;
	for (int p = 0; p < G.P; p++) {
		LocalData<D>  fv = f->getLocalData(p), ev = exact->getLocalData(p);
		const double *st = &G.starts[p * D];
		double        h[D];
		for (size_t a = 0; a < D; a++) h[a] = G.lengths[p * D + a] / n;
		auto X = [&](int a, int i) { return coord(st[a], h[a], n, i); };
		std::array<int, D> c;
		int                total = 1;
		for (size_t a = 0; a < D; a++) total *= n;
		for (int i = 0; i < total; i++) { // x fastest, as the reference's nested loops
			int    r = i;
			double x[D];
			for (size_t a = 0; a < D; a++) {
				c[a] = r % n;
				r /= n;
				x[a] = X((int) a, c[a]);
			}
			fv[c] = ffun(x);
			ev[c] = efun(x);
		}
		for (int s = 0; s < 2 * (int) D; s++) {
			if (G.nbr_kind[p * 2 * D + s] != 0) continue;
			const int ax = s / 2, fixed = (s & 1) ? n - 1 : 0, out = (s & 1) ? n : -1;
			int       ftotal = 1;
			for (size_t a = 0; a + 1 < D; a++) ftotal *= n;
			for (int i = 0; i < ftotal; i++) {
				int    r = i;
				double x[D];
				for (size_t a = 0; a < D; a++) { // the other axes in order, the first fastest
					if ((int) a == ax) {
						c[a] = fixed;
						x[a] = X((int) a, out);
					} else {
						c[a] = r % n;
						r /= n;
						x[a] = X((int) a, c[a]);
					}
				}
				fv[c] += face(s, x, h[ax]);
			}
		}
	}
}
} // namespace detail

/// Init::initDirichlet(domain, f, exact, ffun, efun), Init.cpp:152-245
inline void initDirichlet(const LevelGeometry &G, std::shared_ptr<Vector<3>> f, std::shared_ptr<Vector<3>> exact, detail::Fun3 ffun, detail::Fun3 efun)
{
	detail::init<3>(G, f, exact, [&](const double *x) { return ffun(x[0], x[1], x[2]); }, [&](const double *x) { return efun(x[0], x[1], x[2]); },
	                [&](int, const double *x, double h) { return -(2 * efun(x[0], x[1], x[2]) / (h * h)); });
}
/// Init::initNeumann(domain, f, exact, ffun, efun, nfunx, nfuny, nfunz), Init.cpp:57-151: += n_x / h on the low side, -= on the high side
inline void initNeumann(const LevelGeometry &G, std::shared_ptr<Vector<3>> f, std::shared_ptr<Vector<3>> exact, detail::Fun3 ffun, detail::Fun3 efun,
                        detail::Fun3 nfunx, detail::Fun3 nfuny, detail::Fun3 nfunz)
{
	detail::init<3>(G, f, exact, [&](const double *x) { return ffun(x[0], x[1], x[2]); }, [&](const double *x) { return efun(x[0], x[1], x[2]); },
	                [&](int s, const double *x, double h) {
		                const double g = (s / 2 == 0) ? nfunx(x[0], x[1], x[2]) : (s / 2 == 1 ? nfuny(x[0], x[1], x[2]) : nfunz(x[0], x[1], x[2]));
		                return (s & 1) ? -(g / h) : g / h;
	                });
}
/// Init::initDirichlet2d(domain, f, exact, ffun, efun), Init.cpp:304-361 (-= efun(face) * 2 / h^2 on every physical face)
inline void initDirichlet2d(const LevelGeometry &G, std::shared_ptr<Vector<2>> f, std::shared_ptr<Vector<2>> exact, detail::Fun2 ffun, detail::Fun2 efun)
{
	detail::init<2>(G, f, exact, [&](const double *x) { return ffun(x[0], x[1]); }, [&](const double *x) { return efun(x[0], x[1]); },
	                [&](int, const double *x, double h) { return -(efun(x[0], x[1]) * 2 / (h * h)); });
}
/// Init::initNeumann2d(domain, f, exact, ffun, efun, nfunx, nfuny), Init.cpp:246-303
inline void initNeumann2d(const LevelGeometry &G, std::shared_ptr<Vector<2>> f, std::shared_ptr<Vector<2>> exact, detail::Fun2 ffun, detail::Fun2 efun,
                          detail::Fun2 nfunx, detail::Fun2 nfuny)
{
	detail::init<2>(G, f, exact, [&](const double *x) { return ffun(x[0], x[1]); }, [&](const double *x) { return efun(x[0], x[1]); },
	                [&](int s, const double *x, double h) {
		                const double g = (s / 2 == 0) ? nfunx(x[0], x[1]) : nfuny(x[0], x[1]);
		                return (s & 1) ? -(g / h) : g / h;
	                });
}
} // namespace tehip
#endif
