// 3D kernel launches: stencil (apply, residual, Jacobi, residual + restriction), red-black sweeps, restriction, prolongation (see gmg_internal.hpp).
#include "gmg_ghosts3d.hpp"

namespace tei
{
// one launch of k_stencil3d<N, MODE, ZS> with or without fused sums (RED: march3d.hpp StencilRed)
template <int N, int MODE, int ZS>
void launchStencilZS(te_gmg *g, dim3 grid, const LevelDev &D_, const double *u, const double *f, double *out, double omega,
                     const RestrictDst &rd, int redmode, const RedSrc &rs)
{
	const dim3 blk(Tile3<N>::TPB);
	const LevelDev &D = stamped(g, D_, MODE == MODE_RESID_RESTRICT ? "stencil resid+restrict" : "stencil", grid.x);
	if constexpr (MODE == MODE_APPLY) {
		if (redmode == RED_OUT_A) {
			hipLaunchKernelGGL((k_stencil3d<N, MODE, ZS, RED_OUT_A>), grid, blk, 0, g->stream, D, u, f, out, omega, rd, rs);
			return;
		}
		if (redmode == RED_OUT_A_OUT) {
			hipLaunchKernelGGL((k_stencil3d<N, MODE, ZS, RED_OUT_A_OUT>), grid, blk, 0, g->stream, D, u, f, out, omega, rd, rs);
			return;
		}
	}
	if constexpr (MODE == MODE_RESID) {
		if (redmode == RED_OUT_OUT) {
			hipLaunchKernelGGL((k_stencil3d<N, MODE, ZS, RED_OUT_OUT>), grid, blk, 0, g->stream, D, u, f, out, omega, rd, rs);
			return;
		}
	}
	hipLaunchKernelGGL((k_stencil3d<N, MODE, ZS, RED_NONE>), grid, blk, 0, g->stream, D, u, f, out, omega, rd, rs);
}

// redmode != RED_NONE: the kernel leaves one pair of partial sums per work item in g->partial (red_a: the second operand
// of the dot product); *red_items = their number (the caller runs k_reduce_final2 over them)
template <int N, int MODE> int launchStencilN(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out,
                                              double omega, RestrictDst rd = RestrictDst(), const double *xf_in = nullptr,
                                              int redmode = RED_NONE, const double *red_a = nullptr, int *red_items = nullptr)
{
	// enough workgroups to fill 256 CUs a few times over: split patches into z-slabs when few
	int zs = 1;
	if (N >= 8) {
		while (zs < 4 && (g->cfg.has(O_ZS_FORCE) || (size_t) L.P * zs < 2048) && N / (zs * 2) >= 4) zs *= 2;
		if (zs == 4 && N == 32 && L.P <= 64 && !g->cfg.has(O_NO_ZS8)) zs = 8; // (see rbgsSlabs)
	}
	if (redmode != RED_NONE) {
		if ((size_t) 2 * L.P * zs > g->partial.n) return te::fail(TE_ESTATE, "launchStencil: partial-sum buffer too small");
		if (red_items) *red_items = L.P * zs;
	}
	auto launch = [&](LevelDev D) {
		if (D.count == 0) return;
		Timed t(g, zs > 1 ? KC_STENCIL_SLABS
		                  : (MODE == MODE_APPLY ? (redmode != RED_NONE ? KC_APPLY_DOT : KC_APPLY)
		                                        : (MODE == MODE_RESID ? KC_RESID : (MODE == MODE_JACOBI ? KC_JACOBI : KC_RESID_RESTRICT))),
		        (size_t) D.count * L.nc);
		auto         grid = [&](int z) { return dim3(8 * ((D.count * z + 7) / 8)); };
		const RedSrc rs{red_a, g->partial.p, D.first * zs}; // (interior patches are launched before the boundary patches)
		switch (zs) {
			case 1: launchStencilZS<N, MODE, 1>(g, grid(1), D, u, f, out, omega, rd, redmode, rs); break;
			case 2:
				if constexpr (N >= 8) launchStencilZS<N, MODE, 2>(g, grid(2), D, u, f, out, omega, rd, redmode, rs);
				break;
			case 8:
				if constexpr (N >= 32) launchStencilZS<N, MODE, 8>(g, grid(8), D, u, f, out, omega, rd, redmode, rs);
				break;
			default:
				if constexpr (N >= 16) launchStencilZS<N, MODE, 4>(g, grid(4), D, u, f, out, omega, rd, redmode, rs);
				break;
		}
	};
	int rc = withGhosts<N>(g, L, u, launch, xf_in);
	if (rc) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}

template <int MODE> int launchStencil(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, double omega,
                                      RestrictDst rd, const double *xf_in, int redmode, const double *red_a, int *red_items)
{
	if (red_items) *red_items = 0;
	if (L.P == 0) return TE_OK;
	if (L.dim == 2) {
		if constexpr (MODE == MODE_RESID_RESTRICT)
			return te::fail(TE_EUNSUPPORTED, "fused residual+restrict has no 2D kernel");
		else
			return launchStencil2d<MODE>(g, L, u, f, out, omega, redmode, red_a, red_items);
	}
	switch (L.n) {
		case 4: return launchStencilN<4, MODE>(g, L, u, f, out, omega, rd, xf_in, redmode, red_a, red_items);
		case 8: return launchStencilN<8, MODE>(g, L, u, f, out, omega, rd, xf_in, redmode, red_a, red_items);
		case 16: return launchStencilN<16, MODE>(g, L, u, f, out, omega, rd, xf_in, redmode, red_a, red_items);
		default: return launchStencilN<32, MODE>(g, L, u, f, out, omega, rd, xf_in, redmode, red_a, red_items);
	}
}

// z-slabs per patch for the RB-GS kernels: enough workgroups to occupy 256 CUs x 4 when the level has few patches
template <int N> inline int rbgsSlabs(const te_gmg *g, int count)
{
	int zs = 1;
	while (zs < 4 && !g->cfg.has(O_RBGS_NOSLAB) && (size_t) count * zs < 1024 && N / (zs * 2) >= 4) zs *= 2;
	// very few patches (the coarsest levels of a cycle): a kernel is one patch's march, a dependent chain of plane steps of
	// ~1-2 us each that nothing hides -- eight slabs of four planes (six steps) instead of four of eight (ten steps)
	if (zs == 4 && N == 32 && count <= 64 && !g->cfg.has(O_NO_ZS8)) zs = 8;
	return zs;
}

template <int N, bool ZERO, bool PROLONG>
void launchRbgsKernel(te_gmg *g, const LevelDev &D_, const double *u, const double *f, double *out, const ProlongSrc &ps)
{
	const int  zs = rbgsSlabs<N>(g, D_.count);
	const dim3 grid(8 * ((D_.count * zs + 7) / 8)), blk(Tile3<N>::TPB);
	const LevelDev &D = stamped(g, D_, ZERO ? "rbgs from zero" : (PROLONG ? "rbgs on u + P e" : "rbgs"), grid.x);
	if (zs == 8) {
		if constexpr (N >= 32) hipLaunchKernelGGL((k_rbgs3d<N, ZERO, PROLONG, 8>), grid, blk, 0, g->stream, D, u, f, out, ps);
	} else if (zs == 4) {
		if constexpr (N >= 16) hipLaunchKernelGGL((k_rbgs3d<N, ZERO, PROLONG, 4>), grid, blk, 0, g->stream, D, u, f, out, ps);
	} else if (zs == 2) {
		if constexpr (N >= 8) hipLaunchKernelGGL((k_rbgs3d<N, ZERO, PROLONG, 2>), grid, blk, 0, g->stream, D, u, f, out, ps);
	} else {
		hipLaunchKernelGGL((k_rbgs3d<N, ZERO, PROLONG, 1>), grid, blk, 0, g->stream, D, u, f, out, ps);
	}
}

template <int N> int launchRbgsN(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, bool zero_guess,
                                 const double *prolong_from, const double *xf_in, double *xf_out)
{
	if (prolong_from) { // u + P(coarse) is formed on the fly: only for levels without coarse/fine faces (checked by the caller)
		ProlongSrc ps;
		ps.parent = L.parent.p;
		ps.orth   = L.orth.p;
		ps.coarse = prolong_from;
		ps.cbase  = L.cbase.p; // (null unless every patch is an octant child of a local parent)
		const bool cfp = (L.ncf > 0 || L.has_copy); // refined level: copy-through patches / coarse-fine ghost slots
		if (!cfp && !ps.cbase) return te::fail(TE_ESTATE, "launchRbgs: fused prolongation on a level without its coarse-base table");
		auto launch = [&](LevelDev D) {
			if (D.count == 0) return;
			if (cfp) {
				Timed t(g, KC_RBGS_PROLONG, (size_t) D.count * L.nc);
				hipLaunchKernelGGL((k_rbgs3d<N, false, true, 1, true>), dim3(8 * ((D.count + 7) / 8)), dim3(Tile3<N>::TPB), 0, g->stream,
				                   D, u, f, out, ps);
				return;
			}
			Timed t(g, rbgsSlabs<N>(g, D.count) > 1 ? KC_RBGS_SLABS : KC_RBGS_PROLONG, (size_t) D.count * L.nc);
			launchRbgsKernel<N, false, true>(g, D, u, f, out, ps);
		};
		// neighbours on other ranks receive this rank's face layers of u + P(coarse) (exchange under the interior)
		int rc = withGhosts<N>(g, L, u, launch, xf_in, xf_out, &ps);
		if (rc) return rc;
		HIPCHK(hipGetLastError());
		return TE_OK;
	}
	if (zero_guess) { // a zero iterate has zero ghosts everywhere: nothing to exchange or build
		Timed    t(g, rbgsSlabs<N>(g, L.P) > 1 ? KC_RBGS_SLABS : KC_RBGS_ZERO, (size_t) L.P * L.nc);
		LevelDev D = L.dev();
		D.xf_out   = xf_out;
		launchRbgsKernel<N, true, false>(g, D, u, f, out, ProlongSrc());
		HIPCHK(hipGetLastError());
		return TE_OK;
	}
	auto launch = [&](LevelDev D) {
		if (D.count == 0) return;
		Timed t(g, rbgsSlabs<N>(g, D.count) > 1 ? KC_RBGS_SLABS : KC_RBGS, (size_t) D.count * L.nc);
		launchRbgsKernel<N, false, false>(g, D, u, f, out, ProlongSrc());
	};
	int rc = withGhosts<N>(g, L, u, launch, xf_in, xf_out);
	if (rc) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}

int launchRbgs(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, bool zero_guess, const double *prolong_from,
               const double *xf_in, double *xf_out)
{
	if (L.P == 0) return TE_OK;
	if (L.dim == 2) return launchRbgs2d(g, L, u, f, out, zero_guess, prolong_from);
	switch (L.n) {
		case 4: return launchRbgsN<4>(g, L, u, f, out, zero_guess, prolong_from, xf_in, xf_out);
		case 8: return launchRbgsN<8>(g, L, u, f, out, zero_guess, prolong_from, xf_in, xf_out);
		case 16: return launchRbgsN<16>(g, L, u, f, out, zero_guess, prolong_from, xf_in, xf_out);
		default: return launchRbgsN<32>(g, L, u, f, out, zero_guess, prolong_from, xf_in, xf_out);
	}
}

// Cycle.h:59-65 in one pass: coarse f = AvgRstr(f - A u), r never stored. Children whose parent is
// on another rank write their block into upbuf; received blocks are placed by k_restrict_unpack3d.
template <int N> int residRestrictN(te_gmg *g, LevelHost &L, const double *u, const double *f, double *coarse, const double *xf_in)
{
	RestrictDst rd = RestrictDst();
	rd.parent     = L.parent.p;
	rd.orth       = L.orth.p;
	rd.coarse     = coarse;
	rd.remote     = L.upbuf.p;
	rd.remote_off = L.up_off.p;
	int rc        = TE_OK;
	if (L.P > 0) rc = launchStencilN<N, MODE_RESID_RESTRICT>(g, L, u, f, L.r->d, 0.0, rd, xf_in);
	if (rc) return rc;
	if ((rc = shipRestricted<N>(g, L, coarse))) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}

int residRestrict(te_gmg *g, LevelHost &L, const double *u, const double *f, double *coarse, const double *xf_in)
{
	if (L.dim == 2) return residRestrict2d(g, L, u, f, coarse);
	switch (L.n) {
		case 4: return residRestrictN<4>(g, L, u, f, coarse, xf_in);
		case 8: return residRestrictN<8>(g, L, u, f, coarse, xf_in);
		case 16: return residRestrictN<16>(g, L, u, f, coarse, xf_in);
		default: return residRestrictN<32>(g, L, u, f, coarse, xf_in);
	}
}

template <int N> int restrictN(te_gmg *g, LevelHost &L, const double *fine, double *coarse)
{
	if (L.n_up > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_up * L.nc / 8);
		hipLaunchKernelGGL(k_restrict_pack3d<N>, dim3(L.n_up), dim3(256), 0, g->stream, L.up_desc.p, L.up_off.p, fine,
		                   L.upbuf.p);
	}
	int rc = doExchange(g, 2, L.tx_up, L.upbuf.p, L.downbuf.p);
	if (rc) return rc;
	if (L.Pc == 0) return TE_OK;
	Timed t(g, KC_RESTRICT, (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_restrict3d<N>, dim3(gridFor((size_t) L.Pc * L.nc, 256)), dim3(256), 0, g->stream, L.Pc,
	                   L.child.p, L.copy.p, fine, L.downbuf.p, L.down_off.p, coarse);
	HIPCHK(hipGetLastError());
	return TE_OK;
}

template <int N> int prolongN(te_gmg *g, LevelHost &L, const double *coarse, double *fine)
{
	if (L.n_down > 0 && !L.repl_up) { // (repl_up: those blocks were received for the restriction; every parent is local)
		Timed t(g, KC_PACK, (size_t) L.n_down * L.nc / 8);
		hipLaunchKernelGGL(k_prolong_pack3d<N>, dim3(L.n_down), dim3(256), 0, g->stream, L.down_desc.p, L.down_off.p,
		                   coarse, L.downbuf.p);
	}
	int rc = doExchange(g, 3, L.tx_down, L.downbuf.p, L.upbuf.p);
	if (rc) return rc;
	if (L.P == 0) return TE_OK;
	Timed t(g, KC_PROLONG, (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_prolong3d<N>, dim3(gridFor((size_t) L.P * L.nc / 2, 256)), dim3(256), 0, g->stream, L.P,
	                   L.parent.p, L.orth.p, coarse, L.upbuf.p, L.up_off.p, fine);
	HIPCHK(hipGetLastError());
	return TE_OK;
}

int doRestrict(te_gmg *g, int fine_level, const double *fine, double *coarse)
{
	LevelHost &L = *g->levels[fine_level];
	if (L.dim == 2) return restrict2d(g, L, fine, coarse);
	switch (L.n) {
		case 4: return restrictN<4>(g, L, fine, coarse);
		case 8: return restrictN<8>(g, L, fine, coarse);
		case 16: return restrictN<16>(g, L, fine, coarse);
		default: return restrictN<32>(g, L, fine, coarse);
	}
}

int doProlong(te_gmg *g, int fine_level, const double *coarse, double *fine)
{
	LevelHost &L = *g->levels[fine_level];
	if (L.dim == 2) return prolong2d(g, L, coarse, fine);
	switch (L.n) {
		case 4: return prolongN<4>(g, L, coarse, fine);
		case 8: return prolongN<8>(g, L, coarse, fine);
		case 16: return prolongN<16>(g, L, coarse, fine);
		default: return prolongN<32>(g, L, coarse, fine);
	}
}

template int launchStencil<MODE_APPLY>(te_gmg *, LevelHost &, const double *, const double *, double *, double, RestrictDst, const double *, int, const double *, int *);
template int launchStencil<MODE_RESID>(te_gmg *, LevelHost &, const double *, const double *, double *, double, RestrictDst, const double *, int, const double *, int *);
template int launchStencil<MODE_JACOBI>(te_gmg *, LevelHost &, const double *, const double *, double *, double, RestrictDst, const double *, int, const double *, int *);
} // namespace tei

