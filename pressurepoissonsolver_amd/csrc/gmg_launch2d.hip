// The 2D twins of the kernel launches (see gmg_internal.hpp).
#include "gmg_internal.hpp"

namespace tei
{
// ------------------------------------------------------------------------------ 2D launches
int prepareGhosts2d(te_gmg *g, LevelHost &L, const double *u)
{
	if (L.patch_local) return TE_OK;
	if (L.nremote > 0) {
		{
			Timed t(g, KC_PACK, (size_t) L.nremote * L.nf);
			hipLaunchKernelGGL(k_pack_faces2d, dim3(L.nremote), dim3(64), 0, g->stream, L.n, L.send_faces.p, u, L.sendbuf.p);
		}
		int rc = faceExchange(g, L, L.sendbuf.p);
		if (rc) return rc;
	}
	if (L.ncf == 0) return TE_OK;
	Timed t(g, KC_CFGHOST, (size_t) L.ncf * L.nf);
	hipLaunchKernelGGL(k_cf_ghost2d, dim3(L.ncf), dim3(64), 0, g->stream, L.n, L.cf_desc.p, L.cf_slots.p, u, L.ghostCur());
	return TE_OK;
}

// redmode != RED_NONE (APPLY, RESID): the kernel leaves one pair of partial sums per workgroup in g->partial (red_a: the second
// operand of the dot product); *red_items = their number (the caller runs k_reduce_final2 over them) -- the 2D twin of
// k_stencil3d's RED: te_bicgstab's dot products and the residual norm without passes of their own
template <int MODE> int launchStencil2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, double omega, int redmode,
                                        const double *red_a, int *red_items)
{
	if (red_items) *red_items = 0;
	int rc = prepareGhosts2d(g, L, u);
	if (rc) return rc;
	if (redmode != RED_NONE && L.P > 0) {
		if constexpr (MODE == MODE_JACOBI) {
			return te::fail(TE_EUNSUPPORTED, "fused sums: operator application and residual only");
		} else {
			// a FIXED grid per level size (the partial sums' order must not depend on anything else): flat, one x-pair per thread
			// (g->partial holds a pair per workgroup of the largest 2D level, te_gmg_create)
			const int blocks = gridFor((size_t) L.P * L.nc / 2, 256, (int) std::min<size_t>((size_t) 1 << 30, g->partial.n / 2));
			Timed     t(g, MODE == MODE_APPLY ? KC_APPLY_DOT : KC_RESID, (size_t) L.P * L.nc);
			if (redmode == RED_OUT_A)
				hipLaunchKernelGGL((k_stencil2d<MODE, RED_OUT_A>), dim3(blocks), dim3(256), 0, g->stream, L.dev2(), u, f, out, omega, g->partial.p, red_a);
			else if (redmode == RED_OUT_A_OUT)
				hipLaunchKernelGGL((k_stencil2d<MODE, RED_OUT_A_OUT>), dim3(blocks), dim3(256), 0, g->stream, L.dev2(), u, f, out, omega, g->partial.p, red_a);
			else
				hipLaunchKernelGGL((k_stencil2d<MODE, RED_OUT_OUT>), dim3(blocks), dim3(256), 0, g->stream, L.dev2(), u, f, out, omega, g->partial.p, red_a);
			if (red_items) *red_items = blocks;
			HIPCHK(hipGetLastError());
			return TE_OK;
		}
	}
	Timed t(g, MODE == MODE_APPLY ? KC_APPLY : (MODE == MODE_RESID ? KC_RESID : KC_JACOBI), (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_stencil2d<MODE>, dim3(gridFor((size_t) L.P * L.nc / 2, 256, 65536)), dim3(256), 0, g->stream, L.dev2(),
	                   u, f, out, omega);
	HIPCHK(hipGetLastError());
	return TE_OK;
}

// r = f - A u together with the partial sums of its squares, one PAIR per workgroup, in g->partial (*blocks of them): the residual
// norm without a second pass over r
int residualSumsq2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, int *blocks)
{
	return launchStencil2d<MODE_RESID>(g, L, u, f, out, 0.0, RED_OUT_OUT, nullptr, blocks);
}

// 64^2 patches: 512 threads per workgroup (four x-pairs per thread instead of eight: half the registers, twice the waves per
// CU at the same four resident patches)
static int tpb2d(const te_gmg *g) { return g->cfg.num(O_2D_TPB, 512) == 256 ? 256 : 512; }

// faces of u + P(coarse) for the neighbours on other ranks (u: the stored iterate, or e4: only its edge layers exist), and
// their values into this rank's ghost slots
int packProlongFaces2d(te_gmg *g, LevelHost &L, const double *u, const double *e4, const Prolong2D &ps)
{
	if (L.nremote == 0 || L.patch_local) return TE_OK;
	{
		Timed t(g, KC_PACK, (size_t) L.nremote * L.nf);
		hipLaunchKernelGGL(k_pack_faces_prolong2d, dim3(L.nremote), dim3(64), 0, g->stream, L.n, L.send_faces.p, u, e4, ps, L.sendbuf.p);
	}
	return faceExchange(g, L, L.sendbuf.p);
}

// zero_guess: levels with L.lds2d; prolong_from: levels with L.fuse2d && L.prolong_fusable (the caller checks)
int launchRbgs2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, bool zero_guess, const double *prolong_from)
{
	int rc;
	if (!zero_guess && !prolong_from && (rc = prepareGhosts2d(g, L, u))) return rc;
	if (prolong_from && (rc = packProlongFaces2d(g, L, u, nullptr, Prolong2D{L.parent.p, L.orth.p, prolong_from}))) return rc;
	if (L.P == 0) return TE_OK;
	if (L.n <= 64 && !g->cfg.has(O_2D_SIMPLE)) { // the patch and its halo ring fit in LDS: one pass
		const size_t lds = sizeof(double) * ((size_t) (L.n + 2) * (L.n + 2) + 16);
		Prolong2D    ps{L.parent.p, L.orth.p, prolong_from};
		Timed        t(g, zero_guess ? KC_RBGS_ZERO : (prolong_from ? KC_RBGS_PROLONG : KC_RBGS), (size_t) L.P * L.nc, true);
#define TE_RB2(Z, PR)                                                                                                          \
	if (L.n == 64 && tpb2d(g) == 512)                                                                                           \
		launchT(t, (k_rbgs2d_lds<Z, PR, 64, 512>), dim3(L.P), dim3(512), lds, g->stream, L.dev2(), u, f, out, ps);     \
	else if (L.n == 64)                                                                                                        \
		launchT(t, (k_rbgs2d_lds<Z, PR, 64>), dim3(L.P), dim3(256), lds, g->stream, L.dev2(), u, f, out, ps);          \
	else                                                                                                                       \
		launchT(t, (k_rbgs2d_lds<Z, PR, 0>), dim3(L.P), dim3(256), lds, g->stream, L.dev2(), u, f, out, ps)
		if (zero_guess)
			TE_RB2(true, false);
		else if (prolong_from)
			TE_RB2(false, true);
		else
			TE_RB2(false, false);
#undef TE_RB2
		HIPCHK(hipGetLastError());
		return TE_OK;
	}
	if (zero_guess || prolong_from) return te::fail(TE_ESTATE, "launchRbgs2d: fused variants need patches that fit in LDS");
	Timed      t(g, KC_RBGS, (size_t) L.P * L.nc);
	const dim3 grid(gridFor((size_t) L.P * L.nc, 256, 65536));
	hipLaunchKernelGGL(k_rbgs2d<0>, grid, dim3(256), 0, g->stream, L.dev2(), u, f, out);
	hipLaunchKernelGGL(k_rbgs2d<1>, grid, dim3(256), 0, g->stream, L.dev2(), u, f, out);
	HIPCHK(hipGetLastError());
	return TE_OK;
}

// The restricted blocks the kernels before this have finished -- in upbuf (children whose parent lives on another rank) or, when
// the coarse level lives on EVERY rank (repl_up, TE_REPLICATE: every parent is local), in the local coarse patches, from where the
// quadrants are copied out once and sent to everybody -- and the other ranks' blocks into the local coarse patches. The 2D twin of
// gmg_ghosts3d.hpp shipRestricted (block form). Replaces GMG/InterLevelComm.h:169-189 (scatter / scatterReverse).
static int shipRestricted2d(te_gmg *g, LevelHost &L, double *coarse)
{
	if (L.repl_up && L.n_up > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_up * L.nc / 4);
		hipLaunchKernelGGL(k_prolong_pack2d, dim3(L.n_up), dim3(256), 0, g->stream, L.n, L.bc_desc.p, L.up_off.p, coarse, L.upbuf.p);
	}
	int rc = doExchange(g, 2, L.tx_up, L.upbuf.p, L.downbuf.p);
	if (rc) return rc;
	if (L.n_down > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_down * L.nc / 4);
		hipLaunchKernelGGL(k_restrict_unpack2d, dim3(L.n_down), dim3(256), 0, g->stream, L.n, L.down_desc.p, L.down_off.p, L.downbuf.p, coarse);
	}
	HIPCHK(hipGetLastError());
	return TE_OK;
}

// coarse f = AvgRstr(f - A u) in one pass (levels with L.fuse2d: every parent local -- no transfers at all, or the coarse level
// lives on every rank and the finished blocks travel afterwards)
int residRestrict2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *coarse)
{
	if (L.P > 0) {
		int rc = prepareGhosts2d(g, L, u);
		if (rc) return rc;
		const size_t lds = sizeof(double) * ((size_t) (L.n + 2) * (L.n + 2) + 16);
		Timed        t(g, KC_RESID_RESTRICT, (size_t) L.P * L.nc);
		hipLaunchKernelGGL(k_resid_restrict2d_lds, dim3(L.P), dim3(256), lds, g->stream, L.dev2(), u, f,
		                   Prolong2D{L.parent.p, L.orth.p, nullptr}, coarse);
		HIPCHK(hipGetLastError());
	}
	return L.repl_up ? shipRestricted2d(g, L, coarse) : TE_OK; // (a rank without patches here still receives the others' blocks)
}

// *swapped: the result went to s1 (= L.t) instead of u: the caller exchanges the two vectors' buffers
// The fused forms of the block-Jacobi cycle in 2D: 64^2 patches on the matrix cores, uniformly refined levels.
// ps2dResidFusable: the residual after the pre-sweep is taken on the patch edges only. NOT bit-identical to the residual pass, so
// every rank -- and a single-rank run -- must take the same decision: global facts only (fuse2_ok, the patch size, options).
// patchSolve2dFusable: the post-sweep adds the prolongation itself. Bit-identical to the prolongation pass, so a rank decides by
// itself (every parent and child here: it has no part in the level's inter-level exchanges, whatever its peers do).
bool ps2dResidFusable(const te_gmg *g, const LevelHost &L)
{
	return L.dim == 2 && L.n == 64 && L.fuse2_ok && !g->cfg.has(O_2D_SIMPLE) && !g->cfg.has(O_2D_NO_MFMA) && !g->cfg.has(O_NO_FUSE2);
}
bool patchSolve2dFusable(const te_gmg *g, const LevelHost &L)
{
	// (repl_up: the coarse level lives on every rank -- the blocks that travel serve the restriction only)
	return ps2dResidFusable(g, L) && (L.matsT.p || L.P == 0) && L.fuse2d && L.prolong_fusable && (L.repl_up || (L.tx_up.empty() && L.n_down == 0));
}

// Cycle.h:57-65 after one block-Jacobi sweep from the zero iterate: inside a patch the residual of an exact patch solve vanishes
// (up to the solve's rounding: defined as 0), on the edge cells of a face with a neighbour it is -(g + m)/h^2 (the patch operator
// closed that face with ghost = -m where the operator has the neighbour's g): the coarse right-hand side is zero except along
// the quadrant edges, which k_restrict_fixup2d<own> writes (zeros and edge terms) from the edge values alone -- no pass over u and f
// (17 B per site), no memset.
int interfaceResidRestrict2d(te_gmg *g, LevelHost &L, const double *u, double *coarse, size_t coarse_n)
{
	int rc = prepareGhosts2d(g, L, u); // the new edge values of neighbours on other ranks
	if (rc) return rc;
	(void) coarse_n; // (every coarse block -- local, or on its way to the parent's rank -- is written whole by the kernel)
	if (L.P > 0) {
		Timed t(g, KC_FIXUP, (size_t) L.P * 4 * L.nf);
		hipLaunchKernelGGL(k_restrict_fixup2d, dim3(L.P), dim3(128), 0, g->stream, L.dev2(), u, (const double *) nullptr,
		                   Prolong2D{L.parent.p, L.orth.p, nullptr}, coarse, L.upbuf.p, L.up_off.p, true);
	}
	// children whose parent lives on another rank, or a coarse level on every rank: ship the finished blocks (as zeroSweepResid2d)
	HIPCHK(hipGetLastError());
	return shipRestricted2d(g, L, coarse);
}

int patchSolve2d(te_gmg *g, LevelHost &L, const double *f, double *u, double *s0, double *s1, bool zero_guess, bool *swapped,
                 const double *prolong_from)
{
	*swapped = false;
	int          rc;
	const size_t total = (size_t) L.P * L.nc;
	if (prolong_from && (zero_guess || !patchSolve2dFusable(g, L))) return te::fail(TE_ESTATE, "patchSolve2d: no fused prolongation on this level");
	if (L.n == 64 && L.matsT.p && !g->cfg.has(O_2D_SIMPLE) && !g->cfg.has(O_2D_NO_MFMA)) { // 64^2 patches: the four products on the matrix cores
		const Prolong2D ps{L.parent.p, L.orth.p, prolong_from};
		if (prolong_from) { // neighbours on other ranks send their facing values of u + P e
			if ((rc = packProlongFaces2d(g, L, u, nullptr, ps))) return rc;
		} else if (!zero_guess && (rc = prepareGhosts2d(g, L, u))) {
			return rc;
		}
		if (L.P == 0) return TE_OK;
		const size_t lds = sizeof(double) * 64 * PS2D_LD;
		bool        &attr = g->ps2d_attr;
		const bool   pf = L.P <= 256 && !g->cfg.has(O_2D_NO_PF); // few patches: a workgroup has its CU to itself anyway
		Timed        t(g, KC_PS_MFMA, total, true);
		// patches whose two axes close the same way on both sides take the half-size transforms (k_patch_solve2d_sym), the others
		// (Neumann boundary patches) the full ones: two launches on a level that has both
		const int    nsym = (L.mat2sym.p && !g->cfg.has(O_2D_NO_SYM)) ? L.n_pure2 : 0;
		const int32_t *lst = (nsym > 0 && nsym < L.P) ? L.ps2_list.p : nullptr;
		auto         launch = [&](auto kern) -> int {
            if (nsym < L.P)
                launchT(t, kern, dim3(L.P - nsym), dim3(256), lds, g->stream, L.dev2(), L.plan.p, L.matsT.p, L.lam.p, L.zero_mode.p, f, u, s1, ps,
                        lst ? lst + nsym : nullptr);
            return TE_OK;
		};
		auto launchSym = [&](auto kern) -> int {
			if (nsym > 0)
				launchT(t, kern, dim3(nsym), dim3(256), lds, g->stream, L.dev2(), L.plan.p, L.mat2sym.p, L.psinv.p, L.psitab.p, f, u, s1, lst, ps);
			return TE_OK;
		};
		if (!attr) { // all six once, so that the attribute is set whichever runs first
			const void *ks[12] = {reinterpret_cast<const void *>(k_patch_solve2d_mfma<true, true>), reinterpret_cast<const void *>(k_patch_solve2d_mfma<true, false>),
			                      reinterpret_cast<const void *>(k_patch_solve2d_mfma<false, true>), reinterpret_cast<const void *>(k_patch_solve2d_mfma<false, false>),
			                      reinterpret_cast<const void *>(k_patch_solve2d_mfma<false, true, true>), reinterpret_cast<const void *>(k_patch_solve2d_mfma<false, false, true>),
			                      reinterpret_cast<const void *>(k_patch_solve2d_sym<true, true>), reinterpret_cast<const void *>(k_patch_solve2d_sym<true, false>),
			                      reinterpret_cast<const void *>(k_patch_solve2d_sym<false, true>), reinterpret_cast<const void *>(k_patch_solve2d_sym<false, false>),
			                      reinterpret_cast<const void *>(k_patch_solve2d_sym<false, true, true>), reinterpret_cast<const void *>(k_patch_solve2d_sym<false, false, true>)};
			for (const void *k : ks) HIPCHK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
			attr = true;
		}
		if (zero_guess) {
			rc = pf ? launchSym(k_patch_solve2d_sym<true, true>) : launchSym(k_patch_solve2d_sym<true, false>);
			rc = pf ? launch(k_patch_solve2d_mfma<true, true>) : launch(k_patch_solve2d_mfma<true, false>);
		} else if (prolong_from) {
			rc = pf ? launchSym(k_patch_solve2d_sym<false, true, true>) : launchSym(k_patch_solve2d_sym<false, false, true>);
			rc = pf ? launch(k_patch_solve2d_mfma<false, true, true>) : launch(k_patch_solve2d_mfma<false, false, true>);
		} else {
			rc = pf ? launchSym(k_patch_solve2d_sym<false, true>) : launchSym(k_patch_solve2d_sym<false, false>);
			rc = pf ? launch(k_patch_solve2d_mfma<false, true>) : launch(k_patch_solve2d_mfma<false, false>);
		}
		if (rc) return rc;
		HIPCHK(hipGetLastError());
		*swapped = true;
		return TE_OK;
	}
	if (L.n <= 64 && L.matsT.p && !g->cfg.has(O_2D_SIMPLE)) { // one launch, the patch in LDS
		if (!zero_guess && (rc = prepareGhosts2d(g, L, u))) return rc;
		const size_t lds = sizeof(double) * 2 * L.nc;
		Timed        t(g, KC_PS_MFMA, total);
#define TE_PS2(Z, NC, T)                                                                                                     \
	hipLaunchKernelGGL((k_patch_solve2d_lds<Z, NC, T>), dim3(L.P), dim3(T), lds, g->stream, L.dev2(), L.plan.p, L.mats.p, \
	                   L.matsT.p, L.lam.p, L.zero_mode.p, f, u, s1)
		const bool wide = L.n == 64 && L.P <= 128; // few patches: sixteen waves per patch
		if (zero_guess) {
			if (wide)
				TE_PS2(true, 64, 1024);
			else if (L.n == 64)
				TE_PS2(true, 64, 256);
			else
				TE_PS2(true, 0, 256);
		} else {
			if (wide)
				TE_PS2(false, 64, 1024);
			else if (L.n == 64)
				TE_PS2(false, 64, 256);
			else
				TE_PS2(false, 0, 256);
		}
#undef TE_PS2
		HIPCHK(hipGetLastError());
		*swapped = true;
		return TE_OK;
	}
	if (zero_guess) {
		Timed t(g, KC_VECOP, total);
		HIPCHK(hipMemsetAsync(u, 0, sizeof(double) * total, g->stream));
	}
	if ((rc = prepareGhosts2d(g, L, u))) return rc;
	const dim3   grid(gridFor(total, 256, 65536)), blk(256);
	{
		Timed t(g, KC_PATCH_RHS, total);
		hipLaunchKernelGGL(k_patch_rhs2d, grid, blk, 0, g->stream, L.dev2(), u, f, s0);
	}
	const bool mfma2d = L.n % 16 == 0 && !g->cfg.has(O_2D_SIMPLE); // large patches: the passes on the matrix cores
	const dim3 gridm(((size_t) L.P * (L.n / 16) * (L.n / 16) + 3) / 4);
#define TE_DST2(STAGE, IN, OUT)                                                                                          \
	{                                                                                                                    \
		Timed t(g, KC_DST, total);                                                                                       \
		if (mfma2d)                                                                                                      \
			hipLaunchKernelGGL(k_dst_axis2d_mfma<STAGE>, gridm, blk, 0, g->stream, L.n, L.P, L.plan.p, L.mats.p, L.lam.p, \
			                   L.zero_mode.p, L.rh2.p, IN, OUT);                                                         \
		else                                                                                                             \
			hipLaunchKernelGGL(k_dst_axis2d<STAGE>, grid, blk, 0, g->stream, L.n, L.P, L.plan.p, L.mats.p, L.lam.p,      \
			                   L.zero_mode.p, L.rh2.p, IN, OUT);                                                         \
	}
	TE_DST2(0, s0, s1)
	TE_DST2(1, s1, s0)
	TE_DST2(2, s0, s1)
	TE_DST2(3, s1, u)
#undef TE_DST2
	HIPCHK(hipGetLastError());
	return TE_OK;
}

// PatchSolvers/BiCGStabSolver.h:114-132 for every patch of the level: s0 = f - interface terms of the old iterate, then each
// patch's own BiCGStab from its current values, in place (see k_patch_bcgs2d)
int patchBcgs2d(te_gmg *g, LevelHost &L, const double *f, double *u, double *s0)
{
	if (L.dim != 2) return te::fail(TE_EUNSUPPORTED, "TE_SMOOTH_PATCH_BCGS: the reference builds BiCGStabSolver in its 2D driver only (apps/2d/steady.cpp:326-327)");
	if (L.n > 64) return te::fail(TE_EUNSUPPORTED, "TE_SMOOTH_PATCH_BCGS: patches up to 64^2 (one workgroup keeps the Krylov vectors in registers)");
	int rc;
	if ((rc = prepareGhosts2d(g, L, u))) return rc;
	if (L.P == 0) return TE_OK;
	if (!L.bcgs_its.p && (rc = L.bcgs_its.alloc((size_t) L.P))) return rc;
	const size_t total = (size_t) L.P * L.nc;
	{
		Timed t(g, KC_PATCH_RHS, total);
		hipLaunchKernelGGL(k_patch_rhs2d, dim3(gridFor(total, 256, 65536)), dim3(256), 0, g->stream, L.dev2(), u, f, s0);
	}
	const int    cpt = L.n <= 16 ? 1 : (L.n <= 32 ? 4 : 16), spr = (L.n + cpt - 1) / cpt;
	const size_t lds = sizeof(double) * ((size_t) L.n * spr * (cpt > 1 ? cpt + 2 : 1) + (cpt >= 16 ? 256 * cpt : 0)); // tile (+ rhat)
	const bool   full = L.n % cpt == 0;
	Timed        t(g, KC_PATCH_BCGS, total);
#define TE_BCGS(C, F) \
	hipLaunchKernelGGL((k_patch_bcgs2d<C, F>), dim3(L.P), dim3(256), lds, g->stream, L.dev2(), s0, u, g->bcgs_tol, g->bcgs_max_it, L.bcgs_its.p)
	if (cpt == 16 && !g->bcgs_attr) { // up to 68 KiB of dynamic LDS
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_patch_bcgs2d<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_patch_bcgs2d<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
		g->bcgs_attr = true;
	}
	if (cpt == 1)
		TE_BCGS(1, true);
	else if (cpt == 4) {
		if (full)
			TE_BCGS(4, true);
		else
			TE_BCGS(4, false);
	} else {
		if (full)
			TE_BCGS(16, true);
		else
			TE_BCGS(16, false);
	}
#undef TE_BCGS
	HIPCHK(hipGetLastError());
	return TE_OK;
}

int restrict2d(te_gmg *g, LevelHost &L, const double *fine, double *coarse)
{
	if (L.n_up > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_up * L.nc / 4);
		hipLaunchKernelGGL(k_restrict_pack2d, dim3(L.n_up), dim3(256), 0, g->stream, L.n, L.up_desc.p, L.up_off.p, fine, L.upbuf.p);
	}
	int rc = doExchange(g, 2, L.tx_up, L.upbuf.p, L.downbuf.p);
	if (rc) return rc;
	if (L.Pc == 0) return TE_OK;
	Timed t(g, KC_RESTRICT, (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_restrict2d, dim3(gridFor((size_t) L.Pc * L.nc, 256, 65536)), dim3(256), 0, g->stream, L.n, L.Pc,
	                   L.child.p, L.copy.p, fine, L.downbuf.p, L.down_off.p, coarse);
	HIPCHK(hipGetLastError());
	return TE_OK;
}

int prolong2d(te_gmg *g, LevelHost &L, const double *coarse, double *fine)
{
	if (L.n_down > 0 && !L.repl_up) { // (above a level that lives on every rank nothing travels back up: every parent is local)
		Timed t(g, KC_PACK, (size_t) L.n_down * L.nc / 4);
		hipLaunchKernelGGL(k_prolong_pack2d, dim3(L.n_down), dim3(256), 0, g->stream, L.n, L.down_desc.p, L.down_off.p, coarse,
		                   L.downbuf.p);
	}
	int rc = doExchange(g, 3, L.tx_down, L.downbuf.p, L.upbuf.p);
	if (rc) return rc;
	if (L.P == 0) return TE_OK;
	Timed t(g, KC_PROLONG, (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_prolong2d, dim3(gridFor((size_t) L.P * L.nc, 256, 65536)), dim3(256), 0, g->stream, L.n, L.P, L.parent.p,
	                   L.orth.p, coarse, L.upbuf.p, L.up_off.p, fine);
	HIPCHK(hipGetLastError());
	return TE_OK;
}

// opts.fuse = 2 / 3 in 2D (levels with L.fuse2_ok: patches in LDS, every parent and neighbour local): see kernels2d.hpp
int zeroSweepResid2d(te_gmg *g, LevelHost &L, const double *f, double *out, double *coarse, bool store_u, const Fold2DHost *fold_in,
                     bool skip_fixup)
{
	const size_t lds = sizeof(double) * ((size_t) (L.n + 2) * (L.n + 2) + 16);
	Prolong2D    dst{L.parent.p, L.orth.p, nullptr};
	int          rc;
	if (L.P > 0) {
		Timed t(g, store_u ? KC_ZERO_RESID : KC_ZERO_RESID_FACES, (size_t) L.P * L.nc, true);
		// fold_in: this level's right-hand side still lacks the ghost terms of the finer level's restricted residual; the kernel
		// adds them to f (this solver's own coarse vector) before it reads it
		Fold2D fold = Fold2D();
		if (fold_in && fold_in->fine) {
			LevelHost &Lf = *fold_in->fine;
			if (Lf.Pc != L.P || Lf.n != L.n || !Lf.child.p) return te::fail(TE_ESTATE, "zeroSweepResid2d: folded fix-up from a level that is not this level's finer one");
			fold.fine  = Lf.dev2();
			fold.u     = fold_in->u;
			fold.e4    = fold_in->u ? nullptr : Lf.e4buf.p;
			fold.child = Lf.child.p;
		}
		const bool fo = fold.child != nullptr;
		double    *fw = const_cast<double *>(f);
#define TE_ZR2(S, NC, T)                                                                                                           \
	do {                                                                                                                           \
		if (fo)                                                                                                                    \
			launchT(t, (k_rbgs_zero_resid2d_lds<S, NC, T, true>), dim3(L.P), dim3(T), lds, g->stream, L.dev2(), fw, out, L.e4buf.p, dst, \
			        coarse, L.upbuf.p, L.up_off.p, fold);                                                                          \
		else                                                                                                                       \
			launchT(t, (k_rbgs_zero_resid2d_lds<S, NC, T, false>), dim3(L.P), dim3(T), lds, g->stream, L.dev2(), f, out, L.e4buf.p, dst, \
			        coarse, L.upbuf.p, L.up_off.p, Fold2D());                                                                      \
	} while (0)
		if (L.n == 64 && tpb2d(g) == 512) {
			if (store_u)
				TE_ZR2(true, 64, 512);
			else
				TE_ZR2(false, 64, 512);
		} else if (store_u && L.n == 64)
			TE_ZR2(true, 64, 256);
		else if (store_u)
			TE_ZR2(true, 0, 256);
		else if (L.n == 64)
			TE_ZR2(false, 64, 256);
		else
			TE_ZR2(false, 0, 256);
#undef TE_ZR2
	}
	if (L.nremote > 0) { // the new edge layers of neighbours on other ranks
		{
			Timed t(g, KC_PACK, (size_t) L.nremote * L.nf);
			if (store_u)
				hipLaunchKernelGGL(k_pack_faces2d, dim3(L.nremote), dim3(64), 0, g->stream, L.n, L.send_faces.p, out, L.sendbuf.p);
			else
				hipLaunchKernelGGL(k_pack_edges2d, dim3(L.nremote), dim3(64), 0, g->stream, L.n, L.send_faces.p, L.e4buf.p, L.sendbuf.p);
		}
		if ((rc = faceExchange(g, L, L.sendbuf.p))) return rc;
	}
	if (L.P > 0 && !skip_fixup) { // (skip_fixup: the coarser level's pre-sweep adds the terms itself, Fold2DHost)
		Timed t(g, KC_FIXUP, (size_t) L.P * 4 * L.nf);
		hipLaunchKernelGGL(k_restrict_fixup2d, dim3(L.P), dim3(128), 0, g->stream, L.dev2(), out,
		                   store_u ? (const double *) nullptr : (const double *) L.e4buf.p, dst, coarse, L.upbuf.p, L.up_off.p);
	}
	// children whose parent lives on another rank, or a coarse level on every rank: ship the finished blocks
	HIPCHK(hipGetLastError());
	return shipRestricted2d(g, L, coarse);
}

int resweepProlong2d(te_gmg *g, LevelHost &L, const double *f, double *out, const double *prolong_from)
{
	int rc = packProlongFaces2d(g, L, nullptr, L.e4buf.p, Prolong2D{L.parent.p, L.orth.p, prolong_from});
	if (rc) return rc;
	if (L.P == 0) return TE_OK;
	const size_t lds = sizeof(double) * ((size_t) (L.n + 2) * (L.n + 2) + 16);
	Timed        t(g, KC_RESWEEP, (size_t) L.P * L.nc, true);
	if (L.n == 64 && tpb2d(g) == 512)
		launchT(t, (k_rbgs_resweep_prolong2d_lds<64, 512>), dim3(L.P), dim3(512), lds, g->stream, L.dev2(), f, L.e4buf.p, out,
		                   Prolong2D{L.parent.p, L.orth.p, prolong_from});
	else if (L.n == 64)
		launchT(t, k_rbgs_resweep_prolong2d_lds<64>, dim3(L.P), dim3(256), lds, g->stream, L.dev2(), f, L.e4buf.p, out,
		                   Prolong2D{L.parent.p, L.orth.p, prolong_from});
	else
		launchT(t, k_rbgs_resweep_prolong2d_lds<0>, dim3(L.P), dim3(256), lds, g->stream, L.dev2(), f, L.e4buf.p, out,
		                   Prolong2D{L.parent.p, L.orth.p, prolong_from});
	HIPCHK(hipGetLastError());
	return TE_OK;
}

template int launchStencil2d<MODE_APPLY>(te_gmg *, LevelHost &, const double *, const double *, double *, double, int, const double *, int *);
template int launchStencil2d<MODE_RESID>(te_gmg *, LevelHost &, const double *, const double *, double *, double, int, const double *, int *);
template int launchStencil2d<MODE_JACOBI>(te_gmg *, LevelHost &, const double *, const double *, double *, double, int, const double *, int *);
} // namespace tei

