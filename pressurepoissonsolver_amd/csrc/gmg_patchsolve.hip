// The reference smoother's sweep in 3D: interface terms + exact patch solves (FFTBlockJacobiSmoother.h:55-58, FftwPatchSolver.h:173-206)
// on the fp64 matrix cores (see gmg_internal.hpp).
#include "gmg_ghosts3d.hpp"

namespace tei
{
template <int N> int patchSolveN(te_gmg *g, LevelHost &L, const double *f, double *u, double *s0, double *s1,
                                 bool zero_guess, const double *prolong_from)
{
	const size_t total = (size_t) L.P * L.nc;
	int          rc;
	const bool   faces_req = L.ps_faces_req && zero_guess; // (a request holds for the very next sweep only)
	L.ps_faces_req         = false;
	if (!g->cfg.has(O_PS_SLOW)) { // (3D patches are 4, 8, 16 or 32 cells wide)
		// matrix-core path (patchsolve32.hpp; 16^3 patches: patchsolve16.hpp): interface terms on the face layers only, then x,y forward
		// per plane; z forward + eigenvalue divide + z inverse; x,y inverse. A zero initial guess has no
		// interface term (gamma = 0) and u is overwritten without being read.
		// few patches: the three-pass kernels, each patch spread over `seg` workgroups (one patch per CU would
		// leave most of the chip idle and a single solve takes ~80 us); otherwise the single-pass kernel
		// (TE_PS_MODE = 1pass | 1pass-dense | 3pass pins the choice; 3pass also pins one workgroup per patch: tests)
		const char *mode     = g->cfg.str(O_PS_MODE);
		// (the GLOBAL patch count decides: k_ps_sym and the three-pass kernels differ in the last bits, and a sharded run must
		// take the arithmetic path of the single-rank run -- 512^3 on 8 ranks has 64 local patches of 512 on level 1)
		const bool  one_pass = mode ? !strncmp(mode, "1pass", 5) : L.P_global >= 256;
		const int   seg      = (one_pass || mode) ? 1 : (L.P >= 128 ? 2 : (L.P >= 64 ? 4 : 8));
		const dim3 gp(L.P, seg), b256(256);
		// x-face columns of the old iterate, if its producer exported them (te_vcycle only: see xfFor)
		const double *xf_in = (g->in_cycle && !zero_guess) ? xfFor(L, u) : nullptr;
		L.xf_valid_for      = nullptr; // u is rewritten in place
		if (!zero_guess) {
			ProlongSrc ps;
			ps.parent = L.parent.p;
			ps.orth   = L.orth.p;
			ps.coarse = prolong_from;
			L.pack_f6 = L.ps_faces ? L.f6buf.p : nullptr;
			rc        = prepareGhosts<N>(g, L, u, prolong_from ? &ps : nullptr);
			L.pack_f6 = nullptr;
			if (rc) return rc;
			Timed    t(g, KC_PATCH_RHS, (size_t) L.P * 6 * L.nf);
			LevelDev D = L.dev();
			D.xf       = L.ps_faces ? nullptr : xf_in;
			D.f6       = L.ps_faces ? L.f6buf.p : nullptr;
			L.ps_faces = false; // (this sweep rewrites the whole iterate)
			if (prolong_from)
				hipLaunchKernelGGL((k_face_corr3d<N, true>), dim3(L.P * 6), b256, 0, g->stream, D, u, L.corr.p, ps);
			else
				hipLaunchKernelGGL((k_face_corr3d<N, false>), dim3(L.P * 6), b256, 0, g->stream, D, u, L.corr.p, ps);
		}
		if constexpr (N <= 8) { // 4^3 / 8^3 patches: one launch as well (vector units: k_ps_small)
			Timed t(g, KC_DST, total, true);
			if (zero_guess)
				launchT(t, (k_ps_small<N, false>), dim3(L.P), b256, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, L.zero_mode.p, L.rh2.p, f,
				        (const double *) nullptr, u);
			else
				launchT(t, (k_ps_small<N, true>), dim3(L.P), b256, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, L.zero_mode.p, L.rh2.p, f,
				        (const double *) L.corr.p, u);
			HIPCHK(hipGetLastError());
			return TE_OK;
		}
		if constexpr (N == 16) { // the whole solve of a 16^3 patch in one launch, the patch in LDS (k_ps16)
			Timed t(g, KC_PS_MFMA, total, true);
			if (zero_guess)
				launchT(t, k_ps16<false>, dim3(L.P), b256, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, L.zero_mode.p, L.rh2.p, f,
				        (const double *) nullptr, u);
			else
				launchT(t, k_ps16<true>, dim3(L.P), b256, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, L.zero_mode.p, L.rh2.p, f,
				        (const double *) L.corr.p, u);
			HIPCHK(hipGetLastError());
			return TE_OK;
		}
		if (one_pass) { // the whole solve in one pass over HBM (k_ps_fused)
			bool &lds_ok = g->ps_lds_ok;
			int  &ncu    = g->ncu;
			if (!lds_ok) {
				int dev = 0;
				HIPCHK(hipGetDevice(&dev));
				HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
				HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_fused<false>),
				                           hipFuncAttributeMaxDynamicSharedMemorySize, PSF_LDS_BYTES));
				HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_fused<true>),
				                           hipFuncAttributeMaxDynamicSharedMemorySize, PSF_LDS_BYTES));
				HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_sym<false>),
				                           hipFuncAttributeMaxDynamicSharedMemorySize, PSS_LDS_BYTES));
				HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_sym<true>),
				                           hipFuncAttributeMaxDynamicSharedMemorySize, PSS_LDS_BYTES));
				HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_sym<false, true>),
				                           hipFuncAttributeMaxDynamicSharedMemorySize, PSS_LDS_BYTES));
				lds_ok = true;
			}
			// (decided here as well as below: the class of the launch is part of its timing scope)
			const bool dense_only0 = mode && !strcmp(mode, "1pass-dense");
			const bool faces0 = faces_req && L.f6buf.p && !dense_only0 && (L.sym_ok ? L.P : L.n_pure) == L.P;
			Timed         t(g, faces0 ? KC_PS_MFMA_FACES : KC_PS_MFMA, total, true);
			const dim3    b512(512);
			const double *cp = zero_guess ? (const double *) nullptr : (const double *) L.corr.p;
			// pure axes: half-size transforms, one resident workgroup per CU walks over the patches (k_ps_sym);
			// patches with a mixed Dirichlet/Neumann axis: full transforms, one workgroup per patch (k_ps_fused)
			const bool dense_only = mode && !strcmp(mode, "1pass-dense");
			const int  n_sym = dense_only ? 0 : (L.sym_ok ? L.P : L.n_pure), n_mix = L.P - n_sym;
			const int32_t *lst_sym = (n_sym > 0 && n_mix > 0) ? L.ps_list.p : nullptr;
			const int32_t *lst_mix = (n_sym > 0 && n_mix > 0) ? L.ps_list.p + n_sym : nullptr;
			if (n_sym > 0) {
				const dim3 gs(std::min(n_sym, ncu));
				double    *xo = (g->in_cycle && n_mix == 0 && !g->no_xf_export) ? L.xfbuf[L.xf_cur ^ 1].p : nullptr; // (k_ps_fused does not export)
				const bool faces = faces_req && n_mix == 0 && L.f6buf.p;
				if (faces) { // only the face layers of the result: see k_ps_sym<CORR, FACES>
					L.f6_tab = false; // (written as [p][6])
					launchT(t, (k_ps_sym<false, true>), gs, b512, PSS_LDS_BYTES, g->stream, n_sym, L.plan.p, L.matsym.p, L.psinv.p, L.psitab.p, f, cp, u, (double *) nullptr, lst_sym, L.f6buf.p);
					L.ps_faces = true;
					xo         = nullptr;
				} else if (zero_guess)
					launchT(t, (k_ps_sym<false, false>), gs, b512, PSS_LDS_BYTES, g->stream, n_sym, L.plan.p, L.matsym.p, L.psinv.p, L.psitab.p, f, cp, u, xo, lst_sym, (double *) nullptr);
				else
					launchT(t, (k_ps_sym<true, false>), gs, b512, PSS_LDS_BYTES, g->stream, n_sym, L.plan.p, L.matsym.p, L.psinv.p, L.psitab.p, f, cp, u, xo, lst_sym, (double *) nullptr);
				if (xo) xfProduced(L, u);
			}
			if (n_mix > 0) {
				const dim3 gf(8 * ((n_mix + 7) / 8));
				if (zero_guess)
					launchT(t, k_ps_fused<false>, gf, b512, PSF_LDS_BYTES, g->stream, n_mix, L.plan.p, L.mats.p, L.lam.p,
					                   L.zero_mode.p, L.rh2.p, f, cp, u, lst_mix);
				else
					launchT(t, k_ps_fused<true>, gf, b512, PSF_LDS_BYTES, g->stream, n_mix, L.plan.p, L.mats.p, L.lam.p,
					                   L.zero_mode.p, L.rh2.p, f, cp, u, lst_mix);
			}
			HIPCHK(hipGetLastError());
			return TE_OK;
		}
		if (!one_pass && !mode && L.P <= g->cfg.num(O_PS_HALF_MAX, 8) && !g->cfg.has(O_PS_NO_HALF)) { // at most 8 patches: one wave per workgroup and half plane (k_ps_xy_half), bit-identical
			const dim3 gh(L.P, 64), b64(64);
			{
				Timed t(g, KC_PS_3PASS, total);
				if (zero_guess)
					hipLaunchKernelGGL((k_ps_xy_half<false, false>), gh, b64, 0, g->stream, L.P, L.plan.p, L.matfrag.p, f,
					                   (const double *) nullptr, s1 TE_STAMP_ARG(g, "ps xy forward (half)", gh.x * gh.y));
				else
					hipLaunchKernelGGL((k_ps_xy_half<false, true>), gh, b64, 0, g->stream, L.P, L.plan.p, L.matfrag.p, f,
					                   (const double *) L.corr.p, s1 TE_STAMP_ARG(g, "ps xy forward (half)", gh.x * gh.y));
			}
			{
				Timed t(g, KC_PS_3PASS, total);
				hipLaunchKernelGGL(k_ps_z_half, gh, b64, 0, g->stream, L.P, L.plan.p, L.matfrag.p, L.lam.p, L.zero_mode.p, L.rh2.p, s1,
				                   s0 TE_STAMP_ARG(g, "ps z (half)", gh.x * gh.y));
			}
			{
				Timed t(g, KC_PS_3PASS, total);
				hipLaunchKernelGGL(k_ps_xy_half<true>, gh, b64, 0, g->stream, L.P, L.plan.p, L.matfrag.p, s0, (const double *) nullptr,
				                   u TE_STAMP_ARG(g, "ps xy inverse (half)", gh.x * gh.y));
			}
			HIPCHK(hipGetLastError());
			return TE_OK;
		}
		{
			Timed t(g, KC_PS_3PASS, total);
			if (zero_guess)
				hipLaunchKernelGGL((k_ps_xy<false, false>), gp, b256, 0, g->stream, L.P, L.plan.p, L.matfrag.p, f,
				                   (const double *) nullptr, s1 TE_STAMP_ARG(g, "ps xy forward", gp.x * gp.y));
			else
				hipLaunchKernelGGL((k_ps_xy<false, true>), gp, b256, 0, g->stream, L.P, L.plan.p, L.matfrag.p, f,
				                   (const double *) L.corr.p, s1 TE_STAMP_ARG(g, "ps xy forward", gp.x * gp.y));
		}
		{
			Timed t(g, KC_PS_3PASS, total);
			hipLaunchKernelGGL(k_ps_z, gp, b256, 0, g->stream, L.P, L.plan.p, L.matfrag.p, L.lam.p, L.zero_mode.p, L.rh2.p, s1, s0 TE_STAMP_ARG(g, "ps z", gp.x * gp.y));
		}
		{
			Timed t(g, KC_PS_3PASS, total);
			hipLaunchKernelGGL(k_ps_xy<true>, gp, b256, 0, g->stream, L.P, L.plan.p, L.matfrag.p, s0, (const double *) nullptr, u TE_STAMP_ARG(g, "ps xy inverse", gp.x * gp.y));
		}
		HIPCHK(hipGetLastError());
		return TE_OK;
	}
	L.xf_valid_for = nullptr; // u is rewritten in place
	if (zero_guess) {
		Timed t(g, KC_VECOP, total);
		HIPCHK(hipMemsetAsync(u, 0, sizeof(double) * total, g->stream));
	}
	if ((rc = prepareGhosts<N>(g, L, u))) return rc;
	{
		Timed t(g, KC_PATCH_RHS, total);
		hipLaunchKernelGGL(k_patch_rhs3d<N>, dim3(gridFor(total, 256)), dim3(256), 0, g->stream, L.dev(), u, f, s0);
	}
	constexpr int BPP = (N * N * N + 255) / 256;
	const dim3    grid(L.P * BPP), blk(256);
#define TE_DST(STAGE, IN, OUT)                                                                                \
	{                                                                                                         \
		Timed t(g, KC_DST, total);                                                                                 \
		hipLaunchKernelGGL((k_dst_axis3d<N, STAGE>), grid, blk, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, \
		                   L.zero_mode.p, L.rh2.p, IN, OUT);                                                  \
	}
	TE_DST(0, s0, s1)
	TE_DST(1, s1, s0)
	TE_DST(2, s0, s1)
	TE_DST(3, s1, s0)
	TE_DST(4, s0, s1)
	TE_DST(5, s1, u)
#undef TE_DST
	HIPCHK(hipGetLastError());
	return TE_OK;
}

int patchSolve(te_gmg *g, LevelHost &L, const double *f, double *u, bool zero_guess, const double *prolong_from, bool *swapped)
{
	bool dummy;
	if (!swapped) swapped = &dummy;
	*swapped = false;
	if (L.P == 0) return TE_OK; // (no patches: no faces towards other ranks either)
	double *s0 = L.r->d, *s1 = L.t->d;
	if (L.dim == 2) {
		int rc = patchSolve2d(g, L, f, u, s0, s1, zero_guess, swapped, prolong_from);
		if (rc == TE_OK && *swapped && swapped == &dummy) return te::fail(TE_ESTATE, "patchSolve: 2D result left in scratch");
		return rc;
	}
	switch (L.n) {
		case 4: return patchSolveN<4>(g, L, f, u, s0, s1, zero_guess, prolong_from);
		case 8: return patchSolveN<8>(g, L, f, u, s0, s1, zero_guess, prolong_from);
		case 16: return patchSolveN<16>(g, L, f, u, s0, s1, zero_guess, prolong_from);
		default: return patchSolveN<32>(g, L, f, u, s0, s1, zero_guess, prolong_from);
	}
}
} // namespace tei
