// 3D kernel launches of the fused cycle (opts.fuse = 2, 3): sweep from zero + residual + restriction, the recomputing post-sweep,
// the interface-only residual of the block-Jacobi smoother (see gmg_internal.hpp).
#include "gmg_ghosts3d.hpp"

namespace tei
{
// opts.fuse = 2: first pre-smoothing sweep from a zero iterate + residual + restriction (march3d.hpp,
// k_rbgs_zero_resid3d / k_restrict_fixup3d). out = S(0, f) with its x faces in xf_out, coarse = AvgRstr(f - A out).
// store_u = false (opts.fuse = 3): the new iterate is left in L.f6buf as its six face layers only
template <int N>
int zeroSweepResidN(te_gmg *g, LevelHost &L, const double *f, double *out, double *coarse, double *xf_out, bool store_u,
                    double *fcorr_out, const double *fcorr_in, const PendingRhs *fs)
{
	RestrictDst rd = RestrictDst();
	rd.parent     = L.parent.p;
	rd.orth       = L.orth.p;
	rd.coarse     = coarse;
	rd.remote     = L.upbuf.p;
	rd.remote_off = L.up_off.p;
	// the patches export the 2x2 sums of their face layers (rs6); what lies behind a ghost slot, and a copy-through patch's own faces,
	// the gather kernel forms from the slot / the face layers
	// (round 6: on refined levels too -- the kernel's export does not look at what a patch is, a same-level neighbour's finished sums
	// are what the gather needs behind every FACE_LOCAL face, and they are a quarter of the face layers it read instead:
	// 141 -> 107 (descriptors) -> ... us per launch on `2refine --divide 3`; TE_NO_RS6_CF: as before)
	// ... and (round 6, last) also when a fix-up pass follows instead of the gather (the coarser level is not a fused one): it reads the
	// finished sums of same-level neighbours instead of their face layers (TE_NO_RS6_FIXUP: the layers, as before). Not on a level that
	// itself reads exported terms (fcorr_in): the kernel variant that does both is slower by more than the fix-up gains (measured at
	// 512^3, level 1: +3.5 us per cycle; 256^3, level 0, without fcorr_in: -2.3 us)
	const bool export_rs6 = L.rs6.p && !store_u
	                        && (fcorr_out ? (L.prolong_fusable || !g->cfg.has(O_NO_RS6_CF))
	                                      : (L.Pc > 0 || L.n_up > 0) && !fcorr_in && !g->cfg.has(O_NO_RS6_FIXUP));
	rd.rs6                = export_rs6 ? L.rs6.p : nullptr;
	int rc;
	if (L.P > 0) {
		Timed      t(g, store_u ? KC_ZERO_RESID : (fcorr_in ? KC_ZERO_RESID_FACES_FCORR : KC_ZERO_RESID_FACES), (size_t) L.P * L.nc, true);
		if (!store_u) L.f6_tab = L.f6off.p != nullptr && !g->cfg.has(O_PACK_FACES); // the face layers go where the level's table puts them
		LevelDev   D = L.dev();
		const dim3 grid(8 * ((L.P + 7) / 8)), blk(Tile3<N>::TPB);
		if (store_u) {
			D.xf_out = xf_out;
			launchT(t, (k_rbgs_zero_resid3d<N, true>), grid, blk, 0, g->stream, D, f, out, rd, FSrc());
		} else {
			D.f6_out = L.f6buf.p;
			D.fcorr  = fcorr_in;
			// TE_ZR_AHEAD = 1: the right-hand side requested one plane ahead only (the form before round 3; bit-identical)
			const bool ah1 = g->cfg.num(O_ZR_AHEAD, 4) == 1;
#define TE_ZR(EXP, FC)                                                                                                   \
	if (ah1)                                                                                                             \
		launchT(t, (k_rbgs_zero_resid3d<N, false, EXP, FC, 1>), grid, blk, 0, g->stream, D, f, out, rd, FSrc());         \
	else                                                                                                                 \
		launchT(t, (k_rbgs_zero_resid3d<N, false, EXP, FC, 4>), grid, blk, 0, g->stream, D, f, out, rd, FSrc())
			if (fs) { // the right-hand side is a pending vector statement of te_bicgstab (march3d.hpp FSrc): formed and stored here
				const FSrc a = fs->args;
				if (fs->kind == 1 && export_rs6)
					launchT(t, (k_rbgs_zero_resid3d<N, false, true, false, 4, 1>), grid, blk, 0, g->stream, D, f, out, rd, a);
				else if (fs->kind == 1)
					launchT(t, (k_rbgs_zero_resid3d<N, false, false, false, 4, 1>), grid, blk, 0, g->stream, D, f, out, rd, a);
				else if (export_rs6)
					launchT(t, (k_rbgs_zero_resid3d<N, false, true, false, 4, 2>), grid, blk, 0, g->stream, D, f, out, rd, a);
				else
					launchT(t, (k_rbgs_zero_resid3d<N, false, false, false, 4, 2>), grid, blk, 0, g->stream, D, f, out, rd, a);
			} else if (export_rs6 && fcorr_in) {
				TE_ZR(true, true);
			} else if (export_rs6) {
				TE_ZR(true, false);
			} else if (fcorr_in) {
				TE_ZR(false, true);
			} else {
				TE_ZR(false, false);
			}
#undef TE_ZR
		}
	}
	// the new face layers of neighbours on other ranks (no-op on one rank)
	L.pack_f6 = store_u ? nullptr : L.f6buf.p;
	rc        = prepareGhosts<N>(g, L, out);
	L.pack_f6 = nullptr;
	if (rc) return rc;
	L.ghost_has_v = !store_u; // (the slots of neighbours on other ranks hold their face layers of v until the next exchange of the level)
	if (fcorr_out) { // the ghost terms were formed by the patches that own the face values: sort them into the coarse
		// level's side array (a permutation copy of 6/128 of a vector instead of the fix-up pass)
		if (L.Pc > 0 && !g->cfg.has(O_NO_GTAB2) && !g->cfg.has(O_NO_GTAB)) { // descriptors once per level, then 16-byte data loads only (k_fcorr_gather3d_v2)
			LevelDev  D   = L.dev();
			const int key = (export_rs6 ? 1 : 0) | (D.f6off ? 2 : 0);
			if (L.gdesc_key != key) { // (built once; again if the face layers change their layout: TE_PACK_FACES)
				int rc2 = L.gdesc.p ? TE_OK : L.gdesc.alloc((size_t) L.Pc * 48);
				if (rc2) return rc2;
				hipLaunchKernelGGL(k_gather_desc3d<N>, dim3(L.Pc), dim3(64), 0, g->stream, D, L.child.p, L.copy.p, export_rs6 ? 1 : 0, L.gdesc.p);
				L.gdesc_key = key;
			}
			if (L.fcorr_zeroed_for != fcorr_out) { // (planes without a term are never stored: k_fcorr_gather3d_v2; the other gathers store every entry)
				HIPCHK(hipMemsetAsync(fcorr_out, 0, sizeof(double) * (size_t) L.Pc * 4 * N * N, g->stream));
				L.fcorr_zeroed_for = fcorr_out;
			}
			Timed t(g, KC_FCORR_GATHER, (size_t) L.P * 6 * L.nf / 4);
			hipLaunchKernelGGL(k_fcorr_gather3d_v2<N>, dim3(L.Pc * 12), dim3(256), 0, g->stream, D, (const GatherDesc *) L.gdesc.p,
			                   export_rs6 ? (const double *) L.rs6.p : (const double *) nullptr, (const double *) L.f6buf.p, coarse, fcorr_out);
		} else if (L.Pc > 0) {
			const bool use_gtab = export_rs6 && !g->cfg.has(O_NO_GTAB);
			if (use_gtab && !L.gtab.p && (size_t) L.P * 6 * (N / 2) * (N / 2) < ((size_t) 1 << 31)) { // once per level
				int rc2 = L.gtab.alloc((size_t) L.Pc * 48);
				if (rc2) return rc2;
				hipLaunchKernelGGL(k_gather_table3d<N>, dim3(L.Pc), dim3(64), 0, g->stream, L.dev(), L.child.p, L.copy.p, L.gtab.p);
			}
			Timed t(g, KC_FCORR_GATHER, (size_t) L.P * 6 * L.nf / 4);
			hipLaunchKernelGGL(k_fcorr_gather3d<N>, dim3(L.Pc * 12), dim3(256), 0, g->stream, L.dev(), L.child.p, L.copy.p,
			                   export_rs6 ? (const double *) L.rs6.p : (const double *) nullptr, (const double *) L.f6buf.p, coarse, fcorr_out,
			                   use_gtab ? (const int32_t *) L.gtab.p : (const int32_t *) nullptr);
		}
	} else if (L.P > 0) {
		Timed    t(g, KC_FIXUP, (size_t) L.P * 6 * L.nf);
		LevelDev D = L.dev();
		if (store_u)
			D.xf = xf_out;
		else
			D.f6 = L.f6buf.p;
		hipLaunchKernelGGL((k_restrict_fixup3d<N, false>), dim3(L.P), dim3(256), 0, g->stream, D, out, rd);
	}
	// children whose parent lives on another rank: ship the finished blocks (as residRestrictN)
	if ((rc = shipRestricted<N>(g, L, coarse))) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}

// opts.fuse = 3, post-smoothing: out = S(v + P(prolong_from), f) with v = S(0, f) recomputed (its faces in L.f6buf)
template <int N>
int resweepProlongN(te_gmg *g, LevelHost &L, const double *f, double *out, const double *prolong_from, double *xf_out, const double *fcorr_in)
{
	ProlongSrc ps;
	ps.parent = L.parent.p;
	ps.orth   = L.orth.p;
	ps.coarse = prolong_from;
	auto launch = [&](LevelDev D) {
		if (D.count == 0) return;
		Timed t(g, fcorr_in ? KC_RESWEEP_FCORR : KC_RESWEEP, (size_t) D.count * L.nc, true);
		D.f6    = L.f6buf.p;
		D.fcorr = fcorr_in;
		if constexpr (N >= 4) {
			const dim3  grid(8 * ((D.count + 7) / 8)), blk(Tile3<N>::TPB);
			// tuning variants (march3d.hpp), all bit-identical. Defaults, each measured: the finest level stores u non-temporally
			// (nobody re-reads it) and loads f non-temporally unless f can still be in the Infinity Cache from the pre-sweep
			// that read it (19 instead of 27: 256^3, a rank's share at eight ranks; 53.4 -> 49.0 us at 256^3); a large finest
			// level runs two workgroups per CU with four planes of f in flight each instead of three with two (59: 450 -> 443 us
			// at 512^3, same box; slower at 256^3); a coarser level with exported ghost terms (level 1 of 512^3) keeps ordinary
			// stores as well -- its u is the correction the finer level's post-sweep reads next (3: 64.5 us, against 68.7 with
			// non-temporal stores)
			const char *ve    = g->cfg.str(O_RESWEEP_V);
			const bool  small = (size_t) L.P * L.nc * sizeof(double) <= ((size_t) 160 << 20);
			const int   v     = ve ? atoi(ve) : (fcorr_in ? 3 : (g->cur_level != 0 ? 27 : (small ? 19 : 59)));
			if (L.ncf > 0 || L.has_copy) { // refined level: copy-through patches / coarse-fine ghost slots
				if (v == 3)
					launchT(t, (k_rbgs_resweep_prolong3d<N, 3, false, true>), grid, blk, 0, g->stream, D, f, out, ps);
				else if (v == 59 && !g->cfg.has(O_NO_CFP59))
					launchT(t, (k_rbgs_resweep_prolong3d<N, 59, false, true>), grid, blk, 0, g->stream, D, f, out, ps);
				else
					launchT(t, (k_rbgs_resweep_prolong3d<N, 27, false, true>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (fcorr_in) {
				if (v == 0)
					launchT(t, (k_rbgs_resweep_prolong3d<N, 0, true>), grid, blk, 0, g->stream, D, f, out, ps);
				else if (v == 27)
					launchT(t, (k_rbgs_resweep_prolong3d<N, 27, true>), grid, blk, 0, g->stream, D, f, out, ps);
				else if (v == 19)
					launchT(t, (k_rbgs_resweep_prolong3d<N, 19, true>), grid, blk, 0, g->stream, D, f, out, ps);
				else
					launchT(t, (k_rbgs_resweep_prolong3d<N, 3, true>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 0) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 0, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 7) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 7, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 11) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 11, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 19) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 19, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 59) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 59, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 63) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 63, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 23) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 23, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 31) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 31, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 27) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 27, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 3) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 3, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 27, false>), grid, blk, 0, g->stream, D, f, out, ps);
			}
		}
	};
	if (L.post_exchange_free && L.ghost_has_v && L.ncf == 0 && !L.has_copy && !g->cfg.has(O_POST_EXCHANGE)) {
		// every neighbour's parent is local (the coarser level lives on every rank) and the ghost slots still hold the neighbours'
		// face layers of v: the kernel forms v + P e for them as for local neighbours; nothing travels (a global decision: all
		// ranks of the level take it together)
		ps.gparent    = L.slot_parent.p;
		ps.gorth      = L.slot_orth.p;
		L.ghost_has_v = false;
		LevelDev D    = L.dev();
		D.xf_out      = xf_out;
		launch(D);
		HIPCHK(hipGetLastError());
		return TE_OK;
	}
	L.pack_f6 = L.f6buf.p; // neighbours on other ranks receive the face layers of v + P(coarse)
	int rc    = withGhosts<N>(g, L, out /* unused: the faces come from pack_f6 */, launch, nullptr, xf_out, &ps);
	L.pack_f6 = nullptr;
	if (rc) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}

int resweepProlong(te_gmg *g, LevelHost &L, const double *f, double *out, const double *prolong_from, double *xf_out,
                   const double *fcorr_in)
{
	if (L.dim == 2) return resweepProlong2d(g, L, f, out, prolong_from);
	switch (L.n) {
		case 4: return resweepProlongN<4>(g, L, f, out, prolong_from, xf_out, fcorr_in);
		case 8: return resweepProlongN<8>(g, L, f, out, prolong_from, xf_out, fcorr_in);
		case 16: return resweepProlongN<16>(g, L, f, out, prolong_from, xf_out, fcorr_in);
		default: return resweepProlongN<32>(g, L, f, out, prolong_from, xf_out, fcorr_in);
	}
}

// opts.fuse = 2 with the block-Jacobi smoother: after an exact patch solve from the zero iterate the residual
// vanishes inside every patch (A_patch u = f is what was solved) and equals -(g + m)/h^2 = -2 gamma/h^2 on the face
// layers (the patch operator closes interface faces with ghost = -m, the level operator with the neighbour's g), so
// coarse f = AvgRstr(f - A u) is k_restrict_fixup3d<OWN> applied to a zeroed coarse vector: no pass over u and f
// at all. (What is dropped is the rounding noise of the solve, ~1e-13 |f|.) u: the new iterate, xf: its
// compact x faces or null; coarse: the coarse level's f with `coarse_n` entries.
template <int N> int interfaceResidRestrictN(te_gmg *g, LevelHost &L, const double *u, const double *xf, double *coarse, size_t coarse_n)
{
	RestrictDst rd = RestrictDst();
	rd.parent     = L.parent.p;
	rd.orth       = L.orth.p;
	rd.coarse     = coarse;
	rd.remote     = L.upbuf.p;
	rd.remote_off = L.up_off.p;
	int rc;
	L.pack_f6 = L.ps_faces ? L.f6buf.p : nullptr; // (the iterate exists only as its face layers)
	rc        = prepareGhosts<N>(g, L, u);
	L.pack_f6 = nullptr;
	if (rc) return rc;
	{
		Timed t(g, KC_VECOP, coarse_n);
		// (blocks exchanged in place: only this rank's run -- the others' runs are received, and with the direct-store transport a
		// peer that is ahead may have stored its run already)
		if (L.repl_up && L.repl_direct && !g->cfg.has(O_REPL_BLOCKS) && !L.tx_direct.empty()) {
			if (L.tx_direct.send_cnt[0] > 0)
				HIPCHK(hipMemsetAsync(coarse + L.tx_direct.send_off[0], 0, sizeof(double) * (size_t) L.tx_direct.send_cnt[0], g->stream));
		} else
			HIPCHK(hipMemsetAsync(coarse, 0, sizeof(double) * coarse_n, g->stream));
		if (L.n_up > 0) HIPCHK(hipMemsetAsync(L.upbuf.p, 0, sizeof(double) * L.upbuf.n, g->stream));
	}
	if (L.P > 0) {
		Timed    t(g, KC_FIXUP, (size_t) L.P * 6 * L.nf);
		LevelDev D = L.dev();
		D.xf       = L.ps_faces ? nullptr : xf;
		D.f6       = L.ps_faces ? L.f6buf.p : nullptr;
		hipLaunchKernelGGL((k_restrict_fixup3d<N, true>), dim3(L.P), dim3(256), 0, g->stream, D, u, rd);
	}
	if ((rc = shipRestricted<N>(g, L, coarse))) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}

int interfaceResidRestrict(te_gmg *g, LevelHost &L, const double *u, const double *xf, double *coarse, size_t coarse_n)
{
	switch (L.n) {
		case 4: return interfaceResidRestrictN<4>(g, L, u, xf, coarse, coarse_n);
		case 8: return interfaceResidRestrictN<8>(g, L, u, xf, coarse, coarse_n);
		case 16: return interfaceResidRestrictN<16>(g, L, u, xf, coarse, coarse_n);
		default: return interfaceResidRestrictN<32>(g, L, u, xf, coarse, coarse_n);
	}
}

int zeroSweepResid(te_gmg *g, LevelHost &L, const double *f, double *out, double *coarse, double *xf_out, bool store_u,
                   double *fcorr_out, const double *fcorr_in, const PendingRhs *fs, const Fold2DHost *fold_in, bool skip_fixup)
{
	if (L.dim == 2) return zeroSweepResid2d(g, L, f, out, coarse, store_u, fold_in, skip_fixup);
	if ((fold_in && fold_in->fine) || skip_fixup) return te::fail(TE_ESTATE, "zeroSweepResid: the folded fix-up exists in 2D only");
	switch (L.n) {
		case 4: return zeroSweepResidN<4>(g, L, f, out, coarse, xf_out, store_u, fcorr_out, fcorr_in, fs);
		case 8: return zeroSweepResidN<8>(g, L, f, out, coarse, xf_out, store_u, fcorr_out, fcorr_in, fs);
		case 16: return zeroSweepResidN<16>(g, L, f, out, coarse, xf_out, store_u, fcorr_out, fcorr_in, fs);
		default: return zeroSweepResidN<32>(g, L, f, out, coarse, xf_out, store_u, fcorr_out, fcorr_in, fs);
	}
}
} // namespace tei
