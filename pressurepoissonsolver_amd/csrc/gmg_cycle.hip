// The cycle driver (GMG/Cycle.h:56-126, VCycle.h:44-62, WCycle.h:45-68), the schedule check, te_gmg_autotune and the per-operation entry points (see gmg_internal.hpp).
#include "gmg_internal.hpp"

namespace tei
{
int smoothOnce(te_gmg *g, int level, const te_vec *f, te_vec *u, int smoother, double omega, bool zero_guess = false)
{
	LevelHost &L = *g->levels[level];
	int        rc;
	const bool xfok = (L.dim == 3 && g->in_cycle); // outside te_vcycle nobody keeps xf_valid_for honest
	switch (smoother) {
		case TE_SMOOTH_PATCH_SOLVE: { // keeps xf_valid_for itself
			bool swapped = false;
			rc           = patchSolve(g, L, f->d, u->d, zero_guess, nullptr, &swapped);
			if (rc == TE_OK && swapped) swapData(u, L.t.get()); // (2D: out of place)
			return rc;
		}
		case TE_SMOOTH_JACOBI:
			L.xf_valid_for = nullptr;
			rc = launchStencil<MODE_JACOBI>(g, L, u->d, f->d, L.t->d, omega);
			if (rc) return rc;
			swapData(u, L.t.get());
			return TE_OK;
		case TE_SMOOTH_RBGS:
			rc = launchRbgs(g, L, u->d, f->d, L.t->d, zero_guess, nullptr, xfok ? xfFor(L, u->d) : nullptr,
			                xfok ? L.xfbuf[L.xf_cur ^ 1].p : nullptr);
			if (rc) return rc;
			if (xfok) xfProduced(L, L.t->d);
			swapData(u, L.t.get());
			return TE_OK;
		case TE_SMOOTH_PATCH_BCGS: // (zero_guess never set: the smoothing lambda materialises the zeros first -- they ARE the initial guess)
			L.xf_valid_for = nullptr;
			return patchBcgs2d(g, L, f->d, u->d, L.t->d);
		default: return te::fail(TE_EINVAL, "te_smooth: unknown smoother");
	}
}

// GMG/VCycle.h:44-62, GMG/WCycle.h:45-68, GMG/Cycle.h:56-90.
// `u_zero`: u is logically zero on entry but has NOT been written yet (fused mode): the first RB-GS
// sweep then runs its zero-guess variant and the 8 B/site zero-fill never happens; any other first
// consumer materialises the zeros first. Results are bit-identical to the unfused sequence.
int visit(te_gmg *g, const te_cycle_opts *o, int l, const te_vec *f, te_vec *u, bool u_zero)
{
	const int  nl       = (int) g->levels.size();
	const bool coarsest = (l == nl - 1);
	LevelHost &L        = *g->levels[l];
	int        rc;
	g->cur_level        = l;
	const double *fcorr_in = (L.f_has_corr && L.fcorr.p) ? L.fcorr.p : nullptr; // ghost terms that still belong to f (see below)
	L.f_has_corr           = false;
	const Fold2DHost fold_in = L.fold_pending; // (2D) ghost terms this level's pre-sweep kernel still has to add to f itself
	L.fold_pending           = Fold2DHost();
	if (fold_in.fine && coarsest) return te::fail(TE_ESTATE, "te_vcycle: folded ghost terms without a reader");
	// te_bicgstab may hand over a right-hand side that is still a pending vector statement (PendingRhs): the fused pre-sweep
	// of level 0 forms it while reading its operands; every other path runs the stand-alone kernel first
	const PendingRhs *pend = (l == 0) ? g->pending_rhs : nullptr;
	if (l == 0) g->pending_rhs = nullptr;
	auto formRhs = [&]() -> int {
		if (!pend) return TE_OK;
		const PendingRhs &r = *pend;
		pend                = nullptr;
		if (r.n2 == 0 || g->recording) return TE_OK;
		Timed      t(g, r.kind == 1 ? KC_BICG_S : KC_BICG_P, r.n2 * 2);
		const dim3 grid(gridFor(r.n2, 256, 1 << 30));
		if (r.kind == 1)
			hipLaunchKernelGGL(k_bicg_s, grid, dim3(256), 0, g->stream, r.n2, (double2 *) r.args.out, (const double2 *) r.args.a,
			                   (const double2 *) r.args.b, r.args.s1);
		else
			hipLaunchKernelGGL(k_bicg_p, grid, dim3(256), 0, g->stream, r.n2, (double2 *) r.args.out, (const double2 *) r.args.b,
			                   (const double2 *) r.args.c, r.args.s1, r.args.s2);
		HIPCHK(hipGetLastError());
		return TE_OK;
	};
	auto       materialise = [&]() -> int {
        if (!u_zero) return TE_OK;
        u_zero         = false;
        L.xf_valid_for = nullptr;
        return vecop<VOP_SET>(u, nullptr, nullptr, 0.0, 0.0, 0.0);
	};
	const double *pending_prolong = nullptr; // coarse correction still to be added to u
	bool          u_unstored      = false;   // opts.fuse = 3: u = S(0, f) exists only as its face layers (L.f6buf)
	int           next_sweeps     = 0;       // sweeps that follow the descend() in progress
	// final_call: the post-smoothing of this level -- nobody reads the x-face columns of its last sweep's result
	// (they serve the NEXT kernel on the same level), so that sweep does not export them
	auto smooth = [&](int sweeps, bool at_coarsest, bool final_call = false) -> int {
		int sm = o->smoother;
		if (at_coarsest && o->exact_coarse && L.P_global == 1) sm = TE_SMOOTH_PATCH_SOLVE;
		for (int i = 0; i < sweeps; i++) {
			int r;
			if (pending_prolong) {
				const double *c = pending_prolong;
				pending_prolong = nullptr;
				const bool last = final_call && i == sweeps - 1 && !(l == 0 && g->keep_final_xf);
				if (sm == TE_SMOOTH_PATCH_SOLVE) { // reads u + P c on the face layers only, then overwrites u
					bool swapped    = false;
					g->no_xf_export = last;
					r               = patchSolve(g, L, f->d, u->d, false, c, &swapped);
					g->no_xf_export = false;
					if (r) return r;
					if (swapped) swapData(u, L.t.get()); // (2D: out of place)
					continue;
				}
				double *xo = last ? nullptr : L.xfbuf[L.xf_cur ^ 1].p;
				if (u_unstored) {
					u_unstored = false;
					if ((r = resweepProlong(g, L, f->d, L.t->d, c, xo, fcorr_in))) return r;
				} else if ((r = launchRbgs(g, L, u->d, f->d, L.t->d, false, c, xfFor(L, u->d), xo))) {
					return r;
				}
				if (xo)
					xfProduced(L, L.t->d);
				else
					L.xf_valid_for = nullptr;
				swapData(u, L.t.get());
				continue;
			}
			if (u_zero && ((L.dim == 3 && (sm == TE_SMOOTH_RBGS || sm == TE_SMOOTH_PATCH_SOLVE)) || (L.lds2d && sm == TE_SMOOTH_RBGS)
			               || (L.dim == 2 && sm == TE_SMOOTH_PATCH_SOLVE))) {
				u_zero = false;
				r      = smoothOnce(g, l, f, u, sm, o->omega, true);
			} else {
				if ((r = materialise())) return r;
				r = smoothOnce(g, l, f, u, sm, o->omega);
			}
			if (r) return r;
		}
		return TE_OK;
	};
	if (coarsest) {
		if ((rc = formRhs())) return rc;
		if ((rc = smooth(o->coarse_sweeps, true))) return rc;
		return materialise();
	}
	LevelHost &C       = *g->levels[l + 1];
	bool       have_coarse_f = false;
	// direct-store transport: a gather of restricted blocks fills the coarse right-hand side's buffer of ITS parity (two gathers
	// per visit in a W-cycle): called in front of everything that produces the coarse right-hand side
	auto coarseBuf = [&]() {
		if (g->push.on && L.push_blocks && !g->recording) C.f->d = L.cf_buf[L.blk_epoch & 1];
	};
	coarseBuf();
	auto       descend = [&]() -> int {
        int r = materialise();
        if (r) return r;
        if (!have_coarse_f) coarseBuf();
        if (have_coarse_f) { // the fused pre-sweep already left AvgRstr(f - A u) in C.f
            have_coarse_f = false;
        } else if (o->fuse && (L.dim == 3 || L.fuse2d)) {
            if ((r = residRestrict(g, L, u->d, f->d, C.f->d, xfFor(L, u->d)))) return r;
        } else {
            if ((r = launchStencil<MODE_RESID>(g, L, u->d, f->d, L.r->d, 0.0, RestrictDst(), xfFor(L, u->d)))) return r; // prepCoarser: r = f - A u
            if ((r = doRestrict(g, l, L.r->d, C.f->d))) return r;
            if ((r = vecop<VOP_SET>(C.u.get(), nullptr, nullptr, 0.0, 0.0, 0.0))) return r;
        }
        if ((r = visit(g, o, l + 1, C.f.get(), C.u.get(), o->fuse != 0))) return r;
        g->cur_level = l;
        // prepFiner (Cycle.h:74-80). When the very next step is an RB-GS sweep on a level without ghost
        // slots, that sweep reads u + P(coarse u) on the fly instead (same bits, one HBM pass less).
        if (o->fuse && next_sweeps > 0
            && (L.prolong_fusable || L.prolong_fusable_cf)
            && (o->smoother == TE_SMOOTH_RBGS
                || (o->smoother == TE_SMOOTH_PATCH_SOLVE && L.dim == 3 && !g->cfg.has(O_PS_SLOW))
                || (o->smoother == TE_SMOOTH_PATCH_SOLVE && patchSolve2dFusable(g, L)))) {
            pending_prolong = C.u->d;
            return TE_OK;
        }
        L.xf_valid_for = nullptr; // u changes in place
        return doProlong(g, l, C.u->d, u->d);
	};
	// opts.fuse = 2: one pre-smoothing RB-GS sweep from the zero iterate, the residual and its restriction in one
	// pass over f (plus a pass over the face layers). All ranks take the same decision on a level or the ones
	// that do not would wait for ghost faces nobody sends: it rests on facts every rank knows (dimension, options,
	// global patch count) and on fuse2_ok, which the hierarchy builder sets identically on all ranks.
	// (a level takes the fuse = 3 path when ...; the same predicate for the next level decides whether that level can
	// read its right-hand side together with exported ghost terms, see below)
	auto unstoredAt = [&](LevelHost &LL, bool has_coarser) {
		return o->fuse >= 3 && o->pre_sweeps == 1 && o->smoother == TE_SMOOTH_RBGS && LL.fuse2_ok && has_coarser && o->cycle_type == 0
		       && o->post_sweeps >= 1 && (LL.prolong_fusable || (LL.dim == 3 && LL.prolong_fusable_cf && !g->cfg.has(O_NO_FUSE3_CF))) && LL.n >= 4
		       && !g->cfg.has(O_NO_FUSE2) && !g->cfg.has(O_NO_FUSE3);
	};
	if (o->fuse >= 2 && u_zero && o->pre_sweeps == 1 && o->smoother == TE_SMOOTH_RBGS && L.fuse2_ok && !g->cfg.has(O_NO_FUSE2)) {
		u_zero = false;
		// opts.fuse = 3: if exactly this sweep, the descent and a fused post-sweep follow, the iterate in between is
		// never stored: the post-sweep kernel recomputes it from f (bit-identical to fuse = 2; a rank-local choice,
		// the peers see the same exchanges)
		u_unstored = unstoredAt(L, true);
		// ... and if the next level takes the same path, its two kernels are the only readers of its right-hand side: the
		// ghost terms of the restricted residual go to its side array instead of a fix-up pass (bit-identical; rank-local)
		double *fcorr_out = (u_unstored && L.dim == 3 && !L.repl_up && (L.prolong_fusable || !g->cfg.has(O_NO_FCORR_CF)) && C.fcorr.p && C.prolong_fusable
		                     && unstoredAt(C, l + 2 < nl) && !g->cfg.has(O_NO_FCORR))
		                        ? C.fcorr.p
		                        : nullptr; // (the next level uniformly refined; this one may be refined: the gather forms the terms;
		                                   //  not into a replicated level: its x terms would have to travel with the blocks)
		if (fcorr_in && !u_unstored) return te::fail(TE_ESTATE, "te_vcycle: exported ghost terms without a reader");
		// 2D: if the next level runs this very branch (it is visited with a zero iterate and is not the coarsest), ITS pre-sweep
		// kernel adds the ghost terms of the restricted residual to its right-hand side before reading it, and the fix-up launch
		// of this level does not happen (bit-identical: the same additions in the same order; rank-local: every child of a local
		// coarse patch is local, no block travels)
		const bool fold_out = L.dim == 2 && l + 2 < nl && C.fuse2_ok && L.n >= 4 && L.n <= 64 && L.tx_up.empty() && L.n_down == 0 && !L.repl_up
		                      && L.child.p && L.Pc == C.P && !g->cfg.has(O_2D_NO_FOLD);
		// (the kernel variants that form a pending right-hand side exist for the 3D path that does not store the iterate)
		const PendingRhs *fs = (pend && u_unstored && L.dim == 3 && !fcorr_in && L.P > 0 && !g->recording) ? pend : nullptr;
		if (fs)
			pend = nullptr;
		else if ((rc = formRhs()))
			return rc;
		if ((rc = zeroSweepResid(g, L, f->d, L.t->d, C.f->d, L.xfbuf[L.xf_cur ^ 1].p, !u_unstored, fcorr_out, fcorr_in, fs, &fold_in, fold_out)))
			return rc;
		C.f_has_corr = fcorr_out != nullptr;
		if (fold_out) C.fold_pending = Fold2DHost{&L, u_unstored ? nullptr : L.t->d};
		if (u_unstored) {
			L.xf_valid_for = nullptr;
		} else {
			xfProduced(L, L.t->d);
			swapData(u, L.t.get());
		}
		have_coarse_f = true;
	} else if (o->fuse >= 2 && u_zero && o->pre_sweeps == 1 && o->smoother == TE_SMOOTH_PATCH_SOLVE && L.fuse2_ok && L.dim == 3
	           && !g->cfg.has(O_NO_FUSE2)) {
		// block Jacobi from the zero iterate: the residual lives on the face layers only (interfaceResidRestrictN)
		if ((rc = formRhs())) return rc;
		u_zero = false;
		// opts.fuse = 3: ... and so does everything the post-sweep reads of this iterate (its interface terms, k_face_corr3d
		// on u + P e): the pre-sweep stores the six face layers of its result and nothing else (bit-identical; rank-local)
		L.ps_faces_req = o->fuse >= 3 && o->cycle_type == 0 && o->post_sweeps >= 1 && L.n == 32 && L.P_global >= 256 && L.prolong_fusable
		                 && (L.sym_ok || L.n_pure == L.P) && L.f6buf.p && !g->cfg.has(O_PS_SLOW) && !g->cfg.has(O_PS_MODE)
		                 && !g->cfg.has(O_NO_PS_FACES);
		if ((rc = smoothOnce(g, l, f, u, TE_SMOOTH_PATCH_SOLVE, o->omega, true))) return rc;
		if ((rc = interfaceResidRestrict(g, L, u->d, xfFor(L, u->d), C.f->d, C.f->n))) return rc;
		have_coarse_f = true;
	} else if (o->fuse >= 2 && u_zero && o->pre_sweeps == 1 && o->smoother == TE_SMOOTH_PATCH_SOLVE && ps2dResidFusable(g, L) && !fold_in.fine) {
		// the same in 2D (64^2 patches on the matrix cores): the residual of the exact patch solves lives on the patch edges
		if ((rc = formRhs())) return rc;
		u_zero = false;
		if ((rc = smoothOnce(g, l, f, u, TE_SMOOTH_PATCH_SOLVE, o->omega, true))) return rc;
		if ((rc = interfaceResidRestrict2d(g, L, u->d, C.f->d, C.f->n))) return rc;
		have_coarse_f = true;
	} else if (fcorr_in || fold_in.fine) {
		return te::fail(TE_ESTATE, "te_vcycle: exported ghost terms without a reader");
	} else if ((rc = formRhs()) || (rc = smooth(o->pre_sweeps, false))) {
		return rc;
	}
	next_sweeps = (o->cycle_type == 1) ? o->mid_sweeps : o->post_sweeps;
	if ((rc = descend())) return rc;
	if (o->cycle_type == 1) {
		if ((rc = smooth(o->mid_sweeps, false))) return rc;
		next_sweeps = o->post_sweeps;
		if ((rc = descend())) return rc;
	}
	return smooth(o->post_sweeps, false, true);
}

// Dry run of one te_vcycle with `o` on zero vectors in which every exchange is recorded instead of performed; the
// per-pair summaries (how many messages, how many doubles, a hash of the (tag, level, count) sequence) are summed over
// the ranks -- each entry has one contributor, the sum is exact -- and every rank checks that what r sends to q is
// what q expects from r, in the same order. All ranks see the same matrix, so all of them fail, or none.
static uint64_t mix64(uint64_t h, uint64_t v)
{
	h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
	h *= 0xBF58476D1CE4E5B9ull;
	return h ^ (h >> 31);
}

static int verifySchedule(te_gmg *g, const te_cycle_opts *o)
{
	const int R = g->nranks;
	if (R < 2) return TE_OK;
	if (!g->rccl.comm && !g->allreduce)
		return te::fail(TE_ESTATE, "te_gmg_verify_schedule: needs te_gmg_use_rccl or te_gmg_set_allreduce");
	int     rc;
	te_vec *f = nullptr, *u = nullptr;
	if ((rc = newVec(g, 0, &f))) return rc;
	if ((rc = newVec(g, 0, &u))) {
		te_vec_destroy(f);
		return rc;
	}
	const bool prof = g->profiling;
	g->profiling    = false;
	g->recording    = true;
	g->record.clear();
	for (auto &L : g->levels) L->xf_valid_for = nullptr;
	g->in_cycle = !g->cfg.has(O_NO_XF);
	rc          = visit(g, o, 0, f, u, o->fuse != 0);
	g->in_cycle = false;
	for (auto &L : g->levels) L->xf_valid_for = nullptr;
	g->recording = false;
	g->profiling = prof;
	(void) hipStreamSynchronize(g->stream);
	te_vec_destroy(f);
	te_vec_destroy(u);
	// The reductions below are collective: a rank whose dry run failed still takes part (its peers would otherwise wait
	// in them until the watchdog fires) and reports the failure through one more summed word, so that all ranks fail together.
	int               local_rc  = rc;
	const std::string local_msg = rc ? std::string(te_last_error()) : std::string();
	// [dir 0 = sent by row to column, 1 = expected by column from row][row][col][count, doubles, hash lo, hash hi] + [failed ranks, 0, 0, 0]
	std::vector<double>   m((size_t) 2 * R * R * 4 + 4, 0.0);
	std::vector<uint64_t> hs((size_t) R, 0), hr((size_t) R, 0);
	auto at = [&](int dir, int from, int to, int k) -> double & { return m[(((size_t) dir * R + from) * R + to) * 4 + k]; };
	for (auto &e : g->record) {
		if (local_rc) break;
		if (e.peer < 0 || e.peer >= R) {
			local_rc = te::fail(TE_ESTATE, "te_gmg_verify_schedule: peer out of range");
			break;
		}
		if (e.send_cnt > 0) {
			at(0, g->rank, e.peer, 0) += 1;
			at(0, g->rank, e.peer, 1) += (double) e.send_cnt;
			hs[e.peer] = mix64(mix64(mix64(hs[e.peer], (uint64_t) e.tag), (uint64_t) e.level), (uint64_t) e.send_cnt);
		}
		if (e.recv_cnt > 0) {
			at(1, e.peer, g->rank, 0) += 1;
			at(1, e.peer, g->rank, 1) += (double) e.recv_cnt;
			hr[e.peer] = mix64(mix64(mix64(hr[e.peer], (uint64_t) e.tag), (uint64_t) e.level), (uint64_t) e.recv_cnt);
		}
	}
	for (int q = 0; q < R; q++) { // 2 x 24 bits of each hash: exact in a double
		at(0, g->rank, q, 2) = (double) (hs[q] & 0xFFFFFF), at(0, g->rank, q, 3) = (double) ((hs[q] >> 24) & 0xFFFFFF);
		at(1, q, g->rank, 2) = (double) (hr[q] & 0xFFFFFF), at(1, q, g->rank, 3) = (double) ((hr[q] >> 24) & 0xFFFFFF);
	}
	g->record.clear();
	m[m.size() - 4] = local_rc ? 1.0 : 0.0;
	// sum over ranks, four doubles at a time through the same path as the solver's scalar reductions
	for (size_t i = 0; i < m.size(); i += 4) {
		HIPCHK(hipMemcpyAsync(g->result.p, &m[i], 4 * sizeof(double), hipMemcpyHostToDevice, g->stream));
		if ((rc = finishReduce(g, 4, 0, true))) return rc;
		for (int k = 0; k < 4; k++) m[i + k] = g->result_host[k];
	}
	if (local_rc) return te::fail(local_rc, local_msg.empty() ? std::string(te_last_error()) : local_msg);
	if (m[m.size() - 4] > 0.0)
		return te::fail(TE_ESTATE, "te_gmg_verify_schedule: the dry run of the cycle failed on " + std::to_string((int) m[m.size() - 4]) + " other rank(s)");
	for (int r = 0; r < R; r++)
		for (int q = 0; q < R; q++)
			for (int k = 0; k < 4; k++)
				if (at(0, r, q, k) != at(1, r, q, k)) {
					char buf[320];
					snprintf(buf, sizeof buf,
					         "te_gmg_verify_schedule: rank %d sends rank %d %.0f messages / %.0f doubles per cycle but rank %d "
					         "expects %.0f / %.0f (or in another order): the ranks would issue different exchange sequences "
					         "(different cycle options or hierarchies?)",
					         r, q, at(0, r, q, 0), at(0, r, q, 1), q, at(1, r, q, 0), at(1, r, q, 1));
					return te::fail(TE_ESTATE, buf);
				}
	return TE_OK;
}

// Every rank must have built the same hierarchy placement (te_hier_build's agglomerate / agglomerate_max / replicate -- ranks
// started with different environments would not): the maximum and the minimum of each number over the ranks agree, or
// TE_ESTATE on all ranks, by name. Collective; runs once, before the first cycle of a sharded solver, also under TE_NO_VERIFY.
static int checkPlacement(te_gmg *g)
{
	if (g->placement_checked || g->nranks < 2 || (!g->rccl.comm && !g->allreduce)) return TE_OK;
	double hi[4], lo[4];
	int    rc;
	for (int pass = 0; pass < 2; pass++) {
		double v[4];
		for (int k = 0; k < 4; k++) v[k] = pass ? -g->placement[k] : g->placement[k];
		HIPCHK(hipMemcpyAsync(g->result.p, v, sizeof v, hipMemcpyHostToDevice, g->stream));
		if ((rc = finishReduce(g, 4, 1, true))) return rc;
		for (int k = 0; k < 4; k++) (pass ? lo : hi)[k] = pass ? -g->result_host[k] : g->result_host[k];
	}
	static const char *what[4] = {"agglomerate (TE_AGGLOMERATE)", "agglomerate_max (TE_AGGLOMERATE_MAX)", "replicate (TE_REPLICATE)", "number of levels"};
	for (int k = 0; k < 4; k++)
		if (hi[k] != lo[k]) {
			char buf[256];
			snprintf(buf, sizeof buf, "the ranks built different hierarchies: %s is %g on this rank (%d), between %g and %g over the ranks", what[k],
			         g->placement[k], g->rank, lo[k], hi[k]);
			return te::fail(TE_ESTATE, buf);
		}
	g->placement_checked = true;
	return TE_OK;
}

static uint64_t optsKey(const te_cycle_opts *o)
{
	uint64_t h = 0;
	for (int32_t v : {o->pre_sweeps, o->post_sweeps, o->coarse_sweeps, o->mid_sweeps, o->cycle_type, o->smoother, o->exact_coarse, o->fuse})
		h = mix64(h, (uint64_t) (uint32_t) v);
	return h;
}

// te_vcycle; `pending`: f is still to be formed (te_bicgstab; consumed by level 0's first reader, visit())
int vcycleWith(te_gmg *g, const te_cycle_opts *o, const te_vec *f, te_vec *u, const PendingRhs *pending)
{
	{
	int rc;
	if (!o) return te::fail(TE_EINVAL, "te_vcycle: null options");
	if ((rc = checkLevelVec(g, 0, f, "te_vcycle")) || (rc = checkLevelVec(g, 0, u, "te_vcycle"))) return rc;
	// several ranks: the first cycle with a new set of options checks that all ranks will issue matching exchange
	// sequences (a mismatch would otherwise be a silent hang inside RCCL); TE_NO_VERIFY skips it
	if (!g->recording && (rc = checkPlacement(g))) return rc;
	if (g->nranks > 1 && !g->recording && (g->rccl.comm || g->allreduce) && !g->verified_opts.count(optsKey(o))
	    && !g->cfg.has(O_NO_VERIFY)) {
		if ((rc = verifySchedule(g, o))) return rc;
		g->verified_opts.insert(optsKey(o));
	}
	if (!o->fuse && (rc = te_vec_set(u, 0.0))) return rc; // Cycle.h:118
	for (auto &L : g->levels) L->xf_valid_for = nullptr, L->ps_faces = L->ps_faces_req = false;
	g->in_cycle    = !g->cfg.has(O_NO_XF);
	g->pending_rhs = pending;
	rc             = visit(g, o, 0, f, u, o->fuse != 0);
	g->pending_rhs = nullptr;
	g->in_cycle    = false;
	const double *keep = (g->keep_final_xf && rc == TE_OK) ? g->levels[0]->xf_valid_for : nullptr; // (describes u->d, or nothing)
	for (auto &L : g->levels) L->xf_valid_for = nullptr, L->ps_faces = L->ps_faces_req = false;
	if (keep == u->d) g->levels[0]->xf_valid_for = keep;
	return rc;
	}
}

} // namespace tei

extern "C" {
int te_apply(te_gmg *g, int level, const te_vec *u, te_vec *f)
{
	return guarded([&]() -> int {
		int rc;
		if ((rc = checkLevelVec(g, level, u, "te_apply")) || (rc = checkLevelVec(g, level, f, "te_apply"))) return rc;
		if (u == f) return te::fail(TE_EINVAL, "te_apply: in-place apply is not supported");
		return launchStencil<MODE_APPLY>(g, *g->levels[level], u->d, nullptr, f->d, 0.0);
	});
}

int te_residual(te_gmg *g, int level, const te_vec *u, const te_vec *f, te_vec *r)
{
	return guarded([&]() -> int {
		int rc;
		if ((rc = checkLevelVec(g, level, u, "te_residual")) || (rc = checkLevelVec(g, level, f, "te_residual"))
		    || (rc = checkLevelVec(g, level, r, "te_residual")))
			return rc;
		if (u == r) return te::fail(TE_EINVAL, "te_residual: r must not alias u");
		return launchStencil<MODE_RESID>(g, *g->levels[level], u->d, f->d, r->d, 0.0);
	});
}

int te_residual_norm_sq(te_gmg *g, int level, const te_vec *u, const te_vec *f, te_vec *r, double *norm_sq)
{
	return guarded([&]() -> int {
		int rc;
		if (!norm_sq) return te::fail(TE_EINVAL, "te_residual_norm_sq: null result");
		if ((rc = checkLevelVec(g, level, u, "te_residual_norm_sq")) || (rc = checkLevelVec(g, level, f, "te_residual_norm_sq"))
		    || (rc = checkLevelVec(g, level, r, "te_residual_norm_sq")))
			return rc;
		if (u == r) return te::fail(TE_EINVAL, "te_residual_norm_sq: r must not alias u");
		LevelHost &L = *g->levels[level];
		if (L.dim == 2) { // (2D: the same kernel forms one partial sum per workgroup; a fixed grid, so a fixed order)
			*norm_sq = 0.0;
			if (L.P == 0) return TE_OK;
			int blocks = 0;
			if ((rc = residualSumsq2d(g, L, u->d, f->d, r->d, &blocks))) return rc;
			hipLaunchKernelGGL(k_reduce_final2, dim3(1), dim3(256), 0, g->stream, blocks, g->partial.p, g->result.p);
			if ((rc = finishReduce(g, 1, 0, false))) return rc;
			*norm_sq = g->result_host[0];
			return TE_OK;
		}
		int items = 0;
		if ((rc = launchStencil<MODE_RESID>(g, L, u->d, f->d, r->d, 0.0, RestrictDst(), nullptr, RED_OUT_OUT, nullptr, &items))) return rc;
		if (items > 0)
			hipLaunchKernelGGL(k_reduce_final2, dim3(1), dim3(256), 0, g->stream, items, g->partial.p, g->result.p);
		else
			HIPCHK(hipMemsetAsync(g->result.p, 0, 2 * sizeof(double), g->stream));
		if ((rc = finishReduce(g, 1, 0, false))) return rc;
		*norm_sq = g->result_host[0];
		return TE_OK;
	});
}

int te_smooth(te_gmg *g, int level, const te_vec *f, te_vec *u, int smoother, double omega, int sweeps)
{
	return guarded([&]() -> int {
		int rc;
		if ((rc = checkLevelVec(g, level, u, "te_smooth")) || (rc = checkLevelVec(g, level, f, "te_smooth"))) return rc;
		for (int i = 0; i < sweeps; i++)
			if ((rc = smoothOnce(g, level, f, u, smoother, omega))) return rc;
		return TE_OK;
	});
}

int te_gmg_set_patch_bcgs(te_gmg *g, double tol, int max_it)
{
	return guarded([&]() -> int {
		if (!g || !(tol >= 0.0) || max_it < 0) return te::fail(TE_EINVAL, "te_gmg_set_patch_bcgs: tol >= 0 and max_it >= 0");
		g->bcgs_tol    = tol;
		g->bcgs_max_it = max_it;
		return TE_OK;
	});
}

int te_gmg_patch_bcgs_iterations(te_gmg *g, int level, int32_t *its)
{
	return guarded([&]() -> int {
		if (!g || !its || level < 0 || level >= (int) g->levels.size()) return te::fail(TE_EINVAL, "te_gmg_patch_bcgs_iterations: bad argument");
		LevelHost &L = *g->levels[level];
		if (L.P == 0) return TE_OK;
		if (!L.bcgs_its.p) return te::fail(TE_ESTATE, "te_gmg_patch_bcgs_iterations: no TE_SMOOTH_PATCH_BCGS sweep has run on this level");
		HIPCHK(hipStreamSynchronize(g->stream));
		HIPCHK(hipMemcpy(its, L.bcgs_its.p, sizeof(int32_t) * (size_t) L.P, hipMemcpyDeviceToHost));
		return TE_OK;
	});
}

int te_restrict(te_gmg *g, int fine_level, const te_vec *fine, te_vec *coarse)
{
	return guarded([&]() -> int {
		int rc;
		if ((rc = checkLevelVec(g, fine_level, fine, "te_restrict"))
		    || (rc = checkLevelVec(g, fine_level + 1, coarse, "te_restrict")))
			return rc;
		return doRestrict(g, fine_level, fine->d, coarse->d);
	});
}

int te_prolong_add(te_gmg *g, int fine_level, const te_vec *coarse, te_vec *fine)
{
	return guarded([&]() -> int {
		int rc;
		if ((rc = checkLevelVec(g, fine_level, fine, "te_prolong_add"))
		    || (rc = checkLevelVec(g, fine_level + 1, coarse, "te_prolong_add")))
			return rc;
		return doProlong(g, fine_level, coarse->d, fine->d);
	});
}

int te_gmg_verify_schedule(te_gmg *g, const te_cycle_opts *o)
{
	return guarded([&]() -> int {
		if (!g || !o) return te::fail(TE_EINVAL, "te_gmg_verify_schedule: null argument");
		int rc = verifySchedule(g, o);
		if (rc == TE_OK) g->verified_opts.insert(optsKey(o));
		return rc;
	});
}

int te_vcycle(te_gmg *g, const te_cycle_opts *o, const te_vec *f, te_vec *u)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_vcycle: null solver");
		WatchdogBatch batch(g);
		return vcycleWith(g, o, f, u, nullptr);
	});
}

// How the sweeps of the sharded levels meet their face exchanges (LevelHost::overlap_mode) is a matter of microseconds that
// only the machine the job runs on can settle: kernel launches of a rank's share are short, the wire and the peers' skew are
// not in any single-GPU measurement. Candidates -- every one gives bit-identical results, only the order of independent work on
// the two streams differs -- are timed here on the live communicator: `reps` cycles each behind two warm-up cycles, the
// maximum over the ranks (one reduction per candidate: all ranks see the same numbers and choose the same), the fastest kept;
// the serial form wins ties within 2 %. Collective. One rank: nothing to choose. *best_ms (may be NULL): the chosen form's time
// per cycle; report (may be NULL): one line naming the candidates' times and the choice.
int te_gmg_autotune(te_gmg *g, const te_cycle_opts *o, int reps, double *best_ms, char *report, int report_len)
{
	return guarded([&]() -> int {
		if (!g || !o || reps < 1) return te::fail(TE_EINVAL, "te_gmg_autotune: bad argument");
		auto say = [&](const std::string &s2) {
			g->autotune_report = s2;
			if (report && report_len > 0) {
				strncpy(report, s2.c_str(), (size_t) report_len - 1);
				report[report_len - 1] = 0;
			}
		};
		int     rc;
		te_vec *f = nullptr, *u = nullptr;
		if ((rc = newVec(g, 0, &f))) return rc;
		if ((rc = newVec(g, 0, &u))) {
			te_vec_destroy(f);
			return rc;
		}
		struct Free {
			te_vec *a, *b;
			~Free()
			{
				te_vec_destroy(a);
				te_vec_destroy(b);
			}
		} fr{f, u};
		if (f->n > 0 && (rc = te_init_problem(g, 0, TE_PROBLEM_RANDOM, 0, f, nullptr))) return rc;
		hipEvent_t ea, eb;
		HIPCHK(hipEventCreate(&ea));
		HIPCHK(hipEventCreate(&eb));
		struct Ev {
			hipEvent_t a, b;
			~Ev()
			{
				(void) hipEventDestroy(a);
				(void) hipEventDestroy(b);
			}
		} evs{ea, eb};
		const bool prof = g->profiling;
		g->profiling    = false;
		auto timeIt = [&](double *ms_out) -> int { // max over the ranks of this rank's time per cycle
			int r2;
			for (int i = 0; i < 2; i++)
				if ((r2 = vcycleWith(g, o, f, u, nullptr))) return r2;
			HIPCHK(hipEventRecord(ea, g->stream));
			for (int i = 0; i < reps; i++)
				if ((r2 = vcycleWith(g, o, f, u, nullptr))) return r2;
			HIPCHK(hipEventRecord(eb, g->stream));
			HIPCHK(hipEventSynchronize(eb));
			float ms = 0;
			HIPCHK(hipEventElapsedTime(&ms, ea, eb));
			double v = (double) ms / reps;
			if (g->nranks > 1) {
				HIPCHK(hipMemcpyAsync(g->result.p, &v, sizeof v, hipMemcpyHostToDevice, g->stream));
				if ((r2 = finishReduce(g, 1, 1, true))) return r2;
				v = g->result_host[0];
			}
			*ms_out = v;
			return TE_OK;
		};
		WatchdogBatch batch(g);
		const int nl = (int) g->levels.size();
		// sharded levels = the leading levels that are not gathered (a global fact: the hierarchy's placement)
		int nsh = 0;
		while (nsh < nl && g->levels[nsh]->P_global >= g->nranks && !g->levels[nsh]->gathered) nsh++;
		struct Cand {
			int         mode, depth;
			const char *name;
		};
		// ---- transport first (when te_gmg_use_push has prepared the direct-store exchanges): RCCL groups / the host callback
		// against direct stores, both with everything in line. The direct form must also PROVE itself on this machine: after a
		// cycle on another right-hand side (so that stale ghost data would show), its result on f must equal the other transport's
		// bit for bit on every rank, and no wait may have given up. Otherwise it is switched off, on all ranks alike.
		std::string tnote;
		if (g->push.ready && g->nranks > 1) {
			for (int l = 0; l < nl; l++) g->levels[l]->overlap_mode = 0;
			te_vec *ref = nullptr, *f2 = nullptr;
			if ((rc = newVec(g, 0, &ref))) return rc;
			if ((rc = newVec(g, 0, &f2))) {
				te_vec_destroy(ref);
				return rc;
			}
			Free   fr2{ref, f2};
			double t_other = 0, t_push = 0, bad = 0;
			auto   fail    = [&](int r2) {
                g->push.fatal.store(true);
                g->push.trial = false;
                g->profiling  = prof;
                return r2;
			};
			g->push.fatal.store(false);
			g->push.trial = true; // (the waits of the trial give up after seconds, not after the production budget)
			if ((rc = te_gmg_use_push(g, 0)) || (rc = timeIt(&t_other)) || (rc = vcycleWith(g, o, f, ref, nullptr))) return fail(rc);
			if ((rc = te_vec_copy(f2, f)) || (rc = te_vec_scale(f2, -0.625))) return fail(rc);
			if ((rc = te_gmg_use_push(g, 1))) return fail(rc);
			for (int trial = 0; trial < 3 && bad == 0.0; trial++) { // (three times: a race does not show every time)
				if ((rc = vcycleWith(g, o, f2, u, nullptr)) || (rc = vcycleWith(g, o, f, u, nullptr))) return fail(rc);
				if ((rc = te_vec_add_scaled(u, -1.0, ref))) return fail(rc);
				double dmax = 0;
				if (u->n > 0 && (rc = reduce<RED_MAXABS>(u, nullptr, &dmax))) return fail(rc);
				HIPCHK(hipStreamSynchronize(g->stream));
				if (dmax != 0.0) bad = 9.0; // (a result that differs; otherwise the error word's code, pushkernels.hpp PushErr)
				if (te_gmg_push_failed(g)) bad = (double) te_gmg_push_failed(g);
			}
			HIPCHK(hipMemcpyAsync(g->result.p, &bad, sizeof bad, hipMemcpyHostToDevice, g->stream));
			if ((rc = finishReduce(g, 1, 1, true))) return fail(rc);
			bad = g->result_host[0];
			if (bad == 0.0 && (rc = timeIt(&t_push))) return fail(rc);
			char buf[160];
			if (bad != 0.0) {
				(void) te_gmg_use_push(g, 0);
				g->push.ready = false; // not usable on this machine: never again for this solver
				g->push.rejected = true;
				// (its error word is cleared: the watchdog must not end a process that has gone back to the other transport)
				HIPCHK(hipStreamSynchronize(g->stream));
				HIPCHK(hipMemset(g->push.err, 0, sizeof(int)));
				*g->push.err_host = 0;
				static const char *why[] = {"", "a wait gave up", "a peer's flag was two exchanges ahead", "a peer was behind when its buffer was overwritten",
				                            "epochs out of sequence"};
				const int code = (int) bad;
				snprintf(buf, sizeof buf, "transport: direct-store REJECTED (%s) -> %s; ", code >= 1 && code <= 4 ? why[code] : "result differs from the other transport's",
				         g->rccl.comm ? "rccl" : "callback");
			} else {
				const bool take = t_push < 0.98 * t_other;
				if (!take) (void) te_gmg_use_push(g, 0);
				snprintf(buf, sizeof buf, "transport: %s=%.1fus direct-store=%.1fus (results identical) -> %s; ", g->rccl.comm ? "rccl" : "callback",
				         t_other * 1e3, t_push * 1e3, take ? "direct-store" : (g->rccl.comm ? "rccl" : "callback"));
			}
			tnote = buf;
			g->push.fatal.store(true);
			g->push.trial = false;
		}
		std::vector<Cand> cands = {{0, 0, "serial"}};
		if (g->nranks > 1 && g->overlap) {
			cands.push_back({1, 1, "exchange-under-interior/level0"});
			cands.push_back({2, 1, "interior-on-2nd-stream/level0"});
			if (nsh >= 2) {
				cands.push_back({1, 2, "exchange-under-interior/levels0-1"});
				cands.push_back({2, 2, "interior-on-2nd-stream/levels0-1"});
			}
		}
		std::vector<double> t(cands.size(), 0.0);
		auto apply = [&](const Cand &c) {
			for (int l = 0; l < nl; l++) g->levels[l]->overlap_mode = (l < c.depth) ? c.mode : 0;
		};
		for (size_t i = 0; i < cands.size(); i++) {
			apply(cands[i]);
			if ((rc = timeIt(&t[i]))) {
				g->profiling = prof;
				return rc;
			}
		}
		size_t best = 0;
		for (size_t i = 1; i < cands.size(); i++)
			if (t[i] < t[best] && t[i] < 0.98 * t[0]) best = i;
		apply(cands[best]);
		g->profiling = prof;
		std::string s2 = tnote + "overlap:";
		char        buf[96];
		for (size_t i = 0; i < cands.size(); i++) {
			snprintf(buf, sizeof buf, " %s=%.1fus", cands[i].name, t[i] * 1e3);
			s2 += buf;
		}
		s2 += std::string(" -> ") + cands[best].name;
		say(s2);
		if (best_ms) *best_ms = t[best];
		return TE_OK;
	});
}

// StarPatchOp<D>::apply (StarPatchOp.h:204-319; twins SevenPtPatchOperator.cpp:247-409, FivePtPatchOperator.h:172-261):
// f = A_patch u, every face with a neighbour closed as homogeneous Dirichlet (ghost = -m) -- the operator the exact
// patch solves invert (PatchSolvers/BiCGStabSolver.h:82-85 applies it). Same kernel as te_apply with patch-local face kinds.
int te_patch_apply(te_gmg *g, int level, const te_vec *u, te_vec *f)
{
	return guarded([&]() -> int {
		int rc;
		if ((rc = checkLevelVec(g, level, u, "te_patch_apply")) || (rc = checkLevelVec(g, level, f, "te_patch_apply"))) return rc;
		if (u == f) return te::fail(TE_EINVAL, "te_patch_apply: in-place apply is not supported");
		LevelHost &L  = *g->levels[level];
		L.patch_local = true;
		rc            = launchStencil<MODE_APPLY>(g, L, u->d, nullptr, f->d, 0.0);
		L.patch_local = false;
		return rc;
	});
}

} // extern "C"
