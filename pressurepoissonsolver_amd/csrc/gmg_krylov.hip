// Vector<D> BLAS-1 entry points (Vector.h:190-321) and te_bicgstab (BiCGStab.h:45-106) (see gmg_internal.hpp).
#include "gmg_internal.hpp"

namespace tei
{
} // namespace tei

extern "C" {
int te_vec_set(te_vec *v, double a) { return guarded([&]() -> int { return vecop<VOP_SET>(v, nullptr, nullptr, a, 0, 0); }); }
int te_vec_scale(te_vec *v, double a) { return guarded([&]() -> int { return vecop<VOP_SCALE>(v, nullptr, nullptr, a, 0, 0); }); }
int te_vec_shift(te_vec *v, double d) { return guarded([&]() -> int { return vecop<VOP_SHIFT>(v, nullptr, nullptr, d, 0, 0); }); }
int te_vec_copy(te_vec *v, const te_vec *b) { return guarded([&]() -> int { return vecop<VOP_COPY>(v, b, nullptr, 0, 0, 0); }); }
int te_vec_add(te_vec *v, const te_vec *b) { return guarded([&]() -> int { return vecop<VOP_ADD>(v, b, nullptr, 0, 0, 0); }); }
int te_vec_add_scaled(te_vec *v, double a, const te_vec *b) { return guarded([&]() -> int { return vecop<VOP_ADD_SCALED>(v, b, nullptr, a, 0, 0); }); }
int te_vec_add_scaled2(te_vec *v, double alpha, const te_vec *a, double beta, const te_vec *b)
{
	return guarded([&]() -> int {
		return vecop<VOP_ADD_SCALED2>(v, a, b, alpha, beta, 0);
	});
}
int te_vec_scale_then_add(te_vec *v, double a, const te_vec *b) { return guarded([&]() -> int { return vecop<VOP_SCALE_THEN_ADD>(v, b, nullptr, a, 0, 0); }); }
int te_vec_scale_then_add_scaled(te_vec *v, double a, double be, const te_vec *b)
{
	return guarded([&]() -> int {
		return vecop<VOP_SCALE_THEN_ADD_SCALED>(v, b, nullptr, a, be, 0);
	});
}
int te_vec_scale_then_add_scaled2(te_vec *v, double a, double be, const te_vec *b, double ga, const te_vec *c)
{
	return guarded([&]() -> int {
		return vecop<VOP_SCALE_THEN_ADD_SCALED2>(v, b, c, a, be, ga);
	});
}
int te_vec_two_norm_sq(const te_vec *v, double *out) { return guarded([&]() -> int { return reduce<RED_SUMSQ>(v, nullptr, out); }); }
int te_vec_inf_norm(const te_vec *v, double *out) { return guarded([&]() -> int { return reduce<RED_MAXABS>(v, nullptr, out); }); }
int te_vec_dot(const te_vec *v, const te_vec *b, double *out) { return guarded([&]() -> int { return reduce<RED_DOT>(v, b, out); }); }

// This rank's part of the vector's checksum (te_hip.h): the caller adds the ranks' parts modulo 2^64.
int te_vec_checksum(const te_vec *v, uint64_t *out)
{
	return guarded([&]() -> int {
		if (!v || !out) return te::fail(TE_EINVAL, "te_vec_checksum: null argument");
		te_gmg *g = v->g;
		*out      = 0;
		if (v->n == 0 || (g->rank != 0 && g->levels[v->level]->replicated)) return TE_OK; // (a level on every rank counts once: rank 0's)
		static_assert(sizeof(unsigned long long) == sizeof(double), "the result word doubles as the checksum's accumulator");
		HIPCHK(hipMemsetAsync(g->result.p, 0, sizeof(double), g->stream));
		{
			Timed t(g, KC_REDUCE, v->n);
			hipLaunchKernelGGL(k_checksum, dim3(gridFor(v->n / 2, 256, g->red_blocks)), dim3(256), 0, g->stream, v->n / 2,
			                   reinterpret_cast<const double2 *>(v->d), reinterpret_cast<unsigned long long *>(g->result.p));
		}
		HIPCHK(hipMemcpyAsync(g->result_host, g->result.p, sizeof(double), hipMemcpyDeviceToHost, g->stream));
		HIPCHK(hipStreamSynchronize(g->stream));
		memcpy(out, g->result_host, sizeof *out);
		return TE_OK;
	});
}

// BiCGStab.h:45-106, statement for statement, on device vectors. Several ranks: every scalar is summed over the
// ranks (Vector.h:294,319) -- ncclAllReduce of the one or two doubles on the solver stream with the native RCCL
// back-end, otherwise the te_gmg_set_allreduce callback -- so all ranks take the same branches.
int te_bicgstab(te_gmg *g, const te_cycle_opts *o, te_vec *x, const te_vec *b, int max_it, double tol,
                int *iterations, double *rel_resid)
{
	return guarded([&]() -> int {
		int rc;
		if ((rc = checkLevelVec(g, 0, x, "te_bicgstab")) || (rc = checkLevelVec(g, 0, b, "te_bicgstab"))) return rc;
		WatchdogBatch batch(g);
		if (g->nranks > 1 && !g->rccl.comm && !g->allreduce)
			return te::fail(TE_ESTATE, "te_bicgstab on a sharded hierarchy needs te_gmg_use_rccl or te_gmg_set_allreduce");
		// the eight work vectors stay with the solver (a driver solves again and again: allocating and freeing 8 GiB at 512^3
		// cost 2.5 ms per solve); released in te_gmg_destroy
		te_vec **w   = g->bicg_work;
		auto    done = [&](int code) {
            g->keep_final_xf                = false;
            g->levels[0]->xf_valid_for = nullptr;
            return code;
		};
		for (int i = 0; i < 8; i++)
			if (!w[i] && (rc = newVec(g, 0, &w[i]))) return rc;
		te_vec *resid = w[0], *ms = w[1], *mp = w[2], *rhat = w[3], *p = w[4], *ap = w[5], *as = w[6], *s = w[7];
		double r0sq, rsq, rho, tmp, tmp2;
#define TE_TRY(x)                \
		if ((rc = (x))) return done(rc)
		// The dot products that follow an operator application (BiCGStab.h:73-74, 85-87) and the norm of the first residual
		// (:57-60) are formed by the stencil kernel itself while its result is in registers (k_stencil3d RED, 3D; fixed
		// summation order per launch geometry): 16 B/site per dot that a separate pass over stored vectors would read.
		// TE_NO_BICG_FUSE: the separate passes (k_reduce / k_bicg_omega), as before round 3.
		// 2D: k_stencil2d has the same sums (RED; te_residual_norm_sq uses them) but the solve does not: measured at 4096^2 (round 5,
		// same box, alternating) the fused form is SLOWER, 9.33 -> 9.60 ms per solve -- the vectors of a 2D problem that size sit in
		// the Infinity Cache, where a separate dot-product pass costs less than the second operand and the block sums cost the
		// stencil kernel.
		const bool   fused = g->dim == 3 && !g->cfg.has(O_NO_BICG_FUSE);
		g->keep_final_xf   = fused && o != nullptr && !g->cfg.has(O_NO_XF) && !g->cfg.has(O_NO_BICG_XF);
		LevelHost   &L0    = *g->levels[0];
		const size_t n2    = x->n / 2;
		const int    fat   = gridFor(n2, 256, 1 << 30), rb = gridFor(n2, 256, g->red_blocks / 2);
		auto         two   = [&](int nparts, double *a, double *b2) -> int { // fixed-order sum of the per-block pairs -> all ranks -> host
	        if (nparts > 0)
	            hipLaunchKernelGGL(k_reduce_final2, dim3(1), dim3(256), 0, g->stream, nparts, g->partial.p, g->result.p);
	        else
	            HIPCHK(hipMemsetAsync(g->result.p, 0, 2 * sizeof(double), g->stream));
	        int r2 = finishReduce(g, 2, 0, true);
	        if (r2) return r2;
	        *a  = g->result_host[0];
	        *b2 = g->result_host[1];
	        return TE_OK;
		};
		// out = A in together with the sums `redmode` asks for (second operand a)
		auto applySums = [&](const te_vec *in, te_vec *outv, int redmode, const te_vec *a, double *s0, double *s1) -> int {
			if (L0.xf_valid_for == outv->d) L0.xf_valid_for = nullptr;
			int items = 0;
			// (in = the result of the cycle just before: its x-face columns came out of the cycle's last sweep, keep_final_xf)
			int r2    = launchStencil<MODE_APPLY>(g, L0, in->d, nullptr, outv->d, 0.0, RestrictDst(), xfFor(L0, in->d), redmode, a->d, &items);
			if (r2) return r2;
			return two(items, s0, s1);
		};
		if (fused) {
			int    items = 0;
			double dummy;
			TE_TRY(launchStencil<MODE_RESID>(g, L0, x->d, b->d, resid->d, 0.0, RestrictDst(), nullptr, RED_OUT_OUT, nullptr, &items));
			TE_TRY(two(items, &r0sq, &dummy));
		} else {
			TE_TRY(te_apply(g, 0, x, resid));
			TE_TRY(te_vec_scale_then_add(resid, -1, b));
			TE_TRY(reduce<RED_SUMSQ>(resid, nullptr, &r0sq, true));
		}
		const double r0_norm = sqrt(r0sq);
		TE_TRY(te_vec_copy(rhat, resid));
		TE_TRY(te_vec_copy(p, resid));
		if (fused)
			rho = r0sq; // rhat == resid at this point: the dot product is the sum of the same squares as the norm above
		else
			TE_TRY(reduce<RED_DOT>(rhat, resid, &rho, true));
		int num_its = 0;
		rsq         = r0sq;
		// with a preconditioner the two vector statements whose results are right-hand sides of cycles (s, p) are left to the
		// cycle's first reader (PendingRhs)
		const bool defer       = fused && o != nullptr;
		PendingRhs pend_p{};
		bool       have_pend_p = false;
		// Loop body = BiCGStab.h:71-104 statement for statement; the vector statements between two operator
		// applications are fused into one kernel each (same expressions per element).
		while (sqrt(rsq) / r0_norm > tol && num_its < max_it) {
			const te_vec *ain = o ? mp : p;
			if (o) TE_TRY(vcycleWith(g, o, p, mp, have_pend_p ? &pend_p : nullptr)); // (p = beta (p - omega ap) + resid of the previous iteration rides along)
			have_pend_p = false;
			if (fused) {
				double dummy;
				TE_TRY(applySums(ain, ap, RED_OUT_A, rhat, &tmp, &dummy));
			} else {
				TE_TRY(te_apply(g, 0, ain, ap));
				TE_TRY(reduce<RED_DOT>(rhat, ap, &tmp, true));
			}
			const double alpha = rho / tmp;
			const te_vec *sin = o ? ms : s;
			if (defer) { // s = resid - alpha ap is formed by the first kernel of the cycle that reads it (or just before it)
				const PendingRhs ps{1, FSrc{resid->d, ap->d, nullptr, s->d, -alpha, 0.0}, n2};
				TE_TRY(vcycleWith(g, o, s, ms, &ps));
			} else {
				if (n2 > 0) {
					Timed t(g, KC_BICG_S, x->n);
					hipLaunchKernelGGL(k_bicg_s, dim3(fat), dim3(256), 0, g->stream, n2, (double2 *) s->d, (const double2 *) resid->d,
					                   (const double2 *) ap->d, -alpha);
				}
				if (o) TE_TRY(te_vcycle(g, o, s, ms));
			}
			tmp = tmp2 = 0.0;
			if (fused) {
				TE_TRY(applySums(sin, as, RED_OUT_A_OUT, s, &tmp, &tmp2));
			} else {
				TE_TRY(te_apply(g, 0, sin, as));
				if (n2 > 0) {
					Timed t(g, KC_REDUCE, x->n);
					hipLaunchKernelGGL(k_bicg_omega, dim3(rb), dim3(256), 0, g->stream, n2, (const double2 *) as->d,
					                   (const double2 *) s->d, g->partial.p);
				}
				if (n2 > 0 || g->nranks > 1) TE_TRY(two(n2 > 0 ? rb : 0, &tmp, &tmp2));
			}
			const double   omega = tmp / tmp2;
			const te_vec *dp = o ? mp : p, *ds = o ? ms : s;
			double         rho_new = 0.0;
			if (n2 > 0) {
				Timed t(g, KC_BICG_UPDATE, x->n);
				hipLaunchKernelGGL(k_bicg_update, dim3(rb), dim3(256), 0, g->stream, n2, (double2 *) x->d, (double2 *) resid->d,
				                   (const double2 *) dp->d, (const double2 *) ds->d, (const double2 *) ap->d,
				                   (const double2 *) as->d, (const double2 *) rhat->d, alpha, omega, g->partial.p);
			}
			if (n2 > 0 || g->nranks > 1) TE_TRY(two(n2 > 0 ? rb : 0, &rho_new, &rsq));
			const double beta = rho_new * alpha / (rho * omega);
			if (defer) { // p's only reader is the next iteration's cycle (ap and resid stay as they are until then)
				pend_p      = PendingRhs{2, FSrc{p->d, ap->d, resid->d, p->d, -omega, beta}, n2};
				have_pend_p = true;
			} else if (n2 > 0) {
				Timed t(g, KC_BICG_P, x->n);
				hipLaunchKernelGGL(k_bicg_p, dim3(fat), dim3(256), 0, g->stream, n2, (double2 *) p->d, (const double2 *) ap->d,
				                   (const double2 *) resid->d, -omega, beta);
			}
			num_its++;
			rho = rho_new;
		}
#undef TE_TRY
		if (iterations) *iterations = num_its;
		if (rel_resid) *rel_resid = sqrt(rsq) / r0_norm;
		return done(TE_OK);
	});
}

} // extern "C"
