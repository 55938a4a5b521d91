// Header-only C++ adaptors that put the MI355X path (libte_hip.so, include/te_hip.h) behind the
// reference's own plugin surface, so GMG::Cycle<D>, BiCGStab<D> and apps/3d/steady.cpp run
// unchanged on device-resident vectors. Compiles against the reference's headers
// (-I$THUNDEREGG_REF/src): every class below derives from a reference interface.
//
//   HipVector<D>       : Vector<D>            (Vector.h:179-322)  all BLAS-1 virtuals overridden
//   HipVG<D>           : VectorGenerator<D>   (Vector.h:323-327)
//   HipOperator<D>     : Operator<D>          (Operators/Operator.h:28-38)   == SchurDomainOp / DomainWrapOp
//   HipSmoother<D>     : GMG::Smoother<D>     (GMG/Smoother.h:39)            == FFTBlockJacobiSmoother (kind 0)
//   HipRestrictor<D>   : GMG::Restrictor<D>   (GMG/Restrictor.h:39-40)       == AvgRstr
//   HipInterpolator<D> : GMG::Interpolator<D> (GMG/Interpolator.h:39-40)     == DrctIntp
//   HipCycle<D>        : Operator<D>          whole GMG::Cycle<D>::apply (GMG/Cycle.h:116-126) in one call
//
// Error convention: the reference throws `int` (`throw 3;`, e.g. GMG/InterLevelComm.h:175,
// SchurHelper.h:129); a non-zero te_* status is rethrown the same way.
// Ownership: everything by std::shared_ptr, as the reference does (GMG/Level.h:84-186 setters).
// Threading: none; one solver stream per te_gmg; not re-entrant (same as the reference).
#ifndef THUNDEREGG_HIP_GMG_H
#define THUNDEREGG_HIP_GMG_H
#include <Thunderegg/GMG/Interpolator.h>
#include <Thunderegg/GMG/Restrictor.h>
#include <Thunderegg/GMG/Smoother.h>
#include <Thunderegg/Operators/Operator.h>
#include <Thunderegg/Vector.h>
#include <map>
#include <memory>
#include <te_hip.h>
#include <vector>

namespace tehip
{
inline void check(int status)
{
	if (status != TE_OK) throw 3; // the reference's error convention
}

/// Owns a te_gmg; shared by every adaptor created from it.
struct Context {
	te_gmg *g = nullptr;
	int     n = 0, dim = 0;
	explicit Context(const te_hier *h, int device = -1)
	{
		check(te_gmg_create(h, device, &g));
		n   = te_hier_n(h);
		dim = te_hier_dim(h);
	}
	~Context() { te_gmg_destroy(g); }
	Context(const Context &) = delete;
	Context &operator=(const Context &) = delete;
};

/// Host side of getLocalData(): ONE mirror per vector, shared by all the views handed out, holding only the patches
/// that are currently viewed. The first view of a patch downloads that patch (te_vec_download_patches: n^D doubles,
/// not the vector); further views of the same patch alias the same host memory, as every VecGetArray of a
/// PetscVector aliases the same storage (PetscVector.h:27-58), so concurrent views cannot lose each other's
/// updates; when the last view of a patch goes away and any of them was writable the patch is uploaded back.
struct HipMirror {
	te_vec *v;
	size_t  cells; // per patch
	struct Slot {
		std::vector<double> host;
		int                 views = 0;
		bool                dirty = false;
	};
	std::map<int, Slot> slots;
	int                 failed = 0; // status of the last failed write-back (a destructor cannot throw)
	HipMirror(te_vec *v_, size_t cells_) : v(v_), cells(cells_) {}
	double *acquire(int patch, bool writable)
	{
		Slot &s = slots[patch];
		if (s.views == 0) {
			s.host.resize(cells);
			check(te_vec_download_patches(v, patch, 1, s.host.data()));
		}
		s.views++;
		s.dirty |= writable;
		return s.host.data();
	}
	void release(int patch)
	{
		auto it = slots.find(patch);
		if (it == slots.end() || --it->second.views > 0) return;
		if (it->second.dirty) {
			int rc = te_vec_upload_patches(v, patch, 1, it->second.host.data());
			if (rc != TE_OK) failed = rc;
		}
		slots.erase(it);
	}
};
/// The manager a LocalData carries: releases its patch when the last copy of the view is gone (the role PetscLDM
/// plays for PetscVector).
class HipLDM : public LocalDataManager
{
	std::shared_ptr<HipMirror> mirror;
	int                        patch;

	public:
	double *data;
	HipLDM(std::shared_ptr<HipMirror> m, int patch_, bool writable) : mirror(m), patch(patch_), data(m->acquire(patch_, writable)) {}
	~HipLDM() { mirror->release(patch); }
};

template <size_t D> class HipVector : public Vector<D>
{
	public:
	std::shared_ptr<Context>   ctx;
	te_vec                    *v = nullptr;
	int                        level;
	std::shared_ptr<HipMirror> mirror;
	HipVector(std::shared_ptr<Context> ctx_, int level_) : ctx(ctx_), level(level_)
	{
		check(te_vec_create(ctx->g, level, &v));
		size_t cells = 1;
		for (size_t i = 0; i < D; i++) cells *= ctx->n;
		this->num_local_patches = (int) (te_vec_size(v) / cells);
		mirror.reset(new HipMirror(v, cells));
	}
	~HipVector()
	{
		mirror->v = nullptr; // (no view can outlive the vector: LocalData is used inside patch loops)
		te_vec_destroy(v);
	}
	/// status of the last failed write-back of a released view, TE_OK if none (a destructor cannot throw)
	int writeBackStatus() const { return mirror->failed; }
	static const te_vec *raw(std::shared_ptr<const Vector<D>> b)
	{
		auto p = std::dynamic_pointer_cast<const HipVector<D>>(b);
		if (!p) throw 3; // same failure mode as SchurHelper.h:129 on a foreign vector type
		return p->v;
	}
	LocalData<D> view(int patch, bool wb) const
	{
		if (patch < 0 || patch >= this->num_local_patches) throw 3;
		std::shared_ptr<HipLDM> ldm(new HipLDM(mirror, patch, wb));
		std::array<int, D>      lengths, strides;
		int                     s = 1;
		for (size_t i = 0; i < D; i++) {
			lengths[i] = ctx->n;
			strides[i] = s;
			s *= ctx->n;
		}
		return LocalData<D>(ldm->data, strides, lengths, ldm);
	}
	LocalData<D>       getLocalData(int p) override { return view(p, true); }
	const LocalData<D> getLocalData(int p) const override { return view(p, false); }
	void set(double a) override { check(te_vec_set(v, a)); }
	void scale(double a) override { check(te_vec_scale(v, a)); }
	void shift(double d) override { check(te_vec_shift(v, d)); }
	void copy(std::shared_ptr<const Vector<D>> b) override { check(te_vec_copy(v, raw(b))); }
	void add(std::shared_ptr<const Vector<D>> b) override { check(te_vec_add(v, raw(b))); }
	void addScaled(double a, std::shared_ptr<const Vector<D>> b) override { check(te_vec_add_scaled(v, a, raw(b))); }
	void addScaled(double alpha, std::shared_ptr<const Vector<D>> a, double beta, std::shared_ptr<const Vector<D>> b) override
	{
		check(te_vec_add_scaled2(v, alpha, raw(a), beta, raw(b)));
	}
	void scaleThenAdd(double a, std::shared_ptr<const Vector<D>> b) override { check(te_vec_scale_then_add(v, a, raw(b))); }
	void scaleThenAddScaled(double a, double be, std::shared_ptr<const Vector<D>> b) override
	{
		check(te_vec_scale_then_add_scaled(v, a, be, raw(b)));
	}
	void scaleThenAddScaled(double a, double be, std::shared_ptr<const Vector<D>> b, double ga,
	                        std::shared_ptr<const Vector<D>> c) override
	{
		check(te_vec_scale_then_add_scaled2(v, a, be, raw(b), ga, raw(c)));
	}
	// local partial results + the reference's own MPI_Allreduce (Vector.h:294,306,319)
	double twoNorm() const override
	{
		double s, gs;
		check(te_vec_two_norm_sq(v, &s));
		MPI_Allreduce(&s, &gs, 1, MPI_DOUBLE, MPI_SUM, this->comm);
		return sqrt(gs);
	}
	double infNorm() const override
	{
		double s, gs;
		check(te_vec_inf_norm(v, &s));
		MPI_Allreduce(&s, &gs, 1, MPI_DOUBLE, MPI_MAX, this->comm);
		return gs;
	}
	double dot(std::shared_ptr<const Vector<D>> b) const override
	{
		double s, gs;
		check(te_vec_dot(v, raw(b), &s));
		MPI_Allreduce(&s, &gs, 1, MPI_DOUBLE, MPI_SUM, this->comm);
		return gs;
	}
};

template <size_t D> class HipVG : public VectorGenerator<D>
{
	std::shared_ptr<Context> ctx;
	int                      level;

	public:
	HipVG(std::shared_ptr<Context> ctx_, int level_) : ctx(ctx_), level(level_) {}
	std::shared_ptr<Vector<D>> getNewVector() override { return std::shared_ptr<Vector<D>>(new HipVector<D>(ctx, level)); }
};

template <size_t D> class HipOperator : public Operator<D>
{
	std::shared_ptr<Context> ctx;
	int                      level;

	public:
	HipOperator(std::shared_ptr<Context> ctx_, int level_) : ctx(ctx_), level(level_) {}
	void apply(std::shared_ptr<const Vector<D>> x, std::shared_ptr<Vector<D>> b) const override
	{
		check(te_apply(ctx->g, level, HipVector<D>::raw(x), const_cast<te_vec *>(HipVector<D>::raw(b))));
	}
};

template <size_t D> class HipSmoother : public GMG::Smoother<D>
{
	std::shared_ptr<Context> ctx;
	int                      level, kind;
	double                   omega;

	public:
	/// kind: TE_SMOOTH_PATCH_SOLVE (the reference's FFTBlockJacobiSmoother), TE_SMOOTH_JACOBI, TE_SMOOTH_RBGS, or (D = 2)
	/// TE_SMOOTH_PATCH_BCGS: the block-Jacobi sweep over BiCGStabSolver<2>(p_operator, ps_tol, ps_max_it) that
	/// apps/2d/steady.cpp:326-327 builds for --patch_solver bcgs; its stopping rule through te_gmg_set_patch_bcgs(ctx->g, ...)
	HipSmoother(std::shared_ptr<Context> ctx_, int level_, int kind_ = TE_SMOOTH_PATCH_SOLVE, double omega_ = 6.0 / 7.0)
	: ctx(ctx_), level(level_), kind(kind_), omega(omega_)
	{
	}
	void smooth(std::shared_ptr<const Vector<D>> f, std::shared_ptr<Vector<D>> u) const override
	{
		check(te_smooth(ctx->g, level, HipVector<D>::raw(f), const_cast<te_vec *>(HipVector<D>::raw(u)), kind, omega, 1));
	}
};

template <size_t D> class HipRestrictor : public GMG::Restrictor<D>
{
	std::shared_ptr<Context> ctx;
	int                      fine_level;

	public:
	HipRestrictor(std::shared_ptr<Context> ctx_, int fine_level_) : ctx(ctx_), fine_level(fine_level_) {}
	void restrict(std::shared_ptr<Vector<D>> coarse, std::shared_ptr<const Vector<D>> fine) const override
	{
		check(te_restrict(ctx->g, fine_level, HipVector<D>::raw(fine), const_cast<te_vec *>(HipVector<D>::raw(coarse))));
	}
};

template <size_t D> class HipInterpolator : public GMG::Interpolator<D>
{
	std::shared_ptr<Context> ctx;
	int                      fine_level;

	public:
	HipInterpolator(std::shared_ptr<Context> ctx_, int fine_level_) : ctx(ctx_), fine_level(fine_level_) {}
	void interpolate(std::shared_ptr<const Vector<D>> coarse, std::shared_ptr<Vector<D>> fine) const override
	{
		check(te_prolong_add(ctx->g, fine_level, HipVector<D>::raw(coarse), const_cast<te_vec *>(HipVector<D>::raw(fine))));
	}
};

/// The whole preconditioner M = GMG cycle in one native call (what CycleFactory3d::getCycle returns).
template <size_t D> class HipCycle : public Operator<D>
{
	std::shared_ptr<Context> ctx;
	te_cycle_opts            opts;

	public:
	HipCycle(std::shared_ptr<Context> ctx_, const te_cycle_opts &o) : ctx(ctx_), opts(o) {}
	void apply(std::shared_ptr<const Vector<D>> f, std::shared_ptr<Vector<D>> u) const override
	{
		check(te_vcycle(ctx->g, &opts, HipVector<D>::raw(f), const_cast<te_vec *>(HipVector<D>::raw(u))));
	}
};
} // namespace tehip
#endif
