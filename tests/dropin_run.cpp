// TEST PROGRAM (linked against the reference's own sources and libte_hip.so in the build container by
// oracle/build.py: build_dropin -> oracle/_ref/dropin_run; run on the GPU box by tests/test_gpu_dropin.py).
//
// The reference's OWN BiCGStab<D>::solve (src/Thunderegg/BiCGStab.h:45-106, the call of apps/3d/steady.cpp:519-524 and of
// apps/2d/steady.cpp:563-568) runs UNCHANGED over the adaptors of pressurepoissonsolver_amd/thunderegg/HipGMG.h: HipVG,
// HipOperator (A) and HipCycle (M = the GMG V-cycle), for D = 3 and D = 2 (HipVector<2> / HipVG<2> / HipOperator<2> /
// HipCycle<2> / HipSmoother<2> / HipRestrictor<2> / HipInterpolator<2> are instantiated and run here); the right-hand side is
// filled the way Init::initDirichlet / Init::initDirichlet2d do it (apps/shared/Init.cpp:152-245, :304-361) through
// Vector<D>::getLocalData, one patch at a time. Checks:
//   1. the reference's Krylov loop over the adaptors == the library's te_bicgstab on the same right-hand side
//      (iteration count, solution), and the discretisation error against the analytic solution is second order;
//   2. one V(1,1) cycle driven level by level through Operator / Smoother / Restrictor / Interpolator in the order of
//      GMG/Cycle.h:56-90 + VCycle.h:44-62 == te_vcycle with fuse = 0, bit for bit;
//   3. getLocalData: two writable views of one vector alive at once (different patches, and the same patch) lose no
//      update; a read-only view sees device-side changes.
// usage: dropin_run <mesh file | uniform> <divides> <n> <smoother: 0 patch solve | 2 rbgs> [dim: 3 (default) | 2]
#include <Thunderegg/BiCGStab.h>
#include <Thunderegg/GMG/CycleOpts.h>
#include <HipGMG.h>
#include <HipInit.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <string>

using namespace tehip;

static double trigExact(double x, double y, double z) // apps/3d/steady.cpp:258-263
{
	x += .3, y += .3, z += .3;
	return sin(M_PI * x) * cos(2.0 / 3 * M_PI * y) * sin(5.0 / 6 * M_PI * z);
}
static double trigRhs(double x, double y, double z) { return -77.0 / 36 * M_PI * M_PI * trigExact(x, y, z); } // :252-257
// apps/2d/steady.cpp:314-318 (the default problem of steady2d)
static double trigExact2(double x, double y) { return sinl(M_PI * y) * cosl(2 * M_PI * x); }
static double trigRhs2(double x, double y) { return -5 * M_PI * M_PI * sinl(M_PI * y) * cosl(2 * M_PI * x); }
static void fillRhs(const LevelGeometry &G, std::shared_ptr<Vector<3>> f, std::shared_ptr<Vector<3>> exact) { initDirichlet(G, f, exact, trigRhs, trigExact); }
static void fillRhs(const LevelGeometry &G, std::shared_ptr<Vector<2>> f, std::shared_ptr<Vector<2>> exact) { initDirichlet2d(G, f, exact, trigRhs2, trigExact2); }

#define REQUIRE(cond, ...)                     \
	do {                                       \
		if (!(cond)) {                         \
			fprintf(stderr, "DROPIN_FAIL: " __VA_ARGS__); \
			fprintf(stderr, "\n");             \
			return 1;                          \
		}                                      \
	} while (0)

template <size_t D> struct LevelOps { // what GMG::Level<D> holds (GMG/Level.h:84-186)
	std::shared_ptr<VectorGenerator<D>>   vg;
	std::shared_ptr<Operator<D>>          op;
	std::shared_ptr<GMG::Smoother<D>>     smoother;
	std::shared_ptr<GMG::Restrictor<D>>   restrictor;   // to the next coarser level
	std::shared_ptr<GMG::Interpolator<D>> interpolator; // from the next coarser level
};
// GMG/VCycle.h:44-62 with Cycle.h:56-90 prepCoarser / prepFiner, on the reference's abstract interfaces
template <size_t D>
static void visit(const std::vector<LevelOps<D>> &L, size_t l, const GMG::CycleOpts &o, std::shared_ptr<const Vector<D>> f, std::shared_ptr<Vector<D>> u)
{
	typedef std::shared_ptr<Vector<D>> VecP;
	if (l + 1 == L.size()) {
		for (int i = 0; i < o.coarse_sweeps; i++) L[l].smoother->smooth(f, u);
		return;
	}
	for (int i = 0; i < o.pre_sweeps; i++) L[l].smoother->smooth(f, u);
	VecP r = L[l].vg->getNewVector();
	L[l].op->apply(u, r);        // Cycle.h:60
	r->scaleThenAdd(-1, f);      // :61
	VecP new_u = L[l + 1].vg->getNewVector(), new_f = L[l + 1].vg->getNewVector(); // :63-64
	L[l].restrictor->restrict(new_f, r);                                          // :65
	visit<D>(L, l + 1, o, new_f, new_u);
	L[l].interpolator->interpolate(new_u, u); // Cycle.h:74-80
	for (int i = 0; i < o.post_sweeps; i++) L[l].smoother->smooth(f, u);
}

template <size_t D> static int run(const std::string &mesh_name, int div, int n, int smoother)
{
	typedef std::shared_ptr<Vector<D>> VecP;
	te_mesh *mesh = nullptr;
	te_hier *hier = nullptr;
	try {
		check(mesh_name == "uniform" ? te_mesh_unit_root((int) D, &mesh) : te_mesh_read(mesh_name.c_str(), (int) D, &mesh));
		for (int i = 0; i < div; i++) check(te_mesh_refine_leaves(mesh));
		check(te_hier_build(mesh, n, 0, 0, 0.0, 0, 1, &hier));
		std::shared_ptr<Context>            ctx(new Context(hier));
		std::shared_ptr<VectorGenerator<D>> vg(new HipVG<D>(ctx, 0));
		std::shared_ptr<Operator<D>>        A(new HipOperator<D>(ctx, 0));
		GMG::CycleOpts                      copts; // the reference's defaults: V(1,1), coarse 1 (CycleOpts.h:55-79)
		te_cycle_opts                       o;
		te_cycle_opts_default(&o);
		o.pre_sweeps = copts.pre_sweeps, o.post_sweeps = copts.post_sweeps, o.coarse_sweeps = copts.coarse_sweeps;
		o.mid_sweeps = copts.mid_sweeps, o.cycle_type = (copts.cycle_type == "W"), o.smoother = smoother;
		std::shared_ptr<Operator<D>> M(new HipCycle<D>(ctx, o));

		// ---- right-hand side through getLocalData, as Init::initDirichlet fills a PetscVector
		VecP          f = vg->getNewVector(), exact = vg->getNewVector(), u = vg->getNewVector();
		LevelGeometry G(hier, 0);
		fillRhs(G, f, exact);
		REQUIRE(std::dynamic_pointer_cast<HipVector<D>>(f)->writeBackStatus() == TE_OK, "write-back of a view failed");

		// ---- 1. the reference's Krylov solver, unchanged, over the adaptors
		const int its = BiCGStab<D>::solve(vg, A, u, f, M);
		VecP      au = vg->getNewVector();
		A->apply(u, au);
		au->scaleThenAdd(-1, f);
		const double rel_res = au->twoNorm() / f->twoNorm();
		te_vec      *x = nullptr;
		int          its_native = 0;
		double       rr_native  = 0;
		check(te_vec_create(ctx->g, 0, &x));
		check(te_bicgstab(ctx->g, &o, x, HipVector<D>::raw(f), 1000, 1e-12, &its_native, &rr_native));
		// difference of the two solutions, and the error against the analytic solution
		VecP d = vg->getNewVector();
		d->copy(u);
		check(te_vec_add_scaled(const_cast<te_vec *>(HipVector<D>::raw(d)), -1.0, x));
		const double diff = d->twoNorm() / u->twoNorm();
		d->copy(u);
		d->addScaled(-1.0, exact);
		const double err = d->twoNorm() / exact->twoNorm();
		double       hmin = 1e300;
		for (int p = 0; p < G.P; p++) hmin = std::min(hmin, G.lengths[p * D] / n);
		printf("bicgstab: reference loop over adaptors its=%d rel_resid=%.3e | te_bicgstab its=%d rel_resid=%.3e | rel diff=%.3e | error vs exact=%.3e (h=%g)\n",
		       its, rel_res, its_native, rr_native, diff, err, hmin);
		te_vec_destroy(x);
		REQUIRE(its == its_native, "iteration counts differ: %d vs %d", its, its_native);
		REQUIRE(rel_res <= 1e-11, "reference loop did not converge: %.3e", rel_res);
		REQUIRE(diff <= 1e-10, "solutions differ: %.3e", diff);
		// second order; hmax = 2 hmin on a refined tree
		REQUIRE(err <= 16.0 * hmin * hmin * (mesh_name == "uniform" ? 1 : 4), "discretisation error too large: %.3e", err);

		// ---- 2. level-by-level V(1,1) through the four plugin interfaces == te_vcycle(fuse = 0), bit for bit
		const int             nl = te_gmg_num_levels(ctx->g);
		std::vector<LevelOps<D>> L((size_t) nl);
		for (int l = 0; l < nl; l++) {
			L[l].vg.reset(new HipVG<D>(ctx, l));
			L[l].op.reset(new HipOperator<D>(ctx, l));
			L[l].smoother.reset(new HipSmoother<D>(ctx, l, smoother));
			if (l + 1 < nl) {
				L[l].restrictor.reset(new HipRestrictor<D>(ctx, l));
				L[l].interpolator.reset(new HipInterpolator<D>(ctx, l));
			}
		}
		VecP u1 = vg->getNewVector(), u2 = vg->getNewVector();
		u1->set(0); // Cycle.h:118
		visit<D>(L, 0, copts, f, u1);
		te_cycle_opts o0 = o;
		o0.fuse          = 0;
		o0.exact_coarse  = 0; // the per-level adaptors smooth the coarsest level like any other (kind = `smoother`)
		std::shared_ptr<Operator<D>> M0(new HipCycle<D>(ctx, o0));
		M0->apply(f, u2);
		u2->addScaled(-1.0, u1);
		printf("level-by-level cycle vs te_vcycle(fuse=0): max diff = %.3e\n", u2->infNorm());
		REQUIRE(u2->infNorm() == 0.0, "level-by-level cycle differs from te_vcycle");

		// ---- 3. getLocalData semantics (two patches needed: skipped on a one-patch mesh, config C1)
		if (G.P >= 2) {
			std::array<int, D> c0, c1, c2;
			c0.fill(0), c1.fill(0), c2.fill(0);
			c1[0] = 1, c2[0] = 2;
			VecP w = vg->getNewVector();
			w->set(1.0);
			{
				LocalData<D> a = w->getLocalData(0), b = w->getLocalData(1), c = w->getLocalData(0); // c aliases a
				a[c0] = 5.0;
				b[c1] = 7.0;
				c[c2] = 9.0;
				REQUIRE((c[c0] == 5.0), "two views of one patch do not alias");
			}
			w->scale(2.0); // device side
			std::shared_ptr<const Vector<D>> cw = w;
			const LocalData<D>               r0 = cw->getLocalData(0), r1 = cw->getLocalData(1);
			REQUIRE((r0[c0] == 10.0 && r0[c2] == 18.0 && r0[c1] == 2.0), "patch 0 lost an update");
			REQUIRE((r1[c1] == 14.0 && r1[c0] == 2.0), "patch 1 lost an update");
			REQUIRE(std::dynamic_pointer_cast<HipVector<D>>(w)->writeBackStatus() == TE_OK, "write-back failed");
		}
		printf("DROPIN_OK dim=%d its=%d native_its=%d rel_diff=%.3e err=%.3e\n", (int) D, its, its_native, diff, err);
	} catch (int e) {
		fprintf(stderr, "DROPIN_FAIL: reference-style exception %d: %s\n", e, te_last_error());
		return 1;
	}
	te_hier_destroy(hier);
	te_mesh_destroy(mesh);
	return 0;
}

int main(int argc, char **argv)
{
	MPI_Init(&argc, &argv);
	if (argc < 5) {
		fprintf(stderr, "usage: dropin_run <mesh|uniform> <divides> <n> <smoother> [dim]\n");
		return 2;
	}
	const std::string mesh_name = argv[1];
	const int         div = atoi(argv[2]), n = atoi(argv[3]), smoother = atoi(argv[4]), dim = argc > 5 ? atoi(argv[5]) : 3;
	const int         rc = dim == 2 ? run<2>(mesh_name, div, n, smoother) : run<3>(mesh_name, div, n, smoother);
	MPI_Finalize();
	return rc;
}
