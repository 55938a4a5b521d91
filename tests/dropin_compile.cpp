// Compile-only check (build container, where the reference tree is present): the adaptors are real
// subclasses of the reference's plugin interfaces, and the reference's own BiCGStab<3> template
// instantiates against them. Nothing runs here (no GPU in the build container).
#include <Thunderegg/BiCGStab.h>
#include <Thunderegg/GMG/CycleOpts.h>
#include <HipGMG.h>
#include <HipInit.h>

int drive(const te_hier *h)
{
	using namespace tehip;
	std::shared_ptr<Context>            ctx(new Context(h));
	std::shared_ptr<VectorGenerator<3>> vg(new HipVG<3>(ctx, 0));
	std::shared_ptr<Operator<3>>        A(new HipOperator<3>(ctx, 0));
	te_cycle_opts                       o;
	te_cycle_opts_default(&o);
	GMG::CycleOpts ref_opts; // same field names and defaults (GMG/CycleOpts.h:51-79)
	o.pre_sweeps  = ref_opts.pre_sweeps;
	o.post_sweeps = ref_opts.post_sweeps;
	std::shared_ptr<Operator<3>>      M(new HipCycle<3>(ctx, o));
	std::shared_ptr<GMG::Smoother<3>> S(new HipSmoother<3>(ctx, 0));
	std::shared_ptr<GMG::Restrictor<3>>   R(new HipRestrictor<3>(ctx, 0));
	std::shared_ptr<GMG::Interpolator<3>> I(new HipInterpolator<3>(ctx, 0));
	auto u = vg->getNewVector(), f = vg->getNewVector();
	S->smooth(f, u);
	return BiCGStab<3>::solve(vg, A, u, f, M); // apps/3d/steady.cpp:522, unchanged call
}

// the D = 2 instantiations (apps/2d/steady.cpp:322-331, 494, 523, 563-568) and the 2D twins of Init (Init.cpp:246-361)
int drive2d(const te_hier *h)
{
	using namespace tehip;
	std::shared_ptr<Context>              ctx(new Context(h));
	std::shared_ptr<VectorGenerator<2>>   vg(new HipVG<2>(ctx, 0));
	std::shared_ptr<Operator<2>>          A(new HipOperator<2>(ctx, 0));
	te_cycle_opts                         o;
	te_cycle_opts_default(&o);
	std::shared_ptr<Operator<2>>          M(new HipCycle<2>(ctx, o));
	std::shared_ptr<GMG::Smoother<2>>     S(new HipSmoother<2>(ctx, 0));
	std::shared_ptr<GMG::Restrictor<2>>   R(new HipRestrictor<2>(ctx, 0));
	std::shared_ptr<GMG::Interpolator<2>> I(new HipInterpolator<2>(ctx, 0));
	auto u = vg->getNewVector(), f = vg->getNewVector(), e = vg->getNewVector();
	LevelGeometry G(h, 0);
	initDirichlet2d(G, f, e, [](double x, double y) { return x + y; }, [](double x, double y) { return x * y; });
	initNeumann2d(G, f, e, [](double x, double y) { return x + y; }, [](double x, double y) { return x * y; }, [](double, double y) { return y; },
	              [](double x, double) { return x; });
	S->smooth(f, u);
	return BiCGStab<2>::solve(vg, A, u, f, M); // apps/2d/steady.cpp:563-568, unchanged call
}
