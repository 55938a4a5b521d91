// Compile-only check (build container, where the reference tree is present): the adaptors are real
// subclasses of the reference's plugin interfaces, and the reference's own BiCGStab<3> template
// instantiates against them. Nothing runs here (no GPU in the build container).
#include <Thunderegg/BiCGStab.h>
#include <Thunderegg/GMG/CycleOpts.h>
#include <HipGMG.h>

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
