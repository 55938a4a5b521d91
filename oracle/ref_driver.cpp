// TEST INFRASTRUCTURE. Thin C driver over the reference's OWN code for the PETSc-free slice of
// the hot path. Nothing here restates reference arithmetic: every number comes out of
// reference classes compiled from $THUNDEREGG_REF (never copied into this repo):
//   StarPatchOp<D>            src/Thunderegg/StarPatchOp.h   (applyWithInterface, apply, addInterfaceToRHS)
//   SevenPtPatchOperator      src/Thunderegg/SevenPtPatchOperator.cpp
//   TriLinInterp / BilinearInterpolator   (interface interpolation)
//   ValVector<D>, Vector<D>   (BLAS-1 virtuals, norms)
//   BiCGStab<D>               src/Thunderegg/BiCGStab.h
//   Tree<D>                   src/Thunderegg/OctTree.h (file reader, refineLeaves)
//   PatchInfo<D>, SchurInfo<D>  (metadata the operators consume)
// The glue below only builds PatchInfo/SchurInfo objects from flat tables and moves arrays.
// Built by oracle/Makefile.ref into oracle/_ref/libte_ref.so; used by oracle/gen_golden.py to
// produce tests/golden/ref_*.npz and by the CPU tests (when present) to validate te_oracle.
#include <Thunderegg/BiCGStab.h>
#include <Thunderegg/BilinearInterpolator.h>
#include <Thunderegg/OctTree.h>
#include <Thunderegg/SevenPtPatchOperator.h>
#include <Thunderegg/StarPatchOp.h>
#include <Thunderegg/TriLinInterp.h>
#include <Thunderegg/ValVector.h>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

extern "C" {
typedef struct {
	int32_t        dim, n, P;
	const int32_t *id;
	const double  *h;
	const int32_t *nbr_kind, *nbr, *nbr_orth, *neumann, *parent, *orth_on_parent;
} ref_level;
}

namespace
{
template <size_t D> struct Built {
	std::vector<std::shared_ptr<PatchInfo<D>>> pinfos;
	std::vector<SchurInfo<D>>                  sinfos;
	int                                        num_ifaces = 0;
};

template <size_t D> Built<D> build(const ref_level *L)
{
	Built<D>                                     B;
	constexpr int                                NS = 2 * D, NQ = 1 << (D - 1);
	std::map<int, std::shared_ptr<PatchInfo<D>>> by_id;
	for (int p = 0; p < L->P; p++) {
		std::shared_ptr<PatchInfo<D>> pi(new PatchInfo<D>());
		pi->id          = L->id[p];
		pi->local_index = p;
		pi->ns.fill(L->n);
		for (size_t a = 0; a < D; a++) pi->spacings[a] = L->h[p * D + a];
		for (int s = 0; s < NS; s++) {
			int f = p * NS + s, kind = L->nbr_kind[f];
			if (kind == 1) {
				pi->nbr_info[s].reset(new NormalNbrInfo<D>(L->id[L->nbr[f * 4]]));
			} else if (kind == 2) {
				pi->nbr_info[s].reset(new CoarseNbrInfo<D>(L->id[L->nbr[f * 4]], Orthant<D - 1>(L->nbr_orth[f])));
			} else if (kind == 3) {
				std::array<int, NQ> ids;
				for (int q = 0; q < NQ; q++) ids[q] = L->id[L->nbr[f * 4 + q]];
				pi->nbr_info[s].reset(new FineNbrInfo<D>(ids));
			}
			pi->neumann[s] = (L->neumann[p] >> s) & 1;
		}
		by_id[pi->id] = pi;
		B.pinfos.push_back(pi);
	}
	for (auto &pi : B.pinfos) pi->setPtrs(by_id);
	for (auto &pi : B.pinfos) B.sinfos.emplace_back(pi);
	// local interface indices exactly as SchurHelper<D>::indexDomainIfacesLocal assigns them
	// (SchurHelper.h:377-397): first-seen order of SchurInfo::getIds()
	std::map<int, int> rev;
	for (auto &sd : B.sinfos)
		for (int id : sd.getIds())
			if (!rev.count(id)) {
				int v   = (int) rev.size();
				rev[id] = v;
			}
	for (auto &sd : B.sinfos) sd.setLocalIndexes(rev);
	B.num_ifaces = (int) rev.size();
	return B;
}
template <size_t D> std::array<int, D> lens(int n)
{
	std::array<int, D> a;
	a.fill(n);
	return a;
}
template <size_t D> std::shared_ptr<ValVector<D>> wrap(int n, int P, const double *src)
{
	// num_patches == 1 makes ValVector use patch_stride 0 (ValVector.h:44-48), so always ask for >= 2
	std::shared_ptr<ValVector<D>> v(new ValVector<D>(lens<D>(n), P < 2 ? 2 : P));
	size_t                        cnt = (size_t) P;
	for (size_t i = 0; i < D; i++) cnt *= n;
	if (src) memcpy(&v->vec[0], src, sizeof(double) * cnt);
	return v;
}
template <size_t D> void unwrap(const ValVector<D> &v, int n, int P, double *dst)
{
	size_t cnt = (size_t) P;
	for (size_t i = 0; i < D; i++) cnt *= n;
	memcpy(dst, &v.vec[0], sizeof(double) * cnt);
}

// A u through the reference's interface machinery, single rank: SchurHelper<D>::apply
// (SchurHelper.h:360-376) with the VecScatter pair reduced to the identity it is on one rank.
template <size_t D> class RefOp : public Operator<D>
{
	public:
	mutable Built<D>                   B;
	int                                n, P;
	std::shared_ptr<IfaceInterp<D>>    interp;
	std::shared_ptr<PatchOperator<D>>  op;
	void apply(std::shared_ptr<const Vector<D>> x, std::shared_ptr<Vector<D>> b) const override
	{
		auto gamma = wrap<D - 1>(n, B.num_ifaces, nullptr);
		gamma->set(0);
		for (auto &sd : B.sinfos) interp->interpolate(sd, x, gamma);
		b->set(0);
		for (auto &sd : B.sinfos)
			op->applyWithInterface(sd, x->getLocalData(sd.pinfo->local_index), gamma,
			                       b->getLocalData(sd.pinfo->local_index));
	}
};
template <size_t D> class RefVG : public VectorGenerator<D>
{
	public:
	int n, P;
	std::shared_ptr<Vector<D>> getNewVector() override { return wrap<D>(n, P, nullptr); }
};

template <size_t D> std::shared_ptr<IfaceInterp<D>> makeInterp();
template <> std::shared_ptr<IfaceInterp<3>> makeInterp<3>() { return std::shared_ptr<IfaceInterp<3>>(new TriLinInterp()); }
template <> std::shared_ptr<IfaceInterp<2>> makeInterp<2>() { return std::shared_ptr<IfaceInterp<2>>(new BilinearInterpolator()); }

template <size_t D> int numIfaces(const ref_level *L) { return build<D>(L).num_ifaces; }
template <size_t D> void ifaceIndex(const ref_level *L, int32_t *out)
{
	auto B = build<D>(L);
	for (int p = 0; p < L->P; p++)
		for (int s = 0; s < (int) (2 * D); s++)
			out[p * 2 * D + s] = B.pinfos[p]->hasNbr(s) ? B.sinfos[p].getIfaceLocalIndex(s) : -1;
}
template <size_t D> void interp(const ref_level *L, const double *u, double *gamma)
{
	auto B  = build<D>(L);
	auto uv = wrap<D>(L->n, L->P, u);
	auto gv = wrap<D - 1>(L->n, B.num_ifaces, nullptr);
	gv->set(0);
	auto it = makeInterp<D>();
	for (auto &sd : B.sinfos) it->interpolate(sd, uv, gv);
	unwrap<D - 1>(*gv, L->n, B.num_ifaces, gamma);
}
template <size_t D> void applyWithGamma(const ref_level *L, const double *u, const double *gamma, double *f, int seven)
{
	auto B  = build<D>(L);
	auto uv = wrap<D>(L->n, L->P, u);
	auto gv = wrap<D - 1>(L->n, B.num_ifaces, gamma);
	auto fv = wrap<D>(L->n, L->P, nullptr);
	std::shared_ptr<PatchOperator<D>> op(new StarPatchOp<D>());
	for (auto &sd : B.sinfos)
		op->applyWithInterface(sd, uv->getLocalData(sd.pinfo->local_index), gv, fv->getLocalData(sd.pinfo->local_index));
	(void) seven;
	unwrap<D>(*fv, L->n, L->P, f);
}
template <size_t D> void patchApply(const ref_level *L, const double *u, double *f)
{
	auto B  = build<D>(L);
	auto uv = wrap<D>(L->n, L->P, u);
	auto fv = wrap<D>(L->n, L->P, nullptr);
	StarPatchOp<D> op;
	for (auto &sd : B.sinfos) op.apply(sd, uv->getLocalData(sd.pinfo->local_index), fv->getLocalData(sd.pinfo->local_index));
	unwrap<D>(*fv, L->n, L->P, f);
}
template <size_t D> void addIfaceRhs(const ref_level *L, const double *gamma, double *f)
{
	auto B  = build<D>(L);
	auto gv = wrap<D - 1>(L->n, B.num_ifaces, gamma);
	auto fv = wrap<D>(L->n, L->P, f);
	StarPatchOp<D> op;
	for (auto &sd : B.sinfos) op.addInterfaceToRHS(sd, gv, fv->getLocalData(sd.pinfo->local_index));
	unwrap<D>(*fv, L->n, L->P, f);
}
template <size_t D> int bicgstab(const ref_level *L, const double *b, double *x, int max_it, double tol)
{
	std::shared_ptr<RefOp<D>> A(new RefOp<D>());
	A->B      = build<D>(L);
	A->n      = L->n;
	A->P      = L->P;
	A->interp = makeInterp<D>();
	A->op.reset(new StarPatchOp<D>());
	std::shared_ptr<RefVG<D>> vg(new RefVG<D>());
	vg->n   = L->n;
	vg->P   = L->P;
	auto xv = wrap<D>(L->n, L->P, x);
	auto bv = wrap<D>(L->n, L->P, b);
	int  it = BiCGStab<D>::solve(vg, A, xv, bv, nullptr, max_it, tol);
	unwrap<D>(*xv, L->n, L->P, x);
	return it;
}
// One block-Jacobi sweep with the reference's Krylov patch solver. PatchSolvers/BiCGStabSolver.h itself cannot be compiled here
// (it includes Domain.h -> PETSc and fftw3.h), so the calls its solve() makes (BiCGStabSolver.h:114-132: copy f, addInterfaceToRHS,
// BiCGStab<D>::solve over a one-patch operator from the patch's current values) are issued here on the reference's own classes:
// StarPatchOp<D>::addInterfaceToRHS / apply, BiCGStab<D>::solve, ValVector<D>. The interface values come from the interpolator on
// the old iterate, as SchurHelper<D>::solveWithInterface's caller provides them (SchurHelper.h:318-331).
template <size_t D> class OnePatchOp : public Operator<D>
{
	public:
	SchurInfo<D>                      sinfo;
	std::shared_ptr<PatchOperator<D>> op;
	void apply(std::shared_ptr<const Vector<D>> x, std::shared_ptr<Vector<D>> b) const override
	{
		op->apply(const_cast<SchurInfo<D> &>(sinfo), x->getLocalData(0), b->getLocalData(0));
	}
};
template <size_t D> void smoothBcgs(const ref_level *L, const double *f, double *u, double tol, int max_it, int32_t *its)
{
	auto B  = build<D>(L);
	auto uv = wrap<D>(L->n, L->P, u);
	auto gv = wrap<D - 1>(L->n, B.num_ifaces, nullptr);
	gv->set(0);
	auto it = makeInterp<D>();
	for (auto &sd : B.sinfos) it->interpolate(sd, uv, gv);
	size_t nc = 1;
	for (size_t i = 0; i < D; i++) nc *= L->n;
	std::shared_ptr<PatchOperator<D>> op(new StarPatchOp<D>());
	std::shared_ptr<RefVG<D>>         vg(new RefVG<D>());
	vg->n = L->n;
	vg->P = 1;
	for (auto &sd : B.sinfos) {
		const int p   = sd.pinfo->local_index;
		auto      f1  = wrap<D>(L->n, 1, f + (size_t) p * nc); // f_copy->copy(f_single)
		auto      u1  = wrap<D>(L->n, 1, u + (size_t) p * nc);
		op->addInterfaceToRHS(sd, gv, f1->getLocalData(0));
		std::shared_ptr<OnePatchOp<D>> A(new OnePatchOp<D>());
		A->sinfo = sd;
		A->op    = op;
		int n_it = BiCGStab<D>::solve(vg, A, u1, f1, nullptr, max_it, tol);
		if (its) its[p] = n_it;
		unwrap<D>(*u1, L->n, 1, u + (size_t) p * nc);
	}
}
} // namespace

extern "C" {
int ref_init(void)
{
	int flag = 0;
	MPI_Initialized(&flag);
	if (!flag) MPI_Init(nullptr, nullptr);
	return 0;
}
int  ref_num_ifaces(const ref_level *L) { return L->dim == 3 ? numIfaces<3>(L) : numIfaces<2>(L); }
void ref_iface_index(const ref_level *L, int32_t *out) { L->dim == 3 ? ifaceIndex<3>(L, out) : ifaceIndex<2>(L, out); }
void ref_interp(const ref_level *L, const double *u, double *gamma) { L->dim == 3 ? interp<3>(L, u, gamma) : interp<2>(L, u, gamma); }
void ref_apply_with_gamma(const ref_level *L, const double *u, const double *gamma, double *f)
{
	L->dim == 3 ? applyWithGamma<3>(L, u, gamma, f, 0) : applyWithGamma<2>(L, u, gamma, f, 0);
}
void ref_patch_apply(const ref_level *L, const double *u, double *f) { L->dim == 3 ? patchApply<3>(L, u, f) : patchApply<2>(L, u, f); }
void ref_add_iface_rhs(const ref_level *L, const double *gamma, double *f) { L->dim == 3 ? addIfaceRhs<3>(L, gamma, f) : addIfaceRhs<2>(L, gamma, f); }
// SevenPtPatchOperator twin (3D only) for a second reading of a3
void ref_apply_with_gamma_7pt(const ref_level *L, const double *u, const double *gamma, double *f)
{
	auto B  = build<3>(L);
	auto uv = wrap<3>(L->n, L->P, u);
	auto gv = wrap<2>(L->n, B.num_ifaces, gamma);
	auto fv = wrap<3>(L->n, L->P, nullptr);
	SevenPtPatchOperator op;
	for (auto &sd : B.sinfos)
		op.applyWithInterface(sd, uv->getLocalData(sd.pinfo->local_index), gv, fv->getLocalData(sd.pinfo->local_index));
	unwrap<3>(*fv, L->n, L->P, f);
}
int ref_bicgstab(const ref_level *L, const double *b, double *x, int max_it, double tol)
{
	ref_init();
	return L->dim == 3 ? bicgstab<3>(L, b, x, max_it, tol) : bicgstab<2>(L, b, x, max_it, tol);
}
void ref_smooth_bcgs(const ref_level *L, const double *f, double *u, double tol, int max_it, int32_t *its)
{
	ref_init();
	L->dim == 3 ? smoothBcgs<3>(L, f, u, tol, max_it, its) : smoothBcgs<2>(L, f, u, tol, max_it, its);
}
// Vector<D> BLAS-1 virtuals (Vector.h:190-321) on ValVector<3>; op codes follow te_hip.h order
void ref_vecop(int op, int n, int P, double *v, const double *a, const double *b, double alpha, double beta, double gamma)
{
	auto vv = wrap<3>(n, P, v);
	auto av = a ? wrap<3>(n, P, a) : nullptr;
	auto bv = b ? wrap<3>(n, P, b) : nullptr;
	switch (op) {
		case 0: vv->set(alpha); break;
		case 1: vv->scale(alpha); break;
		case 2: vv->shift(alpha); break;
		case 3: vv->copy(av); break;
		case 4: vv->add(av); break;
		case 5: vv->addScaled(alpha, av); break;
		case 6: vv->addScaled(alpha, av, beta, bv); break;
		case 7: vv->scaleThenAdd(alpha, av); break;
		case 8: vv->scaleThenAddScaled(alpha, beta, av); break;
		case 9: vv->scaleThenAddScaled(alpha, beta, av, gamma, bv); break;
	}
	unwrap<3>(*vv, n, P, v);
}
// Tree<D>(file) + refineLeaves x divides -> node table in ascending id order (OctTree.h:90-213)
int ref_tree_nodes(const char *path, int dim, int divides, int max_nodes, int32_t *ilp, double *lengths, double *starts,
                   int32_t *nbr, int32_t *child, int32_t *num_levels)
{
	if (dim == 3) {
		Tree<3> t(path);
		for (int i = 0; i < divides; i++) t.refineLeaves();
		if ((int) t.nodes.size() > max_nodes) return -(int) t.nodes.size();
		int i = 0;
		for (auto &p : t.nodes) {
			Node<3> &nd = p.second;
			ilp[3 * i] = nd.id, ilp[3 * i + 1] = nd.level, ilp[3 * i + 2] = nd.parent;
			for (int a = 0; a < 3; a++) lengths[3 * i + a] = nd.lengths[a], starts[3 * i + a] = nd.starts[a];
			for (int s = 0; s < 6; s++) nbr[6 * i + s] = nd.nbr_id[s];
			for (int o = 0; o < 8; o++) child[8 * i + o] = nd.child_id[o];
			i++;
		}
		*num_levels = t.num_levels;
		return i;
	}
	Tree<2> t(path);
	for (int i = 0; i < divides; i++) t.refineLeaves();
	if ((int) t.nodes.size() > max_nodes) return -(int) t.nodes.size();
	int i = 0;
	for (auto &p : t.nodes) {
		Node<2> &nd = p.second;
		ilp[3 * i] = nd.id, ilp[3 * i + 1] = nd.level, ilp[3 * i + 2] = nd.parent;
		for (int a = 0; a < 2; a++) lengths[2 * i + a] = nd.lengths[a], starts[2 * i + a] = nd.starts[a];
		for (int s = 0; s < 4; s++) nbr[4 * i + s] = nd.nbr_id[s];
		for (int o = 0; o < 4; o++) child[4 * i + o] = nd.child_id[o];
		i++;
	}
	*num_levels = t.num_levels;
	return i;
}
} // extern "C"
