/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the reference's GMG V-cycle hot path
 * (GEM3D/pressurePoissonSolver, src/Thunderegg). Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library. The product (libte_hip.so)
 * never links, loads or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   orc_interp / orc_apply_with_gamma / orc_patch_apply / orc_add_iface_rhs / vector ops /
 *   orc_bicgstab: pinned against the reference's own compiled code (oracle/_ref, built from
 *     /root/reference sources) through tests/golden/ref_*.npz.
 *   orc_restrict / orc_prolong: pinned by the known-answer patterns of the reference's
 *     test/GMG.cpp:261-435.
 *   orc_patch_solve: the reference's solvers need FFTW / BLAS (absent) — pinned by
 *     (i) being the exact inverse of the reference's compiled StarPatchOp::apply on the golden
 *     vectors, (ii) scipy.fft DST/DCT (types 2/3/4) agreement.
 *   orc_cycle: loop structure restated from GMG/Cycle.h + VCycle.h + WCycle.h (those headers
 *     include PETSc and cannot be compiled here) — pinned only through its parts and through
 *     solve-level convergence: "cycle composition parity unpinned".
 */
#ifndef TE_ORACLE_H
#define TE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* One level, single rank, all patches. Vectors are patch-major, x-fastest, interior cells
 * only: v[p*n^dim + x + n*y + n*n*z]  (src/Thunderegg/PetscVector.h:70-98). */
typedef struct {
	int32_t        dim, n, P;
	const int32_t *id;             /* [P] tree node id (interface ids derive from it) */
	const double  *h;              /* [P*dim] cell spacings */
	const int32_t *nbr_kind;       /* [P*2dim] 0 none 1 normal 2 coarse 3 fine */
	const int32_t *nbr;            /* [P*2dim*4] patch indices */
	const int32_t *nbr_orth;       /* [P*2dim] quadrant on the coarse nbr's face */
	const int32_t *neumann;        /* [P] bit s set = physical Neumann on side s */
	const int32_t *parent;         /* [P] index in next coarser level, -1 */
	const int32_t *orth_on_parent; /* [P] orthant, -1 = copies through */
} orc_level;

typedef struct {
	int32_t pre_sweeps, post_sweeps, coarse_sweeps, mid_sweeps;
	int32_t cycle_type; /* 0 = V, 1 = W */
	int32_t smoother;   /* 0 = reference block-Jacobi patch solve; 1 = weighted Jacobi;
	                       2 = patch-local red-black GS with frozen ghosts */
	double  omega;      /* weight for smoother 1 */
	int32_t exact_coarse; /* 1: pointwise smoothers use the exact patch solve on a 1-patch
	                          coarsest level */
} orc_cycle_opts;

int  orc_num_ifaces(const orc_level *L);
/* iface_index[p*2dim+s] = local index of the interface patch p sees on side s, or -1 */
void orc_iface_index(const orc_level *L, int32_t *iface_index);

/* a6+a7: gamma = sum of every patch's contributions (TriLinInterp.cpp:60-172,
 * BilinearInterpolator.cpp:61-117, SchurHelper.h:145-150 single-rank). */
void orc_interp(const orc_level *L, const double *u, double *gamma);
/* a3: StarPatchOp.h:28-184 */
void orc_apply_with_gamma(const orc_level *L, const double *u, const double *gamma, double *f);
/* a2: SchurHelper.h:360-376 */
void orc_apply(const orc_level *L, const double *u, double *f);
/* a4: StarPatchOp.h:204-319 (no interface term) */
void orc_patch_apply(const orc_level *L, const double *u, double *f);
/* a5: StarPatchOp.h:185-203 */
void orc_add_iface_rhs(const orc_level *L, const double *gamma, double *f);
/* a9: PatchSolvers/DftPatchSolver.h:172-216 (f already holds the interface term) */
void orc_patch_solve(const orc_level *L, const double *gamma, const double *f, double *u);
/* a8: SchurHelper.h:318-331 */
void orc_smooth(const orc_level *L, const double *f, double *u);
/* the same loop with the reference's other patch solver: PatchSolvers/BiCGStabSolver.h:114-132 (BiCGStab.h:45-106 per patch on
 * StarPatchOp::apply, initial guess = the patch's current values); its[P] may be null */
void orc_smooth_bcgs(const orc_level *L, const double *f, double *u, double tol, int max_it, int *its);
/* the stopping rule orc_cycle's smoother == 3 (the loop above) uses; defaults as the reference's constructor */
void orc_set_patch_bcgs(double tol, int max_it);
/* a11 / a12: GMG/AvgRstr.h:78-113, GMG/DrctIntp.h:80-113. coarse is overwritten by restrict. */
void orc_restrict(const orc_level *fine, const orc_level *coarse, const double *fine_v,
                  double *coarse_v);
void orc_prolong_add(const orc_level *fine, const orc_level *coarse, const double *coarse_v,
                     double *fine_v);

/* pointwise smoothers of the product (NOT reference functions; restated on the CPU so the HIP
 * kernels have something to be diffed against) */
void orc_jacobi(const orc_level *L, const double *f, double *u, double omega);
void orc_patch_rbgs(const orc_level *L, const double *f, double *u);

/* a1: GMG/Cycle.h:116-126 + VCycle.h:44-62 + WCycle.h:45-68. levels[0] = finest. */
void orc_cycle(const orc_level *levels, int nlevels, const orc_cycle_opts *o, const double *f,
               double *u);
/* caller: BiCGStab.h:45-106, A = orc_apply on levels[0], M = orc_cycle (use_prec) */
int orc_bicgstab(const orc_level *levels, int nlevels, const orc_cycle_opts *o, int use_prec,
                 const double *b, double *x, int max_it, double tol, double *final_rel_resid);

void orc_set_threads(int nthreads);
/* the exact patch solve's transforms in O(n log n) (own radix-2 FFT: what PatchSolvers/FftwPatchSolver.h:93-206 does through FFTW, the
 * reference's default --patch_solver) instead of the dense products of DftPatchSolver.h:295-347; for the TIMED CPU baseline only: the
 * parity oracle keeps the dense products (the two agree to 1e-12: tests/test_oracle_fast_transforms.py) */
void orc_set_fast_transforms(int on);
#ifdef __cplusplus
}
#endif
#endif
