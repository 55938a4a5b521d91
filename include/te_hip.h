/* te_hip.h — C ABI of the MI355X-native GMG V-cycle path (libte_hip.so).
 *
 * Drop-in boundary for GEM3D/pressurePoissonSolver's src/Thunderegg GMG hot path. Every entry
 * point names the reference interface it stands behind (paths relative to the reference
 * root). Plain pointers, sizes and opaque handles only; no C++ or torch types. All functions
 * return TE_OK (0) or a negative TE_E* code; te_last_error() gives the message. The C++
 * adaptors in pressurepoissonsolver_amd/thunderegg/ turn a non-zero status into the
 * reference's own error convention (`throw 3;`, e.g. GMG/InterLevelComm.h:175).
 *
 * Vector layout on the host side of upload/download is the reference's: patch-major, x-fastest,
 * interior cells only, patch p at offset p*n^dim (src/Thunderegg/PetscVector.h:70-98), with
 * patches in THIS library's local order (te_hier_level_ids gives the tree node id of each).
 */
#ifndef TE_HIP_H
#define TE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TE_OK 0
#define TE_EINVAL -1   /* bad argument / shape mismatch */
#define TE_EHIP -2     /* HIP runtime error (no device, launch failure, OOM) */
#define TE_ESTATE -3   /* call not valid in the object's current state */
#define TE_EIO -4      /* mesh file unreadable */
#define TE_EUNSUPPORTED -5
#define TE_ENOMEM -6    /* host allocation failed */

typedef struct te_mesh te_mesh; /* octree / quadtree              (OctTree.h:34 Tree<D>) */
typedef struct te_hier te_hier; /* host level tables, one rank    (ThundereggDomGen.h:95-222 + Domain.h) */
typedef struct te_gmg  te_gmg;  /* device-resident level stack    (GMG/CycleFactory3d.cpp:69-134 product) */
typedef struct te_vec  te_vec;  /* device vector on one level     (Vector.h:179 Vector<D>, PetscVector.h:59) */

const char *te_last_error(void);
const char *te_version(void);

/* ---------------------------------------------------------------- mesh (host only, no GPU) */
/* Tree<D>::Tree(std::string) OctTree.h:90-118 */
int te_mesh_read(const char *path, int dim, te_mesh **out);
/* one root node on the unit box (what a 1-node mesh file such as apps/3d/meshes/1uni.bin holds) */
int te_mesh_unit_root(int dim, te_mesh **out);
/* Tree<D>::refineLeaves OctTree.h:119-179 (== one unit of the drivers' --divide) */
int te_mesh_refine_leaves(te_mesh *m);
int te_mesh_num_nodes(const te_mesh *m);
int te_mesh_num_levels(const te_mesh *m);
int te_mesh_dim(const te_mesh *m);
/* node table in ascending id order; any pointer may be NULL.
 * ilp[N][3] = id, level, parent; lengths/starts [N][dim]; nbr [N][2*dim]; child [N][2^dim] */
int  te_mesh_get_nodes(const te_mesh *m, int32_t *ilp, double *lengths, double *starts,
                       int32_t *nbr, int32_t *child);
void te_mesh_destroy(te_mesh *m);

/* ----------------------------------------------------- level hierarchy (host only, no GPU) */
/* ThundereggDomGen<D>(t, ns, neumann) + the level loop of CycleFactory3d::getCycle
 * (CycleFactory3d.cpp:98-127; max_levels / patches_per_proc from GMG/CycleOpts.h:55-63).
 * The Zoltan partition is replaced by contiguous Morton ranges over `nranks`. */
int te_hier_build(const te_mesh *m, int n, int neumann, int max_levels, double patches_per_proc,
                  int rank, int nranks, te_hier **out);
/* The same with the placement of the small levels over the ranks spelled out instead of taken from the environment
 * (te_hier_build reads TE_AGGLOMERATE, TE_AGGLOMERATE_MAX, TE_REPLICATE once per call and comes here): a level with fewer
 * than `agglomerate` patches per rank and at most `agglomerate_max` patches in total, and every level below it, is gathered --
 * on every rank (`replicate` != 0; 3D, and since round 5 2D) or on rank 0. A negative value = the default (64, 64, 1); agglomerate = 0 never
 * gathers. What the reference does instead is cut the hierarchy (patches_per_proc, CycleFactory3d.cpp:104). Every rank must
 * pass the same values: te_vcycle / te_bicgstab compare them across the ranks before the first cycle (TE_ESTATE, by name). */
int te_hier_build_placed(const te_mesh *m, int n, int neumann, int max_levels, double patches_per_proc,
                         int rank, int nranks, double agglomerate, int agglomerate_max, int replicate, te_hier **out);
/* the placement this hierarchy was built with (any pointer may be NULL) */
int te_hier_placement(const te_hier *h, double *agglomerate, int *agglomerate_max, int *replicate);
int te_hier_num_levels(const te_hier *h);
int te_hier_dim(const te_hier *h);
int te_hier_n(const te_hier *h);
/* level 0 = finest. P_local = this rank's patches, P_global = all ranks'. */
int te_hier_level_sizes(const te_hier *h, int level, int *P_local, int *P_global);
/* 1: the level lives on EVERY rank (a gathered coarse level, TE_REPLICATE: each rank holds and computes all
 * of it, P_local == P_global, and the `rank` column of te_hier_level_tables names the calling rank for every patch); 0: every
 * patch has one owner; < 0: error. Replaces nothing in the reference (CycleFactory3d.cpp:104 cuts the hierarchy instead). */
int te_hier_level_replicated(const te_hier *h, int level);
/* Global tables of one level (Morton order); any pointer may be NULL.
 * id[P] rank[P] local[P] starts[P][dim] lengths[P][dim] nbr_kind[P][2dim] (0 none, 1 normal,
 * 2 coarse, 3 fine) nbr[P][2dim][4] nbr_orth[P][2dim] parent[P] orth_on_parent[P] */
int te_hier_level_tables(const te_hier *h, int level, int32_t *id, int32_t *rank, int32_t *local,
                         double *starts, double *lengths, int32_t *nbr_kind, int32_t *nbr,
                         int32_t *nbr_orth, int32_t *parent, int32_t *orth_on_parent);
/* local -> global patch index of this rank's patches */
int  te_hier_level_l2g(const te_hier *h, int level, int32_t *l2g);
void te_hier_destroy(te_hier *h);

/* ------------------------------------------------------------------------ device objects */
typedef struct {
	int32_t pre_sweeps, post_sweeps, coarse_sweeps, mid_sweeps; /* GMG/CycleOpts.h:64-79 */
	int32_t cycle_type; /* 0 "V" (VCycle.h), 1 "W" (WCycle.h) */
	int32_t smoother;   /* TE_SMOOTH_* */
	double  omega;      /* Jacobi weight */
	int32_t exact_coarse; /* pointwise smoothers: exact patch solve on a 1-patch coarsest level */
	int32_t fuse;         /* 0: one kernel per reference call (apply, scaleThenAdd, restrict, set, ...);
	                         1: inside te_vcycle use the fused kernels whose results are BIT-IDENTICAL to 0
	                            (residual+restrict, zero-guess first sweep, sweep on u + P e);
	                         2: additionally to 1, with one RB-GS pre-sweep on a uniformly refined 3D level,
	                            the sweep from the zero iterate, the residual and its restriction are one pass over f;
	                            the coarse right-hand side differs from 1 by a few ulp along patch faces (the ghost
	                            term of the residual is added separately), independent of the partition; with one
	                            block-Jacobi pre-sweep the residual after the exact patch solves is taken on the face
	                            layers only (it vanishes inside a patch up to the rounding of the solve);
	                         3 (default): 2, and with exactly one RB-GS pre-sweep and a post-sweep in a V-cycle the
	                            iterate between them is never stored: the post-sweep kernel recomputes it from f
	                            (bit-identical to 2).
	                         In 2 and 3 the fused pre-sweep forms the residual it restricts on RED cells only and takes
	                         the black cells' residual as exactly 0: a black cell was relaxed last, from the very values
	                         its residual is formed with, so what is dropped is the rounding of that one update (~1e-16
	                         relative in the coarse right-hand side) -- inside every stated tolerance, but not
	                         bit-identical to 1 */
} te_cycle_opts;

#define TE_SMOOTH_PATCH_SOLVE 0 /* reference: FFTBlockJacobiSmoother.h:55-58 (block Jacobi, exact patch solves) */
#define TE_SMOOTH_JACOBI 1      /* weighted point Jacobi */
#define TE_SMOOTH_RBGS 2        /* patch-local red-black Gauss-Seidel, neighbour ghosts frozen */
#define TE_SMOOTH_PATCH_BCGS 3  /* 2D only: block Jacobi whose patch solves are the reference's OTHER patch solver, PatchSolvers/
                                   BiCGStabSolver.h:114-132 (apps/2d/steady.cpp:326-327, --patch_solver bcgs): per patch, unpreconditioned
                                   BiCGStab<2>::solve (BiCGStab.h:45-106) on StarPatchOp<2>::apply from the patch's current values, to
                                   te_gmg_set_patch_bcgs's tolerance; one workgroup per patch, the Krylov vectors in registers */

void te_cycle_opts_default(te_cycle_opts *o);

/* Creates the HIP stream, uploads every level's tables, allocates all scratch. Fails with
 * TE_EHIP when no gfx950 device is usable — there is no CPU fallback.
 * device < 0 selects the current device. */
int  te_gmg_create(const te_hier *h, int device, te_gmg **out);
void te_gmg_destroy(te_gmg *g);
int  te_gmg_num_levels(const te_gmg *g);
int  te_gmg_sync(te_gmg *g);            /* hipStreamSynchronize on the solver stream */
void *te_gmg_stream(te_gmg *g);         /* hipStream_t, for event timing by the caller */

/* VectorGenerator<D>::getNewVector (Vector.h:323-327; DomainVG Domain.h:415-429): zero-filled */
int    te_vec_create(te_gmg *g, int level, te_vec **out);
void   te_vec_destroy(te_vec *v);
size_t te_vec_size(const te_vec *v);                 /* doubles (local patches * n^dim) */
int    te_vec_upload(te_vec *v, const double *host); /* Vector<D>::getLocalData write path */
int    te_vec_download(const te_vec *v, double *host);
/* Vector<D>::getLocalData(i) (Vector.h:214-215, PetscVector.h:87-98) for patches [first_patch, first_patch+npatches):
 * host holds npatches * n^dim doubles in the same layout. The per-patch data plane of the C++ adaptor
 * (HipVector::getLocalData) and of Init-style fill loops (apps/shared/Init.cpp:152-245). */
int    te_vec_upload_patches(te_vec *v, int first_patch, int npatches, const double *host);
int    te_vec_download_patches(const te_vec *v, int first_patch, int npatches, double *host);
void  *te_vec_device_ptr(te_vec *v);

/* Vector<D> BLAS-1 virtuals, Vector.h:190-321 (same names, same argument order) */
int te_vec_set(te_vec *v, double alpha);
int te_vec_scale(te_vec *v, double alpha);
int te_vec_shift(te_vec *v, double delta);
int te_vec_copy(te_vec *v, const te_vec *b);
int te_vec_add(te_vec *v, const te_vec *b);
int te_vec_add_scaled(te_vec *v, double alpha, const te_vec *b);
int te_vec_add_scaled2(te_vec *v, double alpha, const te_vec *a, double beta, const te_vec *b);
int te_vec_scale_then_add(te_vec *v, double alpha, const te_vec *b);
int te_vec_scale_then_add_scaled(te_vec *v, double alpha, double beta, const te_vec *b);
int te_vec_scale_then_add_scaled2(te_vec *v, double alpha, double beta, const te_vec *b,
                                  double gamma, const te_vec *c);
/* local (this rank's) partial results; the caller all-reduces (Vector.h:294,306,319) */
int te_vec_two_norm_sq(const te_vec *v, double *out);
int te_vec_inf_norm(const te_vec *v, double *out);
int te_vec_dot(const te_vec *v, const te_vec *b, double *out);
/* This rank's part of the vector's CHECKSUM: the sum modulo 2^64 of the 64-bit patterns of its values. Integer addition
 * commutes, so the sum of the ranks' parts (modulo 2^64; the caller adds them, as for the norms) does not depend on the order of
 * the patches nor on how they are cut over ranks: a sharded run that is bit-identical to the single-rank run -- what every
 * operation of a cycle is by construction -- prints the same number. How a job on hardware nobody can inspect shows that N
 * ranks computed the very bits one rank computes (bench.py: u_checksum_after_timed_region). A level that lives on every rank
 * counts once (rank 0's), as in te_vec_dot. Replaces nothing in the reference; the equality it proves is the one between
 * `mpirun -np N` and `-np 1` of SchurHelper.h:123-150 / GMG/InterLevelComm.h:169-189 for order-independent operations. */
int te_vec_checksum(const te_vec *v, uint64_t *out);

/* Operator<D>::apply (Operators/Operator.h:37) as SchurDomainOp / DomainWrapOp implement it:
 * f = A u through SchurHelper::apply (SchurHelper.h:360-376) */
int te_apply(te_gmg *g, int level, const te_vec *u, te_vec *f);
/* PatchOperator<D>::apply without interface values, StarPatchOp.h:204-319 (SevenPtPatchOperator.cpp:247-409,
 * FivePtPatchOperator.h:172-261): f = A_patch u per patch, faces with a neighbour closed as homogeneous
 * Dirichlet. The operator the exact patch solves invert; the reference uses it in PatchSolvers/BiCGStabSolver.h:82-85. */
int te_patch_apply(te_gmg *g, int level, const te_vec *u, te_vec *f);
/* r = f - A u  (Cycle.h:60-61 fused: apply + scaleThenAdd(-1, f)) */
int te_residual(te_gmg *g, int level, const te_vec *u, const te_vec *f, te_vec *r);
/* the same with ||r||^2 (this rank's part; Vector.h:294 twoNorm before its MPI_Allreduce and sqrt) summed by the residual
 * kernel itself while r is in registers: per thread in plane order, then wave shuffles -> LDS -> one partial per workgroup
 * -> a fixed-order final pass. No second pass over r (3D and 2D). */
int te_residual_norm_sq(te_gmg *g, int level, const te_vec *u, const te_vec *f, te_vec *r, double *norm_sq);
/* GMG::Smoother<D>::smooth(f, u) (GMG/Smoother.h:39), `sweeps` times */
int te_smooth(te_gmg *g, int level, const te_vec *f, te_vec *u, int smoother, double omega,
              int sweeps);
/* GMG::Restrictor<D>::restrict(coarse, fine) (GMG/Restrictor.h:39) == AvgRstr.h:78-113.
 * `fine_level` is the level of `fine`; coarse lives on fine_level+1. */
int te_restrict(te_gmg *g, int fine_level, const te_vec *fine, te_vec *coarse);
/* GMG::Interpolator<D>::interpolate(coarse, fine) (GMG/Interpolator.h:39) == DrctIntp.h:80-113 */
int te_prolong_add(te_gmg *g, int fine_level, const te_vec *coarse, te_vec *fine);
/* GMG::Cycle<D>::apply(f, u) (GMG/Cycle.h:116-126) on level 0 */
int te_vcycle(te_gmg *g, const te_cycle_opts *o, const te_vec *f, te_vec *u);
/* BiCGStab<D>::solve(vg, A, x, b, Mr, max_it, tol) (BiCGStab.h:45-106). Mr = te_vcycle when
 * `o` is non-NULL. On a sharded hierarchy every rank calls it; the scalars of an iteration are summed over
 * the ranks (Vector.h:294,319) by ncclAllReduce on the solver stream (te_gmg_use_rccl) or by the
 * te_gmg_set_allreduce callback, batched as BiCGStab.h:71-97 allows: 1 + 2 + 2 doubles per iteration.
 * TE_ESTATE when several ranks exist and neither is set. */
int te_bicgstab(te_gmg *g, const te_cycle_opts *o, te_vec *x, const te_vec *b, int max_it,
                double tol, int *iterations, double *rel_resid);
/* te_bicgstab allocates its eight level-0 work vectors at its first call and keeps them for the next solve (a driver
 * solves again and again; 8 x the size of x: 8 GiB at 512^3). They are freed by te_gmg_destroy -- or by this call, for a
 * caller that has finished solving and wants the memory back (the next te_bicgstab allocates them again). */
int te_gmg_release_workspace(te_gmg *g);

/* The TE_* switches (docs/SWITCHES.md) are read from the environment once, in te_gmg_create. This call sets (value) or
 * clears (NULL) one of them for this solver afterwards -- how the tests pin one implementation against another. TE_ESTATE
 * for the few that shape the level tables and are therefore fixed at creation; TE_EINVAL for an unknown name. */
int te_gmg_set_option(te_gmg *g, const char *name, const char *value);

/* ------------------------------------------------------------ multi-rank ghost exchange */
/* The library never talks to the network itself. When a level has off-rank neighbours, it
 * packs the needed face layers into one send buffer, calls `exchange`, and reads the receive
 * buffer. `exchange` must move, for every peer r: send[send_off[r] .. +send_cnt[r]) to rank r
 * and fill recv[recv_off[r] .. +recv_cnt[r]) from rank r (counts in doubles, device
 * pointers), ordered on `stream` (hipStream_t). This replaces the PETSc VecScatter of
 * SchurHelper.h:123-150 and GMG/InterLevelComm.h:169-189; the Python host binds it to
 * torch.distributed (RCCL) batch_isend_irecv. */
typedef int (*te_exchange_fn)(void *user, int tag, const double *send, double *recv, int npeers,
                              const int32_t *peers, const int64_t *send_off,
                              const int64_t *send_cnt, const int64_t *recv_off,
                              const int64_t *recv_cnt, void *stream);
int te_gmg_set_exchange(te_gmg *g, te_exchange_fn fn, void *user);
/* Alternative to the callback: let the library issue the exchanges itself as RCCL point-to-point
 * groups (ncclGroupStart; ncclRecv/ncclSend per peer; ncclGroupEnd) on its solver stream — the
 * direct replacement of the MPI path under the PETSc VecScatter (SchurHelper.h:123-150). `libpath`
 * names the librccl.so to dlopen (the one the host process already uses); `id128` is the 128-byte
 * ncclUniqueId produced by te_rccl_unique_id on rank 0 and broadcast by the host. */
int te_rccl_unique_id(const char *libpath, char *id128);
int te_gmg_use_rccl(te_gmg *g, const char *libpath, const char *id128, int rank, int nranks);
/* Sum (op 0) or maximum (op 1) of vals[0..n) over all ranks, in place, the same result on every rank: the
 * MPI_Allreduce of Vector.h:294,306,319 for hosts that registered an exchange callback (with te_gmg_use_rccl the
 * library reduces on the device itself). Used by te_bicgstab and te_gmg_verify_schedule. */
typedef int (*te_allreduce_fn)(void *user, double *vals, int n, int op);
int te_gmg_set_allreduce(te_gmg *g, te_allreduce_fn fn, void *user);
/* Dry run of one te_vcycle with options `o` that records every exchange each rank would issue (tag, level, peer,
 * counts) and compares, through one reduction over the ranks, what every rank sends with what its peer expects,
 * pair by pair and in order. TE_ESTATE on ALL ranks when the ranks would issue different sequences (different
 * options or hierarchies) -- instead of a hang inside RCCL in the middle of a cycle. te_vcycle runs it by itself the
 * first time it sees a set of options on a sharded hierarchy (TE_NO_VERIFY skips that). Collective.
 * Independently, a watchdog thread ends the process (exit status 86, message on stderr) when an exchange has not
 * completed TE_EXCHANGE_TIMEOUT seconds (default 300; 0 = off) after it was issued. */
int te_gmg_verify_schedule(te_gmg *g, const te_cycle_opts *o);
/* Chooses, on the live communicator, how the sweeps of the sharded levels meet their face exchanges: everything in line on
 * the solver stream; the exchange on a second stream under the interior patches, boundary patches behind it (north star:
 * "ghost-cell exchange ... overlapped with interior smoothing"); or the interior patches on the second stream beside
 * exchange + boundary patches. Each candidate runs `reps` te_vcycle(o) behind two warm-up cycles on a scratch right-hand
 * side; the maximum over the ranks decides (one scalar reduction per candidate, so all ranks choose alike), the in-line
 * form winning ties within 2 %. All candidates give bit-identical results -- the choice changes no number. Collective;
 * call it once after te_gmg_use_rccl / te_gmg_set_exchange. *best_ms (may be NULL) = the chosen form's milliseconds per
 * cycle; report (may be NULL) receives one line with every candidate's time and the choice. Without a call the size rule
 * TE_OVERLAP_MIN decides. Replaces nothing in the reference (PETSc's VecScatterBegin/End pair, SchurHelper.h:123-150, is
 * the same idea: start the scatter, compute, finish it). */
int te_gmg_autotune(te_gmg *g, const te_cycle_opts *o, int reps, double *best_ms, char *report, int report_len);
/* A second transport for the exchanges that have a direct form (the face exchanges of 3D levels, the in-place exchange of
 * restricted blocks into a level that lives on every rank): each rank STORES what a peer needs straight into that peer's
 * receive buffer -- device memory of another process / GPU of the node, mapped through hipIpcGetMemHandle /
 * hipIpcOpenMemHandle -- and raises a flag there; a one-workgroup kernel on the receiver's stream waits for the flags. Two
 * small launches per exchange instead of an RCCL group. enable != 0: prepares it (collective; once; needs te_gmg_use_rccl or
 * te_gmg_set_allreduce, through which the handles and receive offsets are published) and switches it on; 0: switches back.
 * Results are bit-identical either way. A wait is bounded (TE_PUSH_TIMEOUT seconds; default: TE_EXCHANGE_TIMEOUT; at most 5 s inside
 * te_gmg_autotune's trial), and the kernels check the protocol themselves (csrc/pushkernels.hpp): te_gmg_push_failed returns 0, or
 * the first failure's code -- 1 a wait gave up, 2 a peer's flag was two exchanges ahead, 3 a peer was behind when its buffer was
 * overwritten, 4 this rank's epochs were out of sequence -- and the watchdog ends the process as for any exchange that never
 * completes (unless TE_PUSH_NONFATAL leaves that to the caller). A set-up that fails on one rank fails on all (the failure travels
 * with the directory reductions) and leaves nothing allocated or mapped. te_gmg_autotune, when this
 * transport has been prepared, first checks it against the other one ON THE MACHINE AT HAND (bit-identical result after a
 * cycle on different data, no wait given up, all ranks agreeing) and keeps it only if it passes and is faster.
 * The RCCL point-to-point path stays the default. Same replacement as te_gmg_use_rccl: SchurHelper.h:123-150,
 * GMG/InterLevelComm.h:169-189. */
int te_gmg_use_push(te_gmg *g, int enable);
int te_gmg_push_failed(te_gmg *g);

/* BiCGStabSolver(op, tol = 1e-12, max_it = 1000), PatchSolvers/BiCGStabSolver.h:103-108: the stopping rule of
 * TE_SMOOTH_PATCH_BCGS's patch solves (||resid|| / ||resid_0|| <= tol or max_it iterations, BiCGStab.h:69). */
int te_gmg_set_patch_bcgs(te_gmg *g, double tol, int max_it);
/* iterations each of this rank's patches of `level` took in the last TE_SMOOTH_PATCH_BCGS sweep there (its[P local], host memory;
 * synchronises the solver's stream). TE_ESTATE when no such sweep has run on the level. */
int te_gmg_patch_bcgs_iterations(te_gmg *g, int level, int32_t *its);
/* ncclCommCount / ncclCommUserRank of the communicator te_gmg_use_rccl created (0 / -1 without one): evidence for a
 * benchmark line that RCCL itself saw N ranks. */
int te_gmg_comm_info(te_gmg *g, int *rccl_nranks, int *rccl_rank);
/* moves n doubles through the active exchange back-end with this rank as its own peer (diagnostic) */
int te_gmg_exchange_selftest(te_gmg *g, int n);
/* diagnostic for the watchdog: `seconds` of exchanges enqueued without a host synchronisation (the host runs ahead of the
 * GPU; each exchange completes in milliseconds). Returns the number issued (> 0), or a TE_E* code; a watchdog that aged
 * its deadline from the first exchange instead of the oldest OUTSTANDING one would end the process with status 86 here
 * once `seconds` exceeds TE_EXCHANGE_TIMEOUT. With one rank the watchdog runs only when TE_EXCHANGE_TIMEOUT is set. */
int te_gmg_watchdog_selftest(te_gmg *g, double seconds);

/* Domain<D>::integrate (Domain.h:258-278) and Domain<D>::volume (:237-251), this rank's part (the host adds the
 * ranks as it does for norms): sum over local patches of (sum of the patch's cells) * (cell volume), resp. of the
 * patch volumes. What the drivers need for pure-Neumann problems (apps/3d/steady.cpp:330-334, 539-549).
 * A level that lives on EVERY rank (te_hier_level_replicated) is counted once: rank 0 returns the whole level, every other
 * rank 0.0, so that the sum over the ranks is the level's integral / volume as on any other level. The same holds for
 * te_vec_two_norm_sq and te_vec_dot on such a level (te_vec_inf_norm returns the level's value on every rank: a maximum
 * over the ranks is unchanged). */
int te_integrate(te_gmg *g, int level, const te_vec *v, double *out);
int te_volume(te_gmg *g, int level, double *out);

/* Init::initDirichlet / Init::initNeumann (apps/shared/Init.cpp:152-245, :57-151; 2D :246-361) for the drivers'
 * canned problems, evaluated on the device straight into `f` and (may be NULL) `exact`: right-hand side and exact
 * solution at the cell centres, physical Dirichlet data folded in as -2 g/h^2, Neumann data as +-g_n/h, in the
 * reference's face order. TE_PROBLEM_TRIG / TE_PROBLEM_GAUSS are apps/3d/steady.cpp:221-292 (2D: apps/2d/steady.cpp:296-318);
 * TE_PROBLEM_RANDOM is the timing input f ~ U(-1,1) from splitmix64(0x5EED + tree node id) (exact := 0). Arbitrary
 * std::function problems go through Vector<D>::getLocalData (thunderegg/HipInit.h). */
#define TE_PROBLEM_TRIG 0
#define TE_PROBLEM_GAUSS 1
#define TE_PROBLEM_RANDOM 2
int te_init_problem(te_gmg *g, int level, int problem, int neumann, te_vec *f, te_vec *exact);

/* kernel timing hooks for bench.py: HIP-event time of the last te_vcycle's dominant kernel */
int te_gmg_profile(te_gmg *g, int enable);
/* name[i] (<=63 chars), calls[i], total_ms[i] (HIP events on the solver stream), cells[i]
 * (lattice sites the launches processed); returns number of rows written (<= max_rows) */
int te_gmg_profile_rows(te_gmg *g, int max_rows, char (*name)[64], int64_t *calls, double *total_ms,
                        int64_t *cells);
int te_gmg_profile_reset(te_gmg *g);
/* time only the launches of kernel class `name` (a row name of te_gmg_profile_rows); NULL or "" = every class.
 * A HIP event pair around a launch costs a few microseconds of stream time, which matters for a V-cycle of
 * a dozen launches: bench.py times only the dominant class inside its timed region. */
int te_gmg_profile_select(te_gmg *g, const char *name);
/* ... and only every `stride`-th launch of a timed class carries its event pair (stride <= 1: every launch). On this runtime an
 * event pair on a dispatch costs microseconds of stream time (profiles/r06_event_cost.txt: 24-28 us of a 210-us 4096^2 cycle with six
 * timed launches per cycle, 0-20 us of a 512^3 cycle with one): bench.py times every fourth launch of the dominant class inside its timed region; calls / cells / total_ms of
 * te_gmg_profile_rows count the timed launches only, so averages stay what they were. */
int te_gmg_profile_stride(te_gmg *g, int stride);

/* Where te_gmg_create spent its time -- the "GMG Setup" timer of apps/3d/steady.cpp:480-484 around GMG/CycleFactory3d.cpp:69-134,
 * broken down (milliseconds, host clock): out[0] device selection, context, streams, events (the first solver of a process also pays
 * the HIP runtime's start and the load of this library's code object here); out[1] the level tables, plans and transform matrices
 * built on the host; out[2] device allocations (hipMalloc), out[3] their number; out[4] uploads of the tables (hipMemcpy);
 * out[5] the work vectors of every level (allocation + zero fill queued); out[6] the final synchronisation; out[7] the whole call.
 * n <= 8 values are written; returns TE_OK. */
int te_gmg_setup_ms(const te_gmg *g, double *out, int n);

#ifdef __cplusplus
}
#endif
#endif
