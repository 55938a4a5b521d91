// What the translation units of the device half of include/te_hip.h share: the solver's state (te_gmg, one LevelHost per level),
// the option table, the timing scopes, and the functions one unit calls in another. Host C++ + HIP for gfx950 only; there is
// no CPU fallback anywhere: if HIP cannot give us a device, te_gmg_create fails with TE_EHIP.
//   gmg_core.hip       level tables on the device (buildLevel), solver / vector life cycle, options, profiling, Init kernels
//   gmg_transport.hip  exchanges between ranks: RCCL binding, host callback, direct-store transport, scalar reductions, watchdog
//   gmg_launch3d.hip   every 3D kernel launch (stencil, sweeps, fused sweeps, patch solves, transfers)
//   gmg_launch2d.hip   the 2D twins
//   gmg_cycle.hip      the cycle driver (GMG/Cycle.h, VCycle.h, WCycle.h), schedule check, te_gmg_autotune, per-operation entries
//   gmg_krylov.hip     Vector<D> BLAS-1 entries and te_bicgstab (BiCGStab.h:45-106)
#pragma once
#include "capi_common.hpp"
#include <hip/hip_ext.h>
#include <rccl/rccl.h> // enum values and ncclUniqueId only: the library itself is dlopen'ed (te_gmg_use_rccl)
#include "kernels3d.hpp"
#include "march3d.hpp"
#include "kernels2d.hpp"
#include "patchsolve32.hpp"
#include "patchsolve32_sym.hpp"
#include "patchsolve16.hpp"
#include "initkernels.hpp"
#include "pushkernels.hpp"
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <tuple>
#include <unistd.h>
#include <vector>

using namespace te;

#define HIPCHK(expr)                                                                              \
	do {                                                                                          \
		hipError_t _e = (expr);                                                                   \
		if (_e != hipSuccess) {                                                                   \
			(void) hipGetLastError(); /* reported here: a later launch check must not find it again */ \
			return te::fail(TE_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
		}                                                                                         \
	} while (0)


struct te_vec {
	te_gmg *g     = nullptr;
	int     level = 0;
	double *d     = nullptr;
	size_t  n     = 0;
};

namespace tei
{
enum KClass : int {
	KC_APPLY, KC_RESID, KC_JACOBI, KC_RBGS, KC_CFGHOST, KC_RESTRICT, KC_PROLONG, KC_PATCH_RHS,
	KC_DST, KC_VECOP, KC_REDUCE, KC_PACK, KC_EXCHANGE, KC_RBGS_ZERO, KC_RESID_RESTRICT, KC_PS_MFMA, KC_RBGS_PROLONG,
	// launches on levels with few patches run other instantiations (z-slabs, split patches): classes of their own, so
	// that a class above is one kernel symbol and its average duration is the one rocprofv3 --stats reports
	KC_RBGS_SLABS, KC_STENCIL_SLABS, KC_PS_3PASS, KC_ZERO_RESID, KC_FIXUP, KC_RESWEEP, KC_ZERO_RESID_FACES,
	// the instantiations that read their right-hand side together with exported ghost terms (FCORR): other symbols again
	KC_RESWEEP_FCORR, KC_ZERO_RESID_FACES_FCORR, KC_FCORR_GATHER,
	// the reference smoother's zero-guess pre-sweep that stores face layers only (k_ps_sym<false, FACES>): other bytes per site
	KC_PS_MFMA_FACES,
	// te_bicgstab's own passes: x / resid update with its two dot products (72 B/site), the stand-alone s and p statements
	// (24 / 32), the operator application that also sums one or two dot products (16 + 8)
	KC_BICG_UPDATE, KC_BICG_S, KC_BICG_P, KC_APPLY_DOT,
	// the patch-local Krylov solve of PatchSolvers/BiCGStabSolver.h (2D): compute-resident, its HBM bytes are one read of the
	// right-hand side and a read + write of the patch (24 B/site) whatever the iteration count
	KC_PATCH_BCGS, KC_COUNT
};
extern const char *kclassName[KC_COUNT]; // (gmg_core.hip)

// Every TE_* switch of this library (docs/SWITCHES.md). They are read from the environment ONCE, in te_gmg_create;
// te_gmg_set_option changes one afterwards (the tests pin one implementation against another that way). Nothing on a
// launch path looks at the environment.
enum Opt : int {
	O_2D_SIMPLE, O_2D_NO_MFMA, O_2D_NO_PF, O_2D_NO_MR_FUSE, O_2D_TPB, O_NO_FUSE2, O_NO_FUSE3, O_NO_FUSE3_CF, O_NO_CFP, O_NO_XF,
	O_NO_FCORR, O_NO_FCORR_CF, O_NO_GTAB, O_NO_OVERLAP, O_OVERLAP_MIN, O_NO_PS_FACES, O_PS_MODE, O_PS_SLOW, O_RBGS_NOSLAB,
	O_ZS_FORCE, O_NO_ZS8, O_RESWEEP_V, O_EXCHANGE_TIMEOUT, O_NO_VERIFY, O_RCCL_LOOPBACK, O_ZR_AHEAD, O_NO_BICG_FUSE, O_POST_EXCHANGE, O_REPL_BLOCKS, O_PACK_FACES, O_OVERLAP_MODE, O_PUSH_TIMEOUT, O_NO_BICG_XF, O_PUSH_FAULT, O_2D_NO_FOLD, O_2D_NO_SYM, O_PUSH_NONFATAL, O_PS_NO_HALF, O_PS_HALF_MAX, O_NO_GTAB2, O_NO_RS6_CF, O_NO_RS6_FIXUP, O_NO_CFP59, O_COUNT
};
extern const char *optName[O_COUNT]; // (gmg_core.hip)
// options that shape the level tables te_gmg_create builds: fixed for the solver's lifetime
inline bool optStructural(int o) { return o == O_2D_SIMPLE || o == O_NO_CFP || o == O_2D_NO_MR_FUSE || o == O_NO_OVERLAP || o == O_EXCHANGE_TIMEOUT; }
struct Cfg {
	bool        on[O_COUNT] = {};
	std::string val[O_COUNT];
	void        set(int o, const char *v)
	{
		on[o]  = v != nullptr;
		val[o] = v ? v : "";
	}
	void fromEnv(); // (gmg_core.hip: the one place of the library's device half that reads the environment, once per solver)
	bool        has(int o) const { return on[o]; }
	const char *str(int o) const { return on[o] ? val[o].c_str() : nullptr; }
	int         num(int o, int dflt) const { return on[o] ? atoi(val[o].c_str()) : dflt; }
	double      real(int o, double dflt) const { return on[o] ? atof(val[o].c_str()) : dflt; }
};

// te_gmg_setup_ms: hipMalloc / hipMemcpy time of the te_gmg_create in progress on this thread (null outside of one)
struct SetupAcc {
	double malloc_ms = 0, copy_ms = 0, nmalloc = 0;
};
inline thread_local SetupAcc *g_setup_acc = nullptr;
struct SetupClock {
	double                               *dst;
	std::chrono::steady_clock::time_point t0;
	explicit SetupClock(double *d) : dst(d), t0(std::chrono::steady_clock::now()) {}
	~SetupClock()
	{
		if (dst) *dst += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	}
};

template <typename T> struct DevBuf {
	T     *p = nullptr;
	size_t n = 0;
	~DevBuf()
	{
		if (p) (void) hipFree(p);
	}
	int alloc(size_t count) // (an earlier allocation is released first)
	{
		if (p) (void) hipFree(p);
		p = nullptr;
		n = count;
		if (count == 0) return TE_OK;
		SetupClock c(g_setup_acc ? &g_setup_acc->malloc_ms : nullptr);
		if (g_setup_acc) g_setup_acc->nmalloc += 1;
		HIPCHK(hipMalloc(&p, sizeof(T) * count));
		return TE_OK;
	}
	int upload(const std::vector<T> &h)
	{
		int rc = alloc(h.size());
		if (rc) return rc;
		SetupClock c(g_setup_acc ? &g_setup_acc->copy_ms : nullptr);
		if (!h.empty()) HIPCHK(hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
		return TE_OK;
	}
};

// one exchange = for every peer: send [send_off, +send_cnt) and receive [recv_off, +recv_cnt) doubles
struct ExPlan {
	std::vector<int32_t> peers;
	std::vector<int64_t> send_off, send_cnt, recv_off, recv_cnt;
	bool empty() const { return peers.empty(); }
};

struct LevelHost;
// 2D, opts.fuse >= 2: the ghost terms of a coarse right-hand side that the FINER level's pre-sweep left out and that no fix-up pass
// has added: the coarse level's own pre-sweep adds them (k_rbgs_zero_resid2d_lds<.., FOLD>). fine == null: nothing pending.
struct Fold2DHost {
	LevelHost    *fine = nullptr;
	const double *u    = nullptr; // the finer level's new iterate when it was stored; null: its edge layers are in fine->e4buf
};
struct LevelHost {
	int    dim = 3, n = 0, P = 0, P_global = 0, index = 0; // index: the level's number in the solver (0 = finest)
	bool   gathered = false; // the level lives on rank 0 alone or on every rank (the hierarchy's placement): no face exchange
	bool   replicated = false; // ... on every rank: sums over the level count it once (rank 0's; te_integrate, te_vec_dot, ...)
	bool   prolong_fusable = false; // every patch is an octant child of a LOCAL parent and there is no coarse/fine face
	// refined levels: every patch has a LOCAL parent (octant children and patches that copy through), coarse/fine
	// faces allowed: the RB-GS sweep on u + P e has a variant for that (k_rbgs3d<..., CFP>)
	bool   prolong_fusable_cf = false, has_copy = false;
	// 2D: lds2d = patches fit in LDS (the same on every rank: it decides whether the zero-guess sweep skips its
	// ghost exchange, and all ranks must agree on that); fuse2d = additionally all parents are local (rank-local:
	// residual+restrict in one pass; peers see the same exchanges either way)
	bool   lds2d = false, fuse2d = false;
	// 3D: the fused pre-sweep + residual + restriction (opts.fuse = 2) applies: a level with at least 256 patches in
	// total. A global fact, so it is the same on every rank and for every partition: sharded runs take the same
	// arithmetic path as the single-rank run.
	bool   fuse2_ok = false;
	size_t nc = 0, nf = 0;
	// stencil tables
	DevBuf<int32_t> face_kind, face_src;
	// the same with every neighbour face closed as homogeneous Dirichlet: the PATCH operator, StarPatchOp::apply
	// (StarPatchOp.h:204-319); patch_local selects it for one launch (te_patch_apply)
	DevBuf<int32_t> face_kind_patch;
	bool            patch_local = false;
	DevBuf<double>  face_kadj, rh2, ghost;
	int             nslots = 0;
	// coarse/fine faces
	int             ncf = 0;
	DevBuf<int32_t> cf_desc, cf_slots;
	// remote same-level faces (multi-rank): ghost slots [0, nremote) are filled by the exchange
	ExPlan          fx;          // per-peer counts; recv lands directly in `ghost`
	int             nremote = 0; // faces received == faces sent
	DevBuf<int32_t> send_faces;  // [nremote][2] (patch, side) in send order
	DevBuf<double>  sendbuf;     // [nremote * nf]
	// transfer to level+1
	int             Pc = 0;
	DevBuf<int32_t> parent, orth, child, copy;
	DevBuf<int64_t> cbase; // [P][7] ProlongSrc::cbase (levels with prolong_fusable)
	// children / parents that live on another rank: blocks of nc/8 (or nc, copy-through) doubles
	ExPlan          tx_up, tx_down; // child side (sends in restrict), parent side (sends in prolong)
	int             n_up = 0, n_down = 0;
	// the coarser level is replicated on every rank (mesh.cpp TE_REPLICATE) and this one is not: every local patch's restricted
	// block goes to every other rank (n_up blocks, one copy in upbuf, the same range sent to each peer; bc_desc = (local coarse
	// patch, orthant) of each block for the paths that have written the coarse octants already); nothing comes back up
	bool            repl_up = false;
	DevBuf<int32_t> bc_desc;
	DevBuf<int32_t> up_desc, down_desc; // [n][2] (patch, orthant)
	DevBuf<int64_t> up_off, down_off;   // block offsets inside upbuf / downbuf
	DevBuf<double>  upbuf, downbuf;
	// patch solve
	DevBuf<int32_t> plan, zero_mode;
	DevBuf<int32_t> bcgs_its; // [P]: iterations of the last TE_SMOOTH_PATCH_BCGS sweep on this level (allocated on first use)
	DevBuf<double>  mats, lam, corr; // corr: [P][6][n^2] interface terms of the patch right-hand sides
	DevBuf<double>  matsT;           // 2D: the transform matrices transposed (k_patch_solve2d_lds)
	DevBuf<double>  matfrag;         // 32^3 patches: mats in the lane order of the three-pass kernels (patchsolve32.hpp matFragSource)
	DevBuf<double>  matsym;          // half matrices in MFMA fragment order (patchsolve32_sym.hpp), 32^3 patches
	DevBuf<double>  psinv;           // k_ps_sym: reciprocals of the eigenvalue sums, one table of PSS_INV doubles per (plan, spacings) of the level
	DevBuf<int32_t> psitab;          // [P] the patch's table in psinv
	bool            sym_ok = false;  // every plan of the level has pure (DST-II/III or DCT-II/III) axes
	DevBuf<int32_t> ps_list;         // otherwise: [patches with pure axes (n_pure) | the others]
	int             n_pure = 0;
	// 2D, 64^2 patches: half-size transforms for patches whose plan has two pure axes (kernels2d.hpp k_patch_solve2d_sym):
	DevBuf<double>  mat2sym;         // [plan][4][8][4][64] fragments (zeros for plans with a mixed axis)
	DevBuf<int32_t> ps2_list;        // [patches with two pure axes (n_pure2) | the others]; empty when all or none are pure
	int             n_pure2 = 0;
	// scratch
	std::unique_ptr<te_vec> u, f, r, t;

	Level2D dev2() const
	{
		Level2D L;
		L.P         = P;
		L.n         = n;
		L.face_kind = patch_local ? face_kind_patch.p : face_kind.p;
		L.face_src  = face_src.p;
		L.face_kadj = face_kadj.p;
		L.rh2       = rh2.p;
		L.ghost     = ghostCur();
		return L;
	}
	LevelDev dev() const
	{
		LevelDev L;
		L.P         = P;
		L.face_kind = patch_local ? face_kind_patch.p : face_kind.p;
		L.face_src  = face_src.p;
		L.face_kadj = face_kadj.p;
		L.rh2       = rh2.p;
		L.ghost     = ghostCur();
		L.order     = nullptr;
		L.first     = 0;
		L.count     = P;
		L.xf        = nullptr;
		L.xf_out    = nullptr;
		L.f6        = nullptr;
		L.f6_out    = nullptr;
		L.f6off     = f6Off();
		L.fcorr     = nullptr;
		return L;
	}
	// compact x-face columns of the level's current iterate inside te_vcycle (ping-pong with the sweeps'
	// out-of-place output); xf_valid_for = the data pointer they describe, or null
	DevBuf<double>  geom_starts, geom_h; // [P][3] lower corner and spacings (te_init_problem)
	DevBuf<int32_t> node_ids;            // [P] tree node ids
	DevBuf<double> cellvol;          // [P] product of the spacings (te_integrate)
	std::vector<double> patch_vol;   // [P] product of the patch lengths (te_volume)
	DevBuf<double> f6buf;            // [P][6][n^2]: the six face layers of an iterate that is never stored (opts.fuse = 3)
	// Where the RB-GS kernels keep face layer (p, s) inside f6buf (LevelDev.f6off): the layers that travel to other ranks
	// first, in the order of the level's face exchange, so that the exchange after the pre-sweep sends them from where they
	// are -- no pack kernel. Empty when a layer travels more than once (a refined level cut by rank boundaries: the pack kernel
	// stays). f6_tab: the face layers in f6buf were written through the table (the patch solve writes [p][6]).
	DevBuf<int32_t> f6off;
	bool            f6_tab = false;
	const int32_t  *f6Off() const { return f6_tab ? f6off.p : nullptr; }
	// The coarser level lives on every rank and this level is uniformly refined everywhere (global facts): the parent of a
	// neighbour on another rank is local, so the post-sweep on v + P e forms that neighbour's correction itself
	// (ProlongSrc::gparent) from the face layers of v its ghost slots still hold from the pre-sweep's exchange
	// (ghost_has_v) -- the second face exchange of the level and its pack kernel do not exist.
	bool            post_exchange_free = false, ghost_has_v = false;
	DevBuf<int32_t> slot_parent, slot_orth; // [nremote]
	// Direct-store transport (te_gmg_use_push, pushkernels.hpp). Faces: the slots of neighbours on other ranks exist twice
	// (ghost_buf[0] = ghost, [1] = ghost_alt); exchange number e of the level lands in buffer e & 1 on every rank, and ghostCur()
	// is the buffer the kernels read. A rank can be at most ONE exchange ahead of a peer it trades faces with (it needs that
	// peer's data of exchange e to get past e), so when it writes buffer (e + 1) & 1 there, the peer has long issued -- in
	// stream order behind every reader of that buffer -- its own push e: no credit message needed. push_peer_ghost[b][i]: peer
	// fx.peers[i]'s buffer b, mapped here, already offset to where my range lands.
	DevBuf<double>        ghost_alt;
	int                   ghost_par = 0;
	double               *ghostCur() const { return ghost_par ? ghost_alt.p : ghost.p; }
	bool                  push_faces = false, push_blocks = false;
	std::vector<double *> push_peer_ghost[2];
	uint64_t              face_epoch = 0, blk_epoch = 0;
	// Blocks (repl_direct): the coarse level's right-hand side exists twice as well (cf_buf[0] = the coarse level's f vector's own
	// storage, [1] = cf_alt): gather number e fills buffer e & 1 everywhere -- a rank that trades no faces with me may still be one
	// whole cycle behind, reading the other buffer. push_peer_cf[b][i]: peer tx_direct.peers[i]'s buffer b (offset 0: the runs
	// sit at the same place on every rank).
	DevBuf<double>        cf_alt;
	double               *cf_buf[2] = {nullptr, nullptr};
	std::vector<double *> push_peer_cf[2];
	DevBuf<unsigned>      push_done; // [2] arrival counters of the two push kernels' workgroups
	// pack + push in one launch (PackPush): where face i of the send order goes in its receiver's ghost buffer, per parity, and
	// the flags to raise
	DevBuf<double *>             push_face_dst[2];
	DevBuf<PushFlag>             push_face_flags;
	// an exchange in progress (between pushBegin and pushFinish): its parity, epoch, and what to wait for
	int                push_par = 0;
	unsigned long long push_ep  = 0;
	PushWait           push_wait;
	// repl_up and every rank's patches restrict into whole coarse patches that are a contiguous run of the coarse level:
	// the restricted blocks are exchanged in place (run to run inside the coarse vector), no pack / unpack kernel
	bool   repl_direct = false;
	ExPlan tx_direct;
	// [P][4][n^2]: the x-face ghost terms of this level's right-hand side that the finer level's pre-sweep exported instead of
	// adding them in a fix-up pass (march3d.hpp FCorrSrc); f_has_corr: they belong to the current L.f (inside te_vcycle)
	DevBuf<double> fcorr;
	bool           f_has_corr = false;
	Fold2DHost     fold_pending; // (2D) belongs to the current L.f, inside te_vcycle: see Fold2DHost
	DevBuf<double> rs6; // [P][6][(n/2)^2]: the 2x2 sums of the face layers, as the producer of the next level's fcorr
	DevBuf<int32_t> gtab; // 3D: [Pc][48] block starts in rs6 for k_fcorr_gather3d (built at its first launch; TE_NO_GTAB2)
	DevBuf<GatherDesc> gdesc; // 3D: [Pc][48] k_fcorr_gather3d_v2's descriptors; gdesc_key: what they were built for (rs6 there? f6 through the table?)
	int                gdesc_key = -1;
	const double      *fcorr_zeroed_for = nullptr; // the coarse level's side array this level's gather has zeroed once (its all-zero planes are never written)
	DevBuf<double> e4buf; // 2D: [P][4][n] edge layers of an iterate that is never stored (the 2D twin of f6buf)
	const double  *pack_f6 = nullptr; // set while that iterate is the one whose faces travel to other ranks
	// reference smoother, opts.fuse = 3: the zero-guess pre-sweep is asked to store only the face layers of its result (ps_faces_req,
	// set by the cycle); ps_faces: it did -- f6buf holds them, the level's u is undefined until the post-sweep rewrites it
	bool ps_faces_req = false, ps_faces = false;
	DevBuf<double> xfbuf[2];
	int            xf_cur       = 0;
	const double  *xf_valid_for = nullptr;
	// interior patches (no ghost-slot face) first, then boundary patches
	DevBuf<int32_t> order;
	int             n_int = 0, n_bnd = 0;
	// how a stencil / sweep launch of this level meets its face exchange (withGhosts): 0 the exchange, then one launch over all
	// patches; 1 the exchange on the communication stream, the interior patches under it on the solver stream, then the boundary
	// patches; 2 the interior patches on the second stream, exchange and boundary patches on the solver stream (no hand-over
	// in front of the exchange); -1: by size (TE_OVERLAP_MIN). Set by te_gmg_autotune from measurements on the live communicator.
	int             overlap_mode = -1;
	LevelDev        devPart(bool boundary) const
	{
		LevelDev L = dev();
		L.order    = order.p;
		L.first    = boundary ? n_int : 0;
		L.count    = boundary ? n_bnd : n_int;
		return L;
	}
};

// A vector statement of te_bicgstab whose result is the right-hand side of the next cycle and that has not been executed:
// kind 1: s = resid + ap * (-alpha) (BiCGStab.h:79-80); kind 2: p = beta (p + ap * (-omega)) + resid (:99-100). The first
// kernel of the cycle that reads its right-hand side forms it (march3d.hpp FSrc) -- or, on any other path, the stand-alone
// kernel k_bicg_s / k_bicg_p runs first (visit()).
struct PendingRhs {
	int    kind;
	FSrc   args;
	size_t n2; // double2 elements of the vectors
};
struct EventPair {
	hipEvent_t a, b;
	int        kc;
	bool       valid; // both events recorded in this use
};
} // namespace tei
using namespace tei;

struct te_gmg {
	Cfg                                     cfg; // the TE_* switches, read once in te_gmg_create
	int                                     device = 0;
	hipStream_t                             stream = nullptr;
	// ghost exchanges run on their own stream so that interior patches compute underneath them
	hipStream_t comm_stream = nullptr;
	hipEvent_t  ev_pack = nullptr, ev_recv = nullptr;
	bool        overlap = true;
	bool        in_cycle = false; // te_vcycle in progress: the levels' xf_valid_for bookkeeping is trustworthy
	bool        no_xf_export = false; // the patch solve in progress is the last kernel on its level: nobody reads its x faces
	// te_bicgstab: the cycle's result is the very next operand of an operator application -- level 0's last sweep exports its
	// compact x-face columns after all, and they stay valid when the cycle returns (the stencil kernel then reads 256 contiguous
	// bytes per plane and side instead of 8 of every 128-byte line of the neighbour patch: 1.24 x -> 1.0x of its algorithmic bytes)
	bool        keep_final_xf = false;
	int                                     dim = 3, n = 0;
	std::vector<std::unique_ptr<LevelHost>> levels;
	DevBuf<double>                          partial, result;
	DevBuf<double>                          loopbuf; // TE_RCCL_LOOPBACK (diagnostic): source and sink of the self-addressed messages
	double                                 *result_host = nullptr; // pinned
	int                                     red_blocks  = 1024;
	te_exchange_fn                          exchange    = nullptr;
	void                                   *exchange_user = nullptr;
	int                                     rank = 0, nranks = 1;
	// sum / max of a few host scalars over the ranks (Vector.h:294,306,319 MPI_Allreduce); with the native RCCL
	// back-end the library reduces on the device instead (ncclAllReduce on the solver stream)
	te_allreduce_fn                         allreduce      = nullptr;
	void                                   *allreduce_user = nullptr;
	// schedule check (te_gmg_verify_schedule): exchanges are recorded instead of performed
	te_vec *bicg_work[8] = {nullptr}; // te_bicgstab's work vectors (level 0), allocated at its first call
	const PendingRhs *pending_rhs = nullptr; // set by te_bicgstab around a cycle: level 0's right-hand side is still to be formed
	bool recording = false;
	double bcgs_tol = 1e-12; // BiCGStabSolver(op, tol = 1e-12, max_it = 1000), BiCGStabSolver.h:103-108
	int    bcgs_max_it = 1000;
	bool ps2d_attr = false, ps_lds_ok = false, bcgs_attr = false; // dynamic-LDS attributes of the patch-solve kernels set on this solver's device
	int  ncu = 0;
	struct ExRec {
		int     tag, level, peer;
		int64_t send_cnt, recv_cnt;
	};
	std::vector<ExRec>         record;
	int                        cur_level = 0;
	std::set<uint64_t>         verified_opts;
	// how the hierarchy placed its small levels (te_hier_build: agglomerate, agglomerate_max, replicate) and its depth: every rank
	// must have built the same (checked across the ranks before the first cycle, whatever TE_NO_VERIFY says)
	double      placement[4]      = {0, 0, 0, 0};
	bool        placement_checked = false;
	std::string autotune_report; // what te_gmg_autotune measured and chose
	// watchdog: an exchange that has not completed TE_EXCHANGE_TIMEOUT seconds after it was issued ends the process.
	// Outstanding exchanges sit in a ring in issue order, each with its own event and issue time: the deadline always
	// belongs to the OLDEST one that has not completed (a host that runs ahead of the GPU keeps the newest event
	// incomplete at every poll; that must not age the deadline of exchanges that did complete).
	struct Watchdog {
		static constexpr int RING = 64;
		struct Slot {
			hipEvent_t                            ev = nullptr;
			bool                                  recorded = false; // false: still inside the (possibly blocking) host call
			std::chrono::steady_clock::time_point since;
			int                                   tag = 0, level = 0;
		};
		std::thread       th;
		std::mutex        mu;
		std::atomic<bool> stop{false};
		Slot              slot[RING];
		uint64_t          head = 0, tail = 0; // [head, tail) outstanding
		int64_t           batch = -1;         // >= 0: inside WatchdogBatch, the slot that stands for the whole call
		double            timeout_s = 300.0;
	} wd;
	// optional: RCCL point-to-point called straight from this library (no host callback per exchange)
	struct Rccl {
		void *lib = nullptr, *comm = nullptr;
		int (*GroupStart)()                                                          = nullptr;
		int (*GroupEnd)()                                                            = nullptr;
		int (*Send)(const void *, size_t, int, int, void *, hipStream_t)             = nullptr;
		int (*Recv)(void *, size_t, int, int, void *, hipStream_t)                   = nullptr;
		int (*CommDestroy)(void *)                                                   = nullptr;
		int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
		int (*CommCount)(void *, int *)                                              = nullptr;
		int (*CommUserRank)(void *, int *)                                           = nullptr;
		const char *(*GetErrorString)(int)                                           = nullptr;
	} rccl;
	// direct-store transport (te_gmg_use_push): flags [nranks][2 * levels] in fine-grained device memory (mine: the peers
	// raise them), every peer's table mapped; err: the solver's error word (a wait that gave up), device + pinned host copy
	struct Push {
		bool                              on = false; // the exchanges that have a direct form use it
		bool                              ready = false;
		bool                              rejected = false; // te_gmg_autotune found it unusable on this machine: it stays off
		unsigned long long               *flags = nullptr;
		std::vector<unsigned long long *> peer_flags; // [nranks] (mine at [rank])
		int                              *err = nullptr, *err_host = nullptr;
		std::vector<void *>               opened; // hipIpcOpenMemHandle results, closed in te_gmg_destroy
		unsigned long long               *sent = nullptr; // [nranks][nslot]: the last epoch this rank raised at each peer and slot (PushFlag::sent)
		// a wait's budget: TE_PUSH_TIMEOUT, by default the watchdog's TE_EXCHANGE_TIMEOUT (a legitimate skew between ranks -- I/O,
		// first-use allocation -- must not end a job sooner on this transport than on the other); inside te_gmg_autotune's trial of
		// the transport (`trial`) at most trial_timeout_s: a transport that does not work here is found out within seconds
		double                            timeout_s = 300.0, trial_timeout_s = 5.0;
		bool                              trial = false;
		int                               nslot = 0;
		std::atomic<bool>                 fatal{true}; // a wait that gave up ends the process (watchdog); false inside te_gmg_autotune's trial
	} push;
#if TE_STAMPS
	// diagnostic build (kernels3d.hpp Stamps): stamps of up to MAXL launches of at most MAXWG workgroups each
	struct StampHost {
		static constexpr int            MAXL = 64, MAXWG = 1024;
		DevBuf<unsigned long long>      buf;
		bool                            on = false;
		std::vector<std::string>        names;
		std::vector<int>                wgs;
	} stamps;
#endif
	double setup_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // te_gmg_setup_ms
	// profiling
	bool                   profiling = false;
	int                    prof_only = -1; // >= 0: only this kernel class is timed
	int                    prof_stride = 1; // > 1: only every prof_stride-th launch of a timed class carries events (te_gmg_profile_stride)
	uint32_t               prof_seq[KC_COUNT] = {}; // launches of each class seen since the stride was set
	std::vector<EventPair> ev_pool;
	size_t                 ev_used = 0;
	int64_t                calls[KC_COUNT];
	int64_t                cells[KC_COUNT]; // lattice sites processed
	double                 total_ms[KC_COUNT];
};

namespace tei
{
struct Timed {
	te_gmg *g;
	int     idx = -1;
	bool    ext, first = true;
	// ext: the launches of this scope go through launchT, which hands the events to the dispatch itself (hipExtLaunchKernelGGL:
	// time stamps of the kernel's own start and end, as rocprofv3 sees it) -- an event recorded on the stream before and after a
	// launch costs a barrier packet each, several microseconds around a kernel of tens
	Timed(te_gmg *g_, int kc, size_t ncells = 0, bool ext_ = false) : g(g_), ext(ext_)
	{
		if (!g->profiling || (g->prof_only >= 0 && g->prof_only != kc)) return;
		if (g->prof_stride > 1 && (g->prof_seq[kc]++ % (uint32_t) g->prof_stride) != 0) return;
		g->cells[kc] += (int64_t) ncells;
		if (g->ev_used == g->ev_pool.size()) {
			EventPair e;
			if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
			g->ev_pool.push_back(e);
		}
		idx                   = (int) g->ev_used++;
		g->ev_pool[idx].kc    = kc;
		g->ev_pool[idx].valid = false;
		if (!ext) (void) hipEventRecord(g->ev_pool[idx].a, g->stream);
	}
	~Timed()
	{
		if (idx >= 0 && !ext) {
			(void) hipEventRecord(g->ev_pool[idx].b, g->stream);
			g->ev_pool[idx].valid = true;
		}
	}
};

// a kernel launch inside an ext scope: the first one carries the start event, every one the stop event (the last record counts)
template <typename K, typename... A> void launchT(Timed &t, K kern, dim3 grid, dim3 blk, size_t shm, hipStream_t s, A... args)
{
	if (t.idx >= 0 && t.ext) {
		EventPair &e = t.g->ev_pool[t.idx];
		hipExtLaunchKernelGGL(kern, grid, blk, shm, s, t.first ? e.a : (hipEvent_t) nullptr, e.b, 0, args...);
		t.first = false;
		e.valid = true;
	} else {
		hipLaunchKernelGGL(kern, grid, blk, shm, s, args...);
	}
}

#if TE_STAMPS
// where the next instrumented launch (`wgs` workgroups) keeps its stamps; null when nobody is collecting
inline StampDst stampNext(te_gmg *g, const char *name, int wgs)
{
	StampDst d;
	auto    &S = g->stamps;
	if (!S.on || !S.buf.p || (int) S.names.size() >= S.MAXL || wgs > S.MAXWG || g->recording) return d;
	d.p = S.buf.p + (size_t) S.names.size() * S.MAXWG * TE_NSTAMP;
	S.names.push_back(std::string(name) + " L" + std::to_string(g->cur_level));
	S.wgs.push_back(wgs);
	return d;
}
inline LevelDev stamped(te_gmg *g, LevelDev D, const char *name, int wgs)
{
	D.stamp_dst = stampNext(g, name, wgs);
	return D;
}
#define TE_STAMP_ARG(g, name, wgs) , stampNext(g, name, wgs)
#else
inline const LevelDev &stamped(te_gmg *, const LevelDev &D, const char *, int) { return D; }
#define TE_STAMP_ARG(g, name, wgs)
#endif

inline int gridFor(size_t work_items, int tpb, int cap = 4096)
{
	size_t b = (work_items + tpb - 1) / tpb;
	if (b < 1) b = 1;
	if (b > (size_t) cap) b = cap;
	return (int) b;
}

inline bool sameShape(const te_vec *a, const te_vec *b) { return a && b && a->g == b->g && a->level == b->level; }

// ---- gmg_core.hip
int  newVec(te_gmg *g, int level, te_vec **out);
void drainEvents(te_gmg *g);

// ---- gmg_transport.hip
void watchdogRetire(te_gmg::Watchdog &w);
void watchdogMakeRoom(te_gmg::Watchdog &w, std::unique_lock<std::mutex> &lk);
void watchdogStart(te_gmg *g);
void watchdogStop(te_gmg *g);
int  doExchange(te_gmg *g, int tag, const ExPlan &pl, const double *send, double *recv, hipStream_t stream = nullptr);
int  finishReduce(te_gmg *g, int n, int op, bool global);
void pushBegin(te_gmg *g, LevelHost &L, int kind);
int  pushFinish(te_gmg *g, LevelHost &L, int kind, hipStream_t stream);
int  pushExchange(te_gmg *g, LevelHost &L, int kind, const double *send, hipStream_t stream = nullptr);
int  faceExchange(te_gmg *g, LevelHost &L, const double *send, hipStream_t stream = nullptr);
void pushTeardown(te_gmg *g, bool final = false); // frees what te_gmg_use_push's set-up allocated and mapped (a failed set-up; final: te_gmg_destroy)
// what a push stores in the peers' flags for epoch `ep`: `ep`, unless TE_PUSH_FAULT (diagnostic) asks for a transport whose data
// never "arrives" (any value but "overrun": the PREVIOUS epoch, so that the give-up / rejection path can be tested) or for a peer
// that claims to be two exchanges ahead ("overrun": ep + 2, the receiver's check PUSH_ERR_OVERRUN)
inline unsigned long long pushRaiseValue(const te_gmg *g, unsigned long long ep)
{
	if (!g->cfg.has(O_PUSH_FAULT) || !strcmp(g->cfg.str(O_PUSH_FAULT), "nonce") || !strncmp(g->cfg.str(O_PUSH_FAULT), "setup:", 6)) return ep;
	return !strcmp(g->cfg.str(O_PUSH_FAULT), "overrun") ? ep + 2 : ep - 1;
}

struct WatchdogArm { // around the issue of one exchange: takes a ring slot (issue time now), records its event behind the exchange
	te_gmg     *g;
	hipStream_t stream;
	int64_t     idx = -1;
	WatchdogArm(te_gmg *g_, hipStream_t st, int tag) : g(g_), stream(st)
	{
		auto &w = g->wd;
		if (!w.th.joinable()) return;
		std::unique_lock<std::mutex> lk(w.mu);
		if (w.batch >= 0) { // inside a V-cycle or a Krylov solve: the call's one slot stands for this exchange too
			auto &sl = w.slot[w.batch % te_gmg::Watchdog::RING];
			sl.since = std::chrono::steady_clock::now(); // (the host got this far: the deadline runs from the newest issue)
			sl.tag   = tag;
			sl.level = g->cur_level;
			return;
		}
		watchdogMakeRoom(w, lk); // (ring full: the host waits for the oldest exchange instead of dropping a watched one)
		idx      = (int64_t) w.tail++;
		auto &sl = w.slot[idx % te_gmg::Watchdog::RING];
		sl.since = std::chrono::steady_clock::now();
		sl.tag   = tag;
		sl.level = g->cur_level;
		sl.recorded = false; // a blocking host callback is covered too: no event yet, only the deadline
	}
	~WatchdogArm()
	{
		auto &w = g->wd;
		if (idx < 0) return;
		std::lock_guard<std::mutex> lk(w.mu);
		auto &sl    = w.slot[idx % te_gmg::Watchdog::RING];
		sl.recorded = (hipEventRecord(sl.ev, stream) == hipSuccess);
	}
};

// Around one call that issues several exchanges (a V-cycle, a Krylov solve): ONE ring slot and ONE event, recorded on the solver
// stream when the call has enqueued everything -- an event behind every exchange costs about 5 us of stream time each (the
// kernel behind it waits for the marker to retire: 27 us of a 428 us cycle at eight ranks). What is watched does not change:
// the slot's deadline restarts whenever the host issues the next exchange of the call (a host that still issues is not stuck;
// one that blocks in a callback or a synchronisation stops issuing), and the event at the end cannot complete before every
// exchange of the call has -- a peer that never posts its half is found TE_EXCHANGE_TIMEOUT after the last issue, as before.
struct WatchdogBatch {
	te_gmg *g;
	int64_t idx = -1;
	explicit WatchdogBatch(te_gmg *g_) : g(g_)
	{
		auto &w = g->wd;
		if (!w.th.joinable()) return;
		std::unique_lock<std::mutex> lk(w.mu);
		if (w.batch >= 0) return; // (nested: the outer call's slot)
		watchdogMakeRoom(w, lk);
		idx      = (int64_t) w.tail++;
		auto &sl = w.slot[idx % te_gmg::Watchdog::RING];
		sl.since = std::chrono::steady_clock::now();
		sl.tag = 0, sl.level = 0;
		sl.recorded = false;
		w.batch     = idx;
	}
	~WatchdogBatch()
	{
		if (idx < 0) return;
		auto &w = g->wd;
		std::lock_guard<std::mutex> lk(w.mu);
		w.batch     = -1;
		auto &sl    = w.slot[idx % te_gmg::Watchdog::RING];
		sl.recorded = (hipEventRecord(sl.ev, g->stream) == hipSuccess);
	}
};

// ---- gmg_launch3d.hip (the dispatchers hand 2D levels to gmg_launch2d.hip)
template <int MODE> int launchStencil(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, double omega,
                                      RestrictDst rd = RestrictDst(), const double *xf_in = nullptr, int redmode = RED_NONE,
                                      const double *red_a = nullptr, int *red_items = nullptr);
extern template int launchStencil<MODE_APPLY>(te_gmg *, LevelHost &, const double *, const double *, double *, double, RestrictDst, const double *, int, const double *, int *);
extern template int launchStencil<MODE_RESID>(te_gmg *, LevelHost &, const double *, const double *, double *, double, RestrictDst, const double *, int, const double *, int *);
extern template int launchStencil<MODE_JACOBI>(te_gmg *, LevelHost &, const double *, const double *, double *, double, RestrictDst, const double *, int, const double *, int *);
int resweepProlong(te_gmg *g, LevelHost &L, const double *f, double *out, const double *prolong_from, double *xf_out, const double *fcorr_in);
int interfaceResidRestrict(te_gmg *g, LevelHost &L, const double *u, const double *xf, double *coarse, size_t coarse_n);
int zeroSweepResid(te_gmg *g, LevelHost &L, const double *f, double *out, double *coarse, double *xf_out, bool store_u,
                   double *fcorr_out = nullptr, const double *fcorr_in = nullptr, const PendingRhs *fs = nullptr,
                   const Fold2DHost *fold_in = nullptr, bool skip_fixup = false);
int launchRbgs(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, bool zero_guess = false,
               const double *prolong_from = nullptr, const double *xf_in = nullptr, double *xf_out = nullptr);
int residRestrict(te_gmg *g, LevelHost &L, const double *u, const double *f, double *coarse, const double *xf_in = nullptr);
int patchSolve(te_gmg *g, LevelHost &L, const double *f, double *u, bool zero_guess = false, const double *prolong_from = nullptr,
               bool *swapped = nullptr);
int doRestrict(te_gmg *g, int fine_level, const double *fine, double *coarse);
int doProlong(te_gmg *g, int fine_level, const double *coarse, double *fine);
// ---- gmg_launch2d.hip
template <int MODE> int launchStencil2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, double omega, int redmode = RED_NONE,
                                        const double *red_a = nullptr, int *red_items = nullptr);
extern template int launchStencil2d<MODE_APPLY>(te_gmg *, LevelHost &, const double *, const double *, double *, double, int, const double *, int *);
extern template int launchStencil2d<MODE_RESID>(te_gmg *, LevelHost &, const double *, const double *, double *, double, int, const double *, int *);
extern template int launchStencil2d<MODE_JACOBI>(te_gmg *, LevelHost &, const double *, const double *, double *, double, int, const double *, int *);
int residualSumsq2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, int *blocks);
int launchRbgs2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, bool zero_guess = false, const double *prolong_from = nullptr);
int residRestrict2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *coarse);
int patchSolve2d(te_gmg *g, LevelHost &L, const double *f, double *u, double *s0, double *s1, bool zero_guess, bool *swapped,
                 const double *prolong_from = nullptr);
bool ps2dResidFusable(const te_gmg *g, const LevelHost &L);    // (global facts) the 2D block-Jacobi cycle takes its residual on the patch edges
bool patchSolve2dFusable(const te_gmg *g, const LevelHost &L); // (rank-local) its post-sweep adds the prolongation itself
int  interfaceResidRestrict2d(te_gmg *g, LevelHost &L, const double *u, double *coarse, size_t coarse_n);
int patchBcgs2d(te_gmg *g, LevelHost &L, const double *f, double *u, double *s0);
int restrict2d(te_gmg *g, LevelHost &L, const double *fine, double *coarse);
int prolong2d(te_gmg *g, LevelHost &L, const double *coarse, double *fine);
int zeroSweepResid2d(te_gmg *g, LevelHost &L, const double *f, double *out, double *coarse, bool store_u, const Fold2DHost *fold_in = nullptr,
                     bool skip_fixup = false);
int resweepProlong2d(te_gmg *g, LevelHost &L, const double *f, double *out, const double *prolong_from);
// ---- gmg_cycle.hip
int visit(te_gmg *g, const te_cycle_opts *o, int l, const te_vec *f, te_vec *u, bool u_zero);
int vcycleWith(te_gmg *g, const te_cycle_opts *o, const te_vec *f, te_vec *u, const PendingRhs *pending); // te_vcycle; `pending`: see PendingRhs

// x-face columns of `d`, if the level still holds them (RB-GS sweeps and the single-pass patch solve produce them, inside te_vcycle)
inline const double *xfFor(LevelHost &L, const double *d) { return (d && L.xf_valid_for == d) ? L.xfbuf[L.xf_cur].p : nullptr; }

// the sweep wrote `out` together with its x-face columns into the other buffer: make them current
inline void xfProduced(LevelHost &L, const double *out)
{
	L.xf_cur ^= 1;
	L.xf_valid_for = out;
}

template <int OP> int vecop(te_vec *v, const te_vec *a, const te_vec *b, double alpha, double beta, double gamma)
{
	if (!v || (OP >= VOP_COPY && !sameShape(v, a))
	    || ((OP == VOP_ADD_SCALED2 || OP == VOP_SCALE_THEN_ADD_SCALED2) && !sameShape(v, b)))
		return te::fail(TE_EINVAL, "te_vec_*: vectors of different levels");
	if (v->n == 0) return TE_OK;
	te_gmg *g = v->g;
	if (g->levels[v->level]->xf_valid_for == v->d) g->levels[v->level]->xf_valid_for = nullptr; // v changes in place
	Timed   t(g, KC_VECOP, v->n);
	// one 16-B element per thread: on this chip a flat grid in address order streams 25-40 % faster than a
	// capped grid-stride loop (tools/membw.hip: fill 6.9 vs 4.9 TB/s, triad 6.0-6.5 vs 4.9 TB/s)
	hipLaunchKernelGGL(k_vecop<OP>, dim3(gridFor(v->n / 2, 256, 1 << 30)), dim3(256), 0, g->stream, v->n / 2,
	                   reinterpret_cast<double2 *>(v->d), a ? reinterpret_cast<const double2 *>(a->d) : nullptr,
	                   b ? reinterpret_cast<const double2 *>(b->d) : nullptr, alpha, beta, gamma);
	HIPCHK(hipGetLastError());
	return TE_OK;
}

template <int OP> int reduce(const te_vec *a, const te_vec *b, double *out, bool global = false)
{
	if (!a || !out || (OP == RED_DOT && !sameShape(a, b))) return te::fail(TE_EINVAL, "te_vec reduce: bad argument");
	te_gmg *g = a->g;
	if (a->n == 0 && !(global && g->nranks > 1)) {
		*out = 0.0;
		return TE_OK;
	}
	if (OP != RED_MAXABS && !global && g->rank != 0 && g->levels[a->level]->replicated) { // a level on every rank counts once (rank 0's)
		*out = 0.0;
		return TE_OK;
	}
	if (a->n == 0) { // a rank without patches still takes part in the reduction over ranks
		HIPCHK(hipMemsetAsync(g->result.p, 0, sizeof(double), g->stream));
		int rc0 = finishReduce(g, 1, OP == RED_MAXABS ? 1 : 0, true);
		*out = g->result_host[0];
		return rc0;
	}
	const int blocks = gridFor(a->n / 2, 256, g->red_blocks);
	{
		Timed t(g, KC_REDUCE, a->n);
		hipLaunchKernelGGL(k_reduce<OP>, dim3(blocks), dim3(256), 0, g->stream, a->n / 2,
		                   reinterpret_cast<const double2 *>(a->d),
		                   b ? reinterpret_cast<const double2 *>(b->d) : nullptr, g->partial.p);
		hipLaunchKernelGGL(k_reduce_final<OP>, dim3(1), dim3(256), 0, g->stream, blocks, g->partial.p, g->result.p);
	}
	int rc = finishReduce(g, 1, OP == RED_MAXABS ? 1 : 0, global);
	if (rc) return rc;
	*out = g->result_host[0];
	return TE_OK;
}

inline void swapData(te_vec *a, te_vec *b) { std::swap(a->d, b->d); }

} // namespace tei

// Nothing may unwind into a C caller (ctypes, the reference's C++ built with other flags): every int-returning entry
// point below runs inside this barrier. std::bad_alloc and friends come from the std::vector / std::map set-up code.
template <class F> static inline int guarded(F body) noexcept
{
	try {
		return body();
	} catch (const std::bad_alloc &) {
		return te::fail(TE_ENOMEM, "out of host memory");
	} catch (const std::exception &e) {
		return te::fail(TE_ESTATE, std::string("unexpected exception: ") + e.what());
	} catch (...) {
		return te::fail(TE_ESTATE, "unexpected exception");
	}
}

static inline int checkLevelVec(te_gmg *g, int level, const te_vec *v, const char *who)
{
	if (!g || !v || level < 0 || level >= (int) g->levels.size() || v->g != g || v->level != level)
		return te::fail(TE_EINVAL, std::string(who) + ": vector does not belong to this level");
	return TE_OK;
}
