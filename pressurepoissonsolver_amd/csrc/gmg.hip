// Device half of include/te_hip.h: level stack on the GPU, kernel launches, V/W cycle,
// BiCGStab. Host C++ + HIP for gfx950 only. There is no CPU fallback anywhere in this file:
// if HIP cannot give us a device, te_gmg_create fails with TE_EHIP.
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
		if (_e != hipSuccess)                                                                     \
			return te::fail(TE_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
	} while (0)

struct te_vec {
	te_gmg *g     = nullptr;
	int     level = 0;
	double *d     = nullptr;
	size_t  n     = 0;
};

namespace
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
	KC_BICG_UPDATE, KC_BICG_S, KC_BICG_P, KC_APPLY_DOT, KC_COUNT
};
const char *kclassName[KC_COUNT] = {"stencil_apply", "stencil_resid", "stencil_jacobi", "stencil_rbgs",
                                    "cf_ghost", "restrict", "prolong_add", "patch_rhs", "dst_axis",
                                    "vecop", "reduce", "pack", "exchange", "stencil_rbgs_zero", "resid_restrict", "patch_solve_mfma", "stencil_rbgs_prolong",
                                    "stencil_rbgs_slabs", "stencil_slabs", "patch_solve_3pass", "rbgs_zero_resid_restrict",
                                    "restrict_fixup", "rbgs_resweep_prolong", "rbgs_zero_resid_restrict_faces",
                                    "rbgs_resweep_prolong_fcorr", "rbgs_zero_resid_restrict_faces_fcorr", "fcorr_gather", "patch_solve_mfma_faces",
                                    "bicg_update", "bicg_s", "bicg_p", "stencil_apply_dot"};

// Every TE_* switch of this library. They are read from the environment ONCE, in te_gmg_create;
// te_gmg_set_option changes one afterwards (the tests pin one implementation against another that way). Nothing on a
// launch path looks at the environment.
enum Opt : int {
	O_2D_SIMPLE, O_2D_NO_MFMA, O_2D_NO_PF, O_2D_NO_MR_FUSE, O_2D_TPB, O_NO_FUSE2, O_NO_FUSE3, O_NO_FUSE3_CF, O_NO_CFP, O_NO_XF,
	O_NO_FCORR, O_NO_FCORR_CF, O_NO_GTAB, O_NO_OVERLAP, O_OVERLAP_MIN, O_NO_PS_FACES, O_PS_MODE, O_PS_SLOW, O_RBGS_NOSLAB,
	O_ZS_FORCE, O_NO_ZS8, O_RESWEEP_V, O_EXCHANGE_TIMEOUT, O_NO_VERIFY, O_NO_GRAPH, O_RCCL_LOOPBACK, O_ZR_AHEAD, O_NO_BICG_FUSE, O_POST_EXCHANGE, O_REPL_BLOCKS, O_PACK_FACES, O_OVERLAP_MODE, O_PUSH_TIMEOUT, O_NO_BICG_XF, O_COUNT
};
const char *optName[O_COUNT] = {"TE_2D_SIMPLE", "TE_2D_NO_MFMA", "TE_2D_NO_PF", "TE_2D_NO_MR_FUSE", "TE_2D_TPB", "TE_NO_FUSE2", "TE_NO_FUSE3",
                                "TE_NO_FUSE3_CF", "TE_NO_CFP", "TE_NO_XF", "TE_NO_FCORR", "TE_NO_FCORR_CF", "TE_NO_GTAB", "TE_NO_OVERLAP",
                                "TE_OVERLAP_MIN", "TE_NO_PS_FACES", "TE_PS_MODE", "TE_PS_SLOW", "TE_RBGS_NOSLAB", "TE_ZS_FORCE", "TE_NO_ZS8",
                                "TE_RESWEEP_V", "TE_EXCHANGE_TIMEOUT", "TE_NO_VERIFY", "TE_NO_GRAPH", "TE_RCCL_LOOPBACK", "TE_ZR_AHEAD", "TE_NO_BICG_FUSE", "TE_POST_EXCHANGE", "TE_REPL_BLOCKS", "TE_PACK_FACES", "TE_OVERLAP_MODE", "TE_PUSH_TIMEOUT", "TE_NO_BICG_XF"};
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
	void fromEnv()
	{
		for (int o = 0; o < O_COUNT; o++) set(o, getenv(optName[o]));
	}
	bool        has(int o) const { return on[o]; }
	const char *str(int o) const { return on[o] ? val[o].c_str() : nullptr; }
	int         num(int o, int dflt) const { return on[o] ? atoi(val[o].c_str()) : dflt; }
	double      real(int o, double dflt) const { return on[o] ? atof(val[o].c_str()) : dflt; }
};

template <typename T> struct DevBuf {
	T     *p = nullptr;
	size_t n = 0;
	~DevBuf()
	{
		if (p) (void) hipFree(p);
	}
	int alloc(size_t count)
	{
		n = count;
		if (count == 0) return TE_OK;
		HIPCHK(hipMalloc(&p, sizeof(T) * count));
		return TE_OK;
	}
	int upload(const std::vector<T> &h)
	{
		int rc = alloc(h.size());
		if (rc) return rc;
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
	DevBuf<double>  mats, lam, corr; // corr: [P][6][n^2] interface terms of the patch right-hand sides
	DevBuf<double>  matsT;           // 2D: the transform matrices transposed (k_patch_solve2d_lds)
	DevBuf<double>  matsym;          // half matrices in MFMA fragment order (patchsolve32_sym.hpp), 32^3 patches
	bool            sym_ok = false;  // every plan of the level has pure (DST-II/III or DCT-II/III) axes
	DevBuf<int32_t> ps_list;         // otherwise: [patches with pure axes (n_pure) | the others]
	int             n_pure = 0;
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
	DevBuf<unsigned long long *> push_face_flags;
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
	DevBuf<double> rs6; // [P][6][(n/2)^2]: the 2x2 sums of the face layers, as the producer of the next level's fcorr
	DevBuf<int32_t> gtab; // 3D: [Pc][48] block starts in rs6 for k_fcorr_gather3d (built at its first launch)
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
} // namespace

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
	bool ps2d_attr = false, ps_lds_ok = false; // dynamic-LDS attributes of the patch-solve kernels set on this solver's device
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
		unsigned long long               *flags = nullptr;
		std::vector<unsigned long long *> peer_flags; // [nranks] (mine at [rank])
		int                              *err = nullptr, *err_host = nullptr;
		std::vector<void *>               opened; // hipIpcOpenMemHandle results, closed in te_gmg_destroy
		double                            timeout_s = 20.0;
		int                               nslot = 0;
		std::atomic<bool>                 fatal{true}; // a wait that gave up ends the process (watchdog); false inside te_gmg_autotune's trial
	} push;
	// profiling
	bool                   profiling = false;
	int                    prof_only = -1; // >= 0: only this kernel class is timed
	std::vector<EventPair> ev_pool;
	size_t                 ev_used = 0;
	int64_t                calls[KC_COUNT];
	int64_t                cells[KC_COUNT]; // lattice sites processed
	double                 total_ms[KC_COUNT];
};

namespace
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
void drainEvents(te_gmg *g)
{
	if (g->ev_used == 0) return;
	(void) hipStreamSynchronize(g->stream);
	for (size_t i = 0; i < g->ev_used; i++) {
		float ms = 0;
		if (g->ev_pool[i].valid && hipEventElapsedTime(&ms, g->ev_pool[i].a, g->ev_pool[i].b) == hipSuccess) {
			g->calls[g->ev_pool[i].kc]++;
			g->total_ms[g->ev_pool[i].kc] += ms;
		}
	}
	g->ev_used = 0;
}

inline int gridFor(size_t work_items, int tpb, int cap = 4096)
{
	size_t b = (work_items + tpb - 1) / tpb;
	if (b < 1) b = 1;
	if (b > (size_t) cap) b = cap;
	return (int) b;
}

// DftPatchSolver.h:237-289 (row-major: y_i = sum_j M[i*n+j] x_j)
void transformMatrix(int type, int n, double *m)
{
	for (int i = 0; i < n * n; i++) m[i] = 0.0;
	switch (type) {
		case 0: // DCT-II
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++) m[i * n + j] = cos(M_PI / n * (i * (j + 0.5)));
			break;
		case 1: // DCT-III
			for (int i = 0; i < n; i++) {
				m[i * n] = 0.5;
				for (int j = 1; j < n; j++) m[i * n + j] = cos(M_PI / n * ((i + 0.5) * j));
			}
			break;
		case 2: // DCT-IV
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++) m[i * n + j] = cos(M_PI / n * ((i + 0.5) * (j + 0.5)));
			break;
		case 3: // DST-II
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++) m[i * n + j] = sin(M_PI / n * ((i + 1) * (j + 0.5)));
			break;
		case 4: // DST-III
			for (int i = 0; i < n; i++) {
				for (int j = 0; j < n - 1; j++) m[i * n + j] = sin(M_PI / n * ((i + 0.5) * (j + 1)));
				m[i * n + n - 1] = (i & 1) ? -0.5 : 0.5;
			}
			break;
		default: // DST-IV
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++) m[i * n + j] = sin(M_PI / n * ((i + 0.5) * (j + 0.5)));
			break;
	}
}

// per-peer ranges of a list of (peer, count) items that is already sorted by peer
void rangesByPeer(const std::vector<std::pair<int, int64_t>> &items, std::vector<int32_t> &peers,
                  std::vector<int64_t> &off, std::vector<int64_t> &cnt)
{
	int64_t pos = 0;
	for (auto &it : items) {
		if (peers.empty() || peers.back() != it.first) {
			peers.push_back(it.first);
			off.push_back(pos);
			cnt.push_back(0);
		}
		cnt.back() += it.second;
		pos += it.second;
	}
}
// merge the send-side and receive-side peer lists of one exchange into one ExPlan
ExPlan mergePlan(const std::vector<std::pair<int, int64_t>> &sends, const std::vector<std::pair<int, int64_t>> &recvs)
{
	std::vector<int32_t> sp, rp;
	std::vector<int64_t> so, sc, ro, rc;
	rangesByPeer(sends, sp, so, sc);
	rangesByPeer(recvs, rp, ro, rc);
	std::map<int, std::array<int64_t, 4>> m;
	for (size_t i = 0; i < sp.size(); i++) m[sp[i]] = {so[i], sc[i], 0, 0};
	for (size_t i = 0; i < rp.size(); i++) {
		auto &e = m[rp[i]];
		e[2]    = ro[i];
		e[3]    = rc[i];
	}
	ExPlan pl;
	for (auto &kv : m) {
		pl.peers.push_back(kv.first);
		pl.send_off.push_back(kv.second[0]);
		pl.send_cnt.push_back(kv.second[1]);
		pl.recv_off.push_back(kv.second[2]);
		pl.recv_cnt.push_back(kv.second[3]);
	}
	return pl;
}

int buildLevel(te_gmg *g, const Hierarchy &H, int li)
{
	const Level &lv = H.levels[li];
	const int n = lv.n, D = lv.dim;
	if (D == 3 && n != 4 && n != 8 && n != 16 && n != 32)
		return te::fail(TE_EUNSUPPORTED, "te_gmg_create: 3D patches must have n = 4, 8, 16 or 32 cells per axis");
	if (D == 2 && (n < 4 || (n & 1))) return te::fail(TE_EUNSUPPORTED, "te_gmg_create: 2D patches need an even n >= 4");
	auto L = std::make_unique<LevelHost>();
	L->dim = D;
	L->n   = n;
	L->P   = lv.P;
	L->P_global = lv.P_global;
	L->index    = li;
	L->replicated = lv.replicated;
	L->gathered = lv.replicated || (H.nranks > 1 && std::all_of(lv.g_rank.begin(), lv.g_rank.end(), [&](int32_t r) { return r == lv.g_rank[0]; }));
	L->nc  = (D == 3) ? (size_t) n * n * n : (size_t) n * n;
	L->nf  = (D == 3) ? (size_t) n * n : (size_t) n;
	const int P = lv.P, NS = 2 * D, NCH = 1 << D, NQ = 1 << (D - 1), me = H.rank;

	// ---- remote same-level faces: canonical order = (peer, receiving patch (global), receiving side),
	// which both ends can compute from the global tables
	// A receiving (patch, side, q) gets ONE ghost slot holding the sender's facing layer: the ghost values
	// themselves on a same-level face, raw neighbour cells for k_cf_ghost on a coarse/fine face (q = which
	// of the finer neighbours; the coarse side of a coarse/fine face receives one slot per fine neighbour).
	struct RFace {
		int peer, key_patch, key_side, key_q, p, s, nb;
		bool operator<(const RFace &o) const
		{
			return std::tie(peer, key_patch, key_side, key_q) < std::tie(o.peer, o.key_patch, o.key_side, o.key_q);
		}
	};
	std::vector<RFace> recvs, sends;
	for (int p = 0; p < P; p++) {
		const int gp = lv.l2g[p];
		for (int s = 0; s < NS; s++) {
			const size_t gf   = (size_t) gp * NS + s;
			const int    kind = lv.g_nbr_kind[gf];
			if (kind == NBR_NONE) continue;
			for (int q = 0; q < NQ; q++) {
				const int nb = lv.g_nbr[gf * 4 + q];
				if (nb < 0 || lv.g_rank[nb] == me) continue;
				recvs.push_back({lv.g_rank[nb], gp, s, q, p, s, nb});
				// what the neighbour files my layer under: its own (patch, side) and, when it is the coarse
				// side, my position among its fine neighbours = my quadrant on its face
				const int their_q = (kind == NBR_COARSE) ? lv.g_nbr_orth[gf] : 0;
				sends.push_back({lv.g_rank[nb], nb, s ^ 1, their_q, p, s, nb});
			}
		}
	}
	std::sort(recvs.begin(), recvs.end());
	std::sort(sends.begin(), sends.end());
	const int                           nremote = (int) recvs.size();
	std::map<std::tuple<int, int, int>, int> remote_slot; // (p, s, q) -> ghost slot
	for (int i = 0; i < nremote; i++) remote_slot[std::make_tuple(recvs[i].p, recvs[i].s, recvs[i].key_q)] = i;
	{
		std::vector<std::pair<int, int64_t>> si, ri;
		std::vector<int32_t>                 sf;
		for (auto &f : sends) {
			si.emplace_back(f.peer, (int64_t) L->nf);
			sf.push_back(f.p);
			sf.push_back(f.s);
		}
		for (auto &f : recvs) ri.emplace_back(f.peer, (int64_t) L->nf);
		L->fx      = mergePlan(si, ri);
		L->nremote = nremote;
		int rc0;
		if ((rc0 = L->send_faces.upload(sf)) || (rc0 = L->sendbuf.alloc((size_t) std::max(nremote, 1) * L->nf))) return rc0;
		if (D == 3 && nremote > 0) { // the place of every face layer in f6buf: sent layers first, in send order (see LevelHost::f6off)
			std::vector<int32_t> off((size_t) P * NS, -1);
			bool                 once = true;
			for (size_t i = 0; i < sends.size() && once; i++) {
				int32_t &o = off[(size_t) sends[i].p * NS + sends[i].s];
				once       = (o < 0);
				o          = (int32_t) i;
			}
			if (once) {
				int32_t next = (int32_t) sends.size();
				for (auto &o : off)
					if (o < 0) o = next++;
				if ((rc0 = L->f6off.upload(off))) return rc0;
			}
		}
	}

	std::vector<int32_t> fk(P * NS), fs(P * NS, -1), cfd, cfs, plan(P, 0);
	std::vector<double>  kadj(P * NS, 0.0), rh2(P * 3), cellvol(P);
	L->patch_vol.assign(P, 0.0);
	std::map<int, int>   plan_of_key;
	std::vector<int>     keys;
	int                  nslots = nremote;
	for (int p = 0; p < P; p++) {
		const int gp  = lv.l2g[p];
		int       key = 0;
		rh2[p * 3 + 2] = 0.0;
		double cv = 1.0, pv = 1.0; // Domain.h:270-272 (patch_sum *= spacings[i]), :242-245
		for (int a = 0; a < D; a++) {
			double h       = lv.g_lengths[(size_t) gp * D + a] / n;
			rh2[p * 3 + a] = 1.0 / (h * h);
			cv *= h;
			pv *= h * n;
		}
		cellvol[p]      = cv;
		L->patch_vol[p] = pv;
		for (int s = 0; s < NS; s++) {
			const size_t gf   = (size_t) gp * NS + s;
			const int    kind = lv.g_nbr_kind[gf];
			if (kind == NBR_NONE) {
				fk[p * NS + s]   = H.neumann ? FACE_NEUMANN : FACE_DIRICHLET;
				kadj[p * NS + s] = H.neumann ? -1.0 : 1.0;
				if (H.neumann) key |= 1 << s;
			} else if (kind == NBR_NORMAL) {
				const int nb = lv.g_nbr[gf * 4];
				if (lv.g_rank[nb] == me) {
					fk[p * NS + s] = FACE_LOCAL;
					fs[p * NS + s] = lv.g_local[nb];
				} else { // the neighbour's face cells arrive in a ghost slot; diagonal unchanged
					fk[p * NS + s] = FACE_GHOST;
					fs[p * NS + s] = remote_slot.at(std::make_tuple(p, s, 0));
				}
			} else {
				fk[p * NS + s]   = FACE_GHOST;
				fs[p * NS + s]   = nslots;
				kadj[p * NS + s] = (kind == NBR_COARSE) ? (D == 3 ? -5.0 / 6.0 : -2.0 / 3.0) : 1.0 / 3.0;
				cfd.push_back(p);
				cfd.push_back(s);
				cfd.push_back(kind);
				cfd.push_back(lv.g_nbr_orth[gf]);
				for (int q = 0; q < 4; q++) { // local patch index, or -(slot+2) of the raw layer received for it
					int nb = (q < NQ) ? lv.g_nbr[gf * 4 + q] : -1;
					if (nb >= 0 && lv.g_rank[nb] != me)
						cfd.push_back(-(remote_slot.at(std::make_tuple(p, s, q)) + 2));
					else
						cfd.push_back(nb >= 0 ? lv.g_local[nb] : -1);
				}
				cfs.push_back(nslots);
				nslots++;
			}
		}
		auto it = plan_of_key.find(key);
		if (it == plan_of_key.end()) {
			plan_of_key[key] = (int) keys.size();
			plan[p]          = (int) keys.size();
			keys.push_back(key);
		} else {
			plan[p] = it->second;
		}
	}
	L->nslots = nslots;
	L->lds2d  = (D == 2 && n <= 64 && n % 2 == 0 && !g->cfg.has(O_2D_SIMPLE));
	// see LevelHost::fuse2_ok: a global fact only. (Refined levels qualify: patches that copy through and
	// coarse/fine faces -- whose ghost slots carry the interpolated value -- are handled by both kernels.)
	L->fuse2_ok = (D == 3 && li + 1 < (int) H.levels.size() && lv.P_global >= 256); // (TE_NO_FUSE2 is looked at where the path is chosen)
	L->ncf    = (int) cfs.size();
	int rc;
	{
		std::vector<int32_t> ord, bnd;
		for (int p = 0; p < P; p++) {
			bool b = false;
			for (int s = 0; s < NS; s++) b |= (fk[p * NS + s] == FACE_GHOST);
			(b ? bnd : ord).push_back(p);
		}
		L->n_int = (int) ord.size();
		L->n_bnd = (int) bnd.size();
		ord.insert(ord.end(), bnd.begin(), bnd.end());
		if ((rc = L->order.upload(ord))) return rc;
	}
	if (D == 3 && ((rc = L->xfbuf[0].alloc((size_t) std::max(P, 1) * 2 * L->nf)) || (rc = L->xfbuf[1].alloc((size_t) std::max(P, 1) * 2 * L->nf))
	               || (rc = L->f6buf.alloc((size_t) std::max(P, 1) * 6 * L->nf))))
		return rc;
	if (D == 3 && li > 0 && L->fuse2_ok && P > 0) { // a level that can read its right-hand side with FCORR
		if ((rc = L->fcorr.alloc((size_t) P * 4 * L->nf))) return rc;
		HIPCHK(hipMemset(L->fcorr.p, 0, sizeof(double) * L->fcorr.n));
	}
	if (D == 3 && L->fuse2_ok && P > 0 && (rc = L->rs6.alloc((size_t) P * 6 * L->nf / 4))) return rc;
	{
		std::vector<double>  gs((size_t) P * 3, 0.0), gh((size_t) P * 3, 1.0);
		std::vector<int32_t> ids(P);
		for (int p = 0; p < P; p++) {
			const int gp = lv.l2g[p];
			ids[p]       = lv.g_id[gp];
			for (int a = 0; a < D; a++) {
				gs[(size_t) p * 3 + a] = lv.g_starts[(size_t) gp * D + a];
				gh[(size_t) p * 3 + a] = lv.g_lengths[(size_t) gp * D + a] / n;
			}
		}
		if ((rc = L->geom_starts.upload(gs)) || (rc = L->geom_h.upload(gh)) || (rc = L->node_ids.upload(ids))) return rc;
	}
	if ((rc = L->cellvol.upload(cellvol))) return rc;
	{
		std::vector<int32_t> fkp(fk);
		for (auto &k : fkp)
			if (k >= FACE_LOCAL) k = FACE_DIRICHLET;
		if ((rc = L->face_kind_patch.upload(fkp))) return rc;
	}
	if ((rc = L->face_kind.upload(fk)) || (rc = L->face_src.upload(fs)) || (rc = L->face_kadj.upload(kadj))
	    || (rc = L->rh2.upload(rh2)) || (rc = L->cf_desc.upload(cfd)) || (rc = L->cf_slots.upload(cfs))
	    || (rc = L->ghost.alloc((size_t) std::max(nslots, 1) * L->nf)))
		return rc;

	// patch-solve plans (FftwPatchSolver.h:93-172: transform kinds per axis, eigenvalues)
	{
		const int            np = (int) keys.size();
		std::vector<double>  mats((size_t) np * 2 * D * n * n), lam((size_t) np * D * n);
		std::vector<int32_t> zm(np, 0);
		for (int k = 0; k < np; k++) {
			const int key = keys[k];
			zm[k]         = (key == (1 << NS) - 1);
			for (int a = 0; a < D; a++) {
				bool lo = (key >> (2 * a)) & 1, hi = (key >> (2 * a + 1)) & 1;
				int  tf, ti;
				if (lo && hi) {
					tf = 0;
					ti = 1;
				} else if (lo) {
					tf = ti = 2;
				} else if (hi) {
					tf = ti = 5;
				} else {
					tf = 3;
					ti = 4;
				}
				transformMatrix(tf, n, &mats[((size_t) k * 2 * D + a) * n * n]);
				transformMatrix(ti, n, &mats[((size_t) k * 2 * D + D + a) * n * n]);
				for (int i = 0; i < n; i++) {
					double s;
					if (lo && hi)
						s = sin(i * M_PI / (2 * n));
					else if (lo || hi)
						s = sin((i + 0.5) * M_PI / (2 * n));
					else
						s = sin((i + 1) * M_PI / (2 * n));
					lam[((size_t) k * D + a) * n + i] = 4 * s * s;
				}
			}
		}
		if (D == 3 && n == 32) { // k_ps_sym's tables: [plan][transform 6][parity 2][k-step 4][lane 64]
			std::vector<double> fs((size_t) np * PSS_FRAG, 0.0);
			bool                pure = true;
			for (int k = 0; k < np; k++)
				for (int a = 0; a < 3; a++) {
					const bool lo = (keys[k] >> (2 * a)) & 1, hi = (keys[k] >> (2 * a + 1)) & 1;
					if (lo != hi) {
						pure = false;
						continue;
					}
					const double *F = &mats[((size_t) k * 6 + a) * n * n], *G = &mats[((size_t) k * 6 + 3 + a) * n * n];
					for (int p = 0; p < 2; p++)
						for (int q = 0; q < 4; q++)
							for (int ln = 0; ln < 64; ln++) {
								const int j = ln & 15, g = ln >> 4;
								// forward: y as B operand and z as A operand take k = n = 4q + g, x as A operand k = g + 4q
								// (y comes first in the kernel: slot 0 = y, 1 = x, 2 = z)
								// inverse: x as B operand (k = m = 4q + g), y and z as A operands with k = m = g + 4q
								const int nf = (a == 0) ? g + 4 * q : 4 * q + g, mi = (a == 0) ? 4 * q + g : g + 4 * q;
								const int sf = (a == 0) ? 1 : (a == 1 ? 0 : 2);
								fs[(size_t) k * PSS_FRAG + ((sf * 2 + p) * 4 + q) * 64 + ln]      = F[(2 * j + p) * n + nf];
								fs[(size_t) k * PSS_FRAG + (((3 + a) * 2 + p) * 4 + q) * 64 + ln] = G[j * n + 2 * mi + p];
							}
				}
			L->sym_ok = pure;
			if ((rc = L->matsym.upload(fs))) return rc;
			if (!pure) { // per-patch choice between k_ps_sym and k_ps_fused
				std::vector<int32_t> lst, mixed;
				for (int p = 0; p < P; p++) {
					bool ok = true;
					for (int a = 0; a < 3; a++) ok &= (((keys[plan[p]] >> (2 * a)) & 1) == ((keys[plan[p]] >> (2 * a + 1)) & 1));
					(ok ? lst : mixed).push_back(p);
				}
				L->n_pure = (int) lst.size();
				lst.insert(lst.end(), mixed.begin(), mixed.end());
				if ((rc = L->ps_list.upload(lst))) return rc;
			}
		}
		if (D == 2 && n <= 64) {
			std::vector<double> mt(mats.size());
			for (size_t m = 0; m < mats.size() / ((size_t) n * n); m++)
				for (int i = 0; i < n; i++)
					for (int j = 0; j < n; j++) mt[m * n * n + (size_t) j * n + i] = mats[m * n * n + (size_t) i * n + j];
			if ((rc = L->matsT.upload(mt))) return rc;
		}
		if ((rc = L->corr.alloc((size_t) std::max(P, 1) * NS * L->nf))) return rc;
		if ((rc = L->plan.upload(plan)) || (rc = L->mats.upload(mats)) || (rc = L->lam.upload(lam))
		    || (rc = L->zero_mode.upload(zm)))
			return rc;
	}

	// transfers to level li+1. A child (or a copy-through patch) whose parent lives on another rank
	// ships its restricted block there; the parent's rank ships octant blocks back for prolongation.
	// Canonical block order on both ends: (peer, parent patch (global), orthant).
	if (li + 1 < (int) H.levels.size()) {
		const Level         &cv = H.levels[li + 1];
		std::vector<int32_t> parent(P), orth(P), child((size_t) cv.P * NCH, -1), copy(cv.P, 0);
		struct Blk {
			int     peer, gpar, o, patch;
			int64_t size;
			bool    operator<(const Blk &b) const { return std::tie(peer, gpar, o) < std::tie(b.peer, b.gpar, b.o); }
		};
		std::vector<Blk> up, down;
		const bool       repl = cv.replicated && !lv.replicated;
		for (int p = 0; p < P; p++) {
			const int gp = lv.l2g[p], gpar = lv.g_parent[gp];
			orth[p]      = lv.g_orth_on_parent[gp];
			if (cv.g_rank[gpar] == me) {
				const int pc = cv.g_local[gpar];
				parent[p]    = pc;
				if (orth[p] < 0) {
					copy[pc]                  = 1;
					child[(size_t) pc * NCH] = p;
				} else {
					child[(size_t) pc * NCH + orth[p]] = p;
				}
			} else {
				up.push_back({cv.g_rank[gpar], gpar, orth[p] < 0 ? 0 : orth[p], p,
				              (int64_t) (orth[p] < 0 ? L->nc : L->nc / NCH)});
			}
		}
		for (int gf = 0; gf < lv.P_global; gf++) {
			const int gpar = lv.g_parent[gf];
			if (cv.g_rank[gpar] != me || lv.g_rank[gf] == me) continue;
			const int o = lv.g_orth_on_parent[gf];
			down.push_back({lv.g_rank[gf], gpar, o < 0 ? 0 : o, cv.g_local[gpar], (int64_t) (o < 0 ? L->nc : L->nc / NCH)});
			if (o < 0) copy[cv.g_local[gpar]] = 1;
		}
		std::sort(up.begin(), up.end());
		std::sort(down.begin(), down.end());
		std::vector<int32_t>                 upd, downd;
		std::vector<int64_t>                 upo, downo;
		std::vector<std::pair<int, int64_t>> ups, downs;
		int64_t                              pos = 0;
		for (size_t i = 0; i < up.size(); i++) {
			upd.push_back(up[i].patch);
			upd.push_back(orth[up[i].patch]);
			upo.push_back(pos);
			ups.emplace_back(up[i].peer, up[i].size);
			parent[up[i].patch] = -((int) i + 2); // prolong reads block i of upbuf
			pos += up[i].size;
		}
		std::vector<int32_t> bcd;
		if (repl) { // (up is empty: every parent is local) one block per local patch, in the order the receivers expect: (parent, orthant)
			std::vector<Blk> bc;
			for (int p = 0; p < P; p++)
				bc.push_back({0, lv.g_parent[lv.l2g[p]], orth[p] < 0 ? 0 : orth[p], p, (int64_t) (orth[p] < 0 ? L->nc : L->nc / NCH)});
			std::sort(bc.begin(), bc.end());
			for (size_t i = 0; i < bc.size(); i++) {
				upd.push_back(bc[i].patch); // (fine patch, orthant): k_restrict_pack restricts it into its block
				upd.push_back(orth[bc[i].patch]);
				bcd.push_back(parent[bc[i].patch]); // (coarse patch, orthant or -1): k_prolong_pack copies the finished octant out
				bcd.push_back(orth[bc[i].patch] < 0 ? -1 : bc[i].o);
				upo.push_back(pos);
				pos += bc[i].size;
			}
		}
		const int64_t up_total = pos;
		pos                    = 0;
		for (size_t i = 0; i < down.size(); i++) {
			const int pc = down[i].patch;
			const bool cp = down[i].size == (int64_t) L->nc;
			downd.push_back(pc);
			downd.push_back(cp ? -1 : down[i].o);
			downo.push_back(pos);
			downs.emplace_back(down[i].peer, down[i].size);
			child[(size_t) pc * NCH + (cp ? 0 : down[i].o)] = -((int) i + 2); // restrict reads block i of downbuf
			pos += down[i].size;
		}
		const int64_t down_total = pos;
		for (int pc = 0; pc < cv.P; pc++) {
			if (copy[pc]) continue;
			for (int o = 0; o < NCH; o++)
				if (child[(size_t) pc * NCH + o] == -1)
					return te::fail(TE_EINVAL, "te_gmg_create: coarse patch with a missing child");
		}
		L->Pc      = cv.P;
		// (repl: the blocks in `down` are received for the restriction only; every parent is local)
		const bool parents_local = up.empty() && (down.empty() || repl);
		L->prolong_fusable = (D == 3 && L->ncf == 0 && parents_local
		                      && std::all_of(orth.begin(), orth.end(), [](int32_t o) { return o >= 0; }));
		L->has_copy           = std::any_of(orth.begin(), orth.end(), [](int32_t o) { return o < 0; });
		L->prolong_fusable_cf = (D == 3 && parents_local && !g->cfg.has(O_NO_CFP));
		if (D == 2 && L->lds2d && up.empty() && down.empty()) {
			L->fuse2d          = true;
			// (faces on other ranks are fine: their values of u + P e arrive in ghost slots, packProlongFaces2d)
			L->prolong_fusable = ((g->cfg.has(O_2D_NO_MR_FUSE) ? L->nslots == 0 : L->ncf == 0)
			                      && std::all_of(orth.begin(), orth.end(), [](int32_t o) { return o >= 0; }));
		}
		if (D == 2 && L->lds2d) { // the 3D fusions in 2D (kernels2d.hpp)
			// a global fact, as in 3D (all ranks and every partition take the same arithmetic path): the level is uniformly
			// refined everywhere -- no coarse/fine face, every patch a quadrant child
			bool uniform = true;
			for (int gp = 0; gp < lv.P_global && uniform; gp++) {
				uniform = lv.g_orth_on_parent[gp] >= 0;
				for (int s2 = 0; s2 < NS && uniform; s2++) uniform = lv.g_nbr_kind[(size_t) gp * NS + s2] <= NBR_NORMAL;
			}
			L->fuse2_ok = uniform;
			if (uniform && (rc = L->e4buf.alloc((size_t) std::max(P, 1) * 4 * n))) return rc;
		}
		L->n_up    = (int) (upd.size() / 2);
		L->n_down  = (int) down.size();
		L->repl_up = repl;
		if (D == 3 && repl) {
			bool uniform = true;
			for (int gp = 0; gp < lv.P_global && uniform; gp++) {
				uniform = lv.g_orth_on_parent[gp] >= 0;
				for (int s2 = 0; s2 < NS && uniform; s2++) uniform = lv.g_nbr_kind[(size_t) gp * NS + s2] <= NBR_NORMAL;
			}
			L->post_exchange_free = uniform;
			if (uniform && nremote > 0) {
				std::vector<int32_t> sp(nremote), so(nremote);
				for (int i = 0; i < nremote; i++) {
					sp[i] = cv.g_local[lv.g_parent[recvs[i].nb]];
					so[i] = lv.g_orth_on_parent[recvs[i].nb];
				}
				if ((rc = L->slot_parent.upload(sp)) || (rc = L->slot_orth.upload(so))) return rc;
			}
			// in-place exchange of the restricted blocks: who fills which coarse patches
			std::vector<int> owner(cv.P_global, -1), lo(H.nranks, cv.P_global), hi(H.nranks, -1), cnt(H.nranks, 0);
			bool             direct = true;
			for (int gf = 0; gf < lv.P_global && direct; gf++) {
				int &o = owner[lv.g_parent[gf]];
				if (o >= 0 && o != lv.g_rank[gf]) direct = false;
				o = lv.g_rank[gf];
			}
			for (int pc = 0; pc < cv.P_global && direct; pc++) {
				const int r = owner[pc], lc = cv.g_local[pc];
				if (r < 0) {
					direct = false;
					break;
				}
				lo[r] = std::min(lo[r], lc), hi[r] = std::max(hi[r], lc), cnt[r]++;
			}
			for (int r = 0; r < H.nranks && direct; r++) direct = (cnt[r] == 0 || cnt[r] == hi[r] - lo[r] + 1);
			if (direct) {
				for (int r = 0; r < H.nranks; r++) {
					if (r == me || (cnt[r] == 0 && cnt[me] == 0)) continue;
					L->tx_direct.peers.push_back(r);
					L->tx_direct.send_off.push_back(cnt[me] ? (int64_t) lo[me] * (int64_t) L->nc : 0);
					L->tx_direct.send_cnt.push_back((int64_t) cnt[me] * (int64_t) L->nc);
					L->tx_direct.recv_off.push_back(cnt[r] ? (int64_t) lo[r] * (int64_t) L->nc : 0);
					L->tx_direct.recv_cnt.push_back((int64_t) cnt[r] * (int64_t) L->nc);
				}
				L->repl_direct = true;
			}
		}
		if (repl) {
			// restrict: the same range of upbuf to every other rank (if this rank has patches here at all), and from every rank
			// that has patches here its blocks; prolong: nothing
			L->tx_up = mergePlan({}, downs);
			ExPlan &pl = L->tx_up;
			if (up_total > 0) {
				ExPlan full;
				size_t k = 0;
				for (int r = 0; r < H.nranks; r++) {
					if (r == me) continue;
					while (k < pl.peers.size() && pl.peers[k] < r) k++;
					const bool have = k < pl.peers.size() && pl.peers[k] == r;
					full.peers.push_back(r);
					full.send_off.push_back(0);
					full.send_cnt.push_back(up_total);
					full.recv_off.push_back(have ? pl.recv_off[k] : 0);
					full.recv_cnt.push_back(have ? pl.recv_cnt[k] : 0);
				}
				pl = full;
			}
			L->tx_down = ExPlan();
			if ((rc = L->bc_desc.upload(bcd))) return rc;
		} else {
			L->tx_up   = mergePlan(ups, downs);   // restrict: send child blocks, receive into downbuf
			L->tx_down = mergePlan(downs, ups);   // prolong: send octants, receive into upbuf
		}
		if ((rc = L->parent.upload(parent)) || (rc = L->orth.upload(orth)) || (rc = L->child.upload(child))
		    || (rc = L->copy.upload(copy)) || (rc = L->up_desc.upload(upd)) || (rc = L->down_desc.upload(downd))
		    || (rc = L->up_off.upload(upo)) || (rc = L->down_off.upload(downo))
		    || (rc = L->upbuf.alloc((size_t) std::max<int64_t>(up_total, 1)))
		    || (rc = L->downbuf.alloc((size_t) std::max<int64_t>(down_total, 1))))
			return rc;
	}
	g->levels.push_back(std::move(L));
	return TE_OK;
}

int newVec(te_gmg *g, int level, te_vec **out)
{
	LevelHost &L = *g->levels[level];
	auto       v = new te_vec;
	v->g         = g;
	v->level     = level;
	v->n         = (size_t) L.P * L.nc;
	hipError_t e = hipMalloc(&v->d, sizeof(double) * std::max<size_t>(v->n, 2));
	if (e != hipSuccess) {
		delete v;
		return te::fail(TE_EHIP, std::string("hipMalloc(vector): ") + hipGetErrorString(e));
	}
	e = hipMemsetAsync(v->d, 0, sizeof(double) * v->n, g->stream);
	if (e != hipSuccess) {
		(void) hipFree(v->d);
		delete v;
		return te::fail(TE_EHIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
	}
	*out = v;
	return TE_OK;
}

inline bool sameShape(const te_vec *a, const te_vec *b) { return a && b && a->g == b->g && a->level == b->level; }

// ------------------------------------------------------------------------------ launches
// Watchdog (multi-rank only): a peer that never posts its half of an exchange leaves RCCL (or the host callback)
// waiting for ever, with no error. Every exchange arms a deadline and records an event behind itself on its
// stream; a thread polls the event and ends the PROCESS (exit status 86, message on stderr) when the deadline
// passes first -- the launcher (torchrun, mpirun) then takes the job down instead of hanging the node.
void watchdogRetire(te_gmg::Watchdog &w) // (mutex held) drop completed exchanges from the front
{
	while (w.head < w.tail) {
		auto &sl = w.slot[w.head % te_gmg::Watchdog::RING];
		if (!sl.recorded || hipEventQuery(sl.ev) != hipSuccess) break;
		w.head++;
	}
}
// (mutex held through `lk`) The ring is full -- the host is RING exchanges ahead of the GPU: wait for the oldest outstanding one
// (without the mutex: the polling thread needs it) until a slot is free. Nothing that is being watched is ever given up.
void watchdogMakeRoom(te_gmg::Watchdog &w, std::unique_lock<std::mutex> &lk)
{
	watchdogRetire(w);
	while (w.tail - w.head == te_gmg::Watchdog::RING) {
		auto      &sl = w.slot[w.head % te_gmg::Watchdog::RING];
		hipEvent_t ev = sl.ev;
		const bool rec = sl.recorded;
		lk.unlock();
		if (rec)
			(void) hipEventSynchronize(ev);
		else
			std::this_thread::sleep_for(std::chrono::milliseconds(1)); // (still inside its host call on another thread)
		lk.lock();
		watchdogRetire(w);
	}
}
void watchdogLoop(te_gmg *g)
{
	auto &w = g->wd;
	while (!w.stop.load()) {
		std::this_thread::sleep_for(std::chrono::milliseconds(50));
		if (g->push.err_host && *g->push.err_host && g->push.fatal.load()) {
			fprintf(stderr,
			        "te_hip watchdog: rank %d: a direct-store exchange gave up waiting for a peer's data -- a peer is missing or issued a "
			        "different exchange sequence; ending the process\n",
			        g->rank);
			fflush(stderr);
			_exit(86);
		}
		std::lock_guard<std::mutex> lk(w.mu);
		watchdogRetire(w);
		if (w.head == w.tail) continue;
		const auto  &sl     = w.slot[w.head % te_gmg::Watchdog::RING];
		const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - sl.since).count();
		if (waited > w.timeout_s) {
			fprintf(stderr,
			        "te_hip watchdog: rank %d: exchange (tag %d, level %d) has not completed after %.0f s -- a peer is "
			        "missing or issued a different exchange sequence; ending the process\n",
			        g->rank, sl.tag, sl.level, waited);
			fflush(stderr);
			_exit(86);
		}
	}
}
void watchdogStart(te_gmg *g)
{
	auto &w = g->wd;
	if (w.th.joinable() || (g->nranks < 2 && !g->cfg.has(O_EXCHANGE_TIMEOUT))) return; // (one rank: only when asked for, te_gmg_watchdog_selftest)
	w.timeout_s = g->cfg.real(O_EXCHANGE_TIMEOUT, w.timeout_s);
	if (w.timeout_s <= 0) return; // TE_EXCHANGE_TIMEOUT=0 disables it
	for (auto &sl : w.slot)
		if (hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) return;
	w.th = std::thread(watchdogLoop, g);
}
void watchdogStop(te_gmg *g)
{
	auto &w = g->wd;
	if (w.th.joinable()) {
		w.stop.store(true);
		w.th.join();
	}
	for (auto &sl : w.slot) {
		if (sl.ev) (void) hipEventDestroy(sl.ev);
		sl.ev = nullptr;
	}
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

int doExchange(te_gmg *g, int tag, const ExPlan &pl, const double *send, double *recv, hipStream_t stream = nullptr)
{
	if (!stream) stream = g->stream;
	const bool timed = (stream == g->stream);
	if (pl.empty()) return TE_OK;
	if (g->recording) { // te_gmg_verify_schedule: who would talk to whom, in which order; nothing moves
		for (size_t i = 0; i < pl.peers.size(); i++)
			g->record.push_back({tag, g->cur_level, pl.peers[i], pl.send_cnt[i], pl.recv_cnt[i]});
		return TE_OK;
	}
	WatchdogArm arm(g, stream, tag);
	if (g->rccl.comm) {
		// one RCCL group per exchange, enqueued on the solver stream behind the pack kernel: every
		// send/recv of the exchange progresses together over the direct xGMI links, no host round trip
		int64_t total = 0; // (loop-back only)
		if (g->cfg.has(O_RCCL_LOOPBACK)) {
			for (size_t i = 0; i < pl.peers.size(); i++) total += std::max(pl.send_cnt[i], pl.recv_cnt[i]);
			if ((size_t) (2 * total) > g->loopbuf.n) { // (grows to the largest plan of the solver within the first cycle; outside the timed scope)
				HIPCHK(hipStreamSynchronize(g->stream));
				HIPCHK(hipStreamSynchronize(g->comm_stream)); // (overlapped exchanges use it too)
				if (g->loopbuf.p) HIPCHK(hipFree(g->loopbuf.p));
				g->loopbuf.p = nullptr;
				g->loopbuf.n = 0;
				int rc0      = g->loopbuf.alloc((size_t) (2 * total));
				if (rc0) {
					g->loopbuf.p = nullptr;
					g->loopbuf.n = 0;
					return rc0;
				}
			}
		}
		std::unique_ptr<Timed> t(timed ? new Timed(g, KC_EXCHANGE, 0) : nullptr);
		if (g->cfg.has(O_RCCL_LOOPBACK)) {
			// DIAGNOSTIC (tools/mr8_budget.py): one rank of an N-rank hierarchy alone on a GPU, every peer replaced by the rank
			// itself -- the same group of ncclRecv/ncclSend calls with the same message sizes, between scratch buffers. What
			// is measured is real (host enqueue cost, RCCL's launch, this rank's kernels with the GPU to themselves); the
			// exchanged DATA are not: results are meaningless in this mode.
			int     rc  = g->rccl.GroupStart();
			int64_t off = 0;
			for (size_t i = 0; i < pl.peers.size() && rc == 0; i++) {
				const size_t c = (size_t) std::max(pl.send_cnt[i], pl.recv_cnt[i]);
				if (c == 0) continue;
				rc = g->rccl.Recv(g->loopbuf.p + total + off, c, ncclFloat64, 0, g->rccl.comm, stream);
				if (rc == 0) rc = g->rccl.Send(g->loopbuf.p + off, c, ncclFloat64, 0, g->rccl.comm, stream);
				off += (int64_t) c;
			}
			int rc2 = g->rccl.GroupEnd();
			if (rc || rc2) return te::fail(TE_ESTATE, std::string("RCCL loopback exchange failed: ") + g->rccl.GetErrorString(rc ? rc : rc2));
			return TE_OK;
		}
		int rc = g->rccl.GroupStart();
		for (size_t i = 0; i < pl.peers.size() && rc == 0; i++) {
			if (pl.recv_cnt[i] > 0)
				rc = g->rccl.Recv(recv + pl.recv_off[i], (size_t) pl.recv_cnt[i], ncclFloat64, pl.peers[i], g->rccl.comm, stream);
			if (rc == 0 && pl.send_cnt[i] > 0)
				rc = g->rccl.Send(send + pl.send_off[i], (size_t) pl.send_cnt[i], ncclFloat64, pl.peers[i], g->rccl.comm, stream);
		}
		int rc2 = g->rccl.GroupEnd();
		if (rc || rc2) return te::fail(TE_ESTATE, std::string("RCCL exchange failed: ") + g->rccl.GetErrorString(rc ? rc : rc2));
		return TE_OK;
	}
	if (!g->exchange)
		return te::fail(TE_ESTATE, "this level has off-rank neighbours: call te_gmg_set_exchange or te_gmg_use_rccl first");
	std::unique_ptr<Timed> t(timed ? new Timed(g, KC_EXCHANGE, 0) : nullptr);
	int   rc = g->exchange(g->exchange_user, tag, send, recv, (int) pl.peers.size(), pl.peers.data(), pl.send_off.data(),
	                       pl.send_cnt.data(), pl.recv_off.data(), pl.recv_cnt.data(), (void *) stream);
	if (rc) return te::fail(TE_ESTATE, "exchange callback failed with status " + std::to_string(rc));
	return TE_OK;
}
// Sum (op 0) or maximum (op 1) over the ranks of n (<= 4) doubles that the solver stream has left in g->result;
// returns them in g->result_host. One rank: a copy. Replaces the MPI_Allreduce of Vector.h:294,306,319.
int finishReduce(te_gmg *g, int n, int op, bool global)
{
	if (global && g->nranks > 1 && g->rccl.comm) {
		WatchdogArm arm(g, g->stream, 100 + op);
		int rc = g->rccl.AllReduce(g->result.p, g->result.p, (size_t) n, ncclFloat64, op ? ncclMax : ncclSum, g->rccl.comm, g->stream);
		if (rc) return te::fail(TE_ESTATE, std::string("ncclAllReduce failed: ") + g->rccl.GetErrorString(rc));
	}
	HIPCHK(hipMemcpyAsync(g->result_host, g->result.p, n * sizeof(double), hipMemcpyDeviceToHost, g->stream));
	HIPCHK(hipStreamSynchronize(g->stream));
	if (global && g->nranks > 1 && !g->rccl.comm) {
		if (!g->allreduce)
			return te::fail(TE_ESTATE, "a reduction over ranks needs te_gmg_use_rccl or te_gmg_set_allreduce");
		WatchdogArm arm(g, g->stream, 100 + op);
		int rc = g->allreduce(g->allreduce_user, g->result_host, n, op);
		if (rc) return te::fail(TE_ESTATE, "allreduce callback failed with status " + std::to_string(rc));
	}
	return TE_OK;
}
// One exchange through the direct-store transport (pushkernels.hpp): kind 1 = the level's face exchange (send: the layers in
// send order; lands in the peers' ghost slots of this exchange's parity), kind 2 = the in-place exchange of restricted blocks
// (send = the coarse level's right-hand side of this gather's parity; every rank's run lands at the same offsets there).
// Two launches on `stream`: push, wait. The epochs only ever grow, so a flag that is already ahead (a fast peer) passes.
// pushBegin: the exchange's parity and epoch (L.push_par / push_ep) and what it will wait for (L.push_wait); pushFinish: the wait
// kernel, and the switch of the ghost buffer the kernels read. Between the two: k_push_ranges, or a pack kernel that stores
// into the peers' buffers itself (PackPush).
void pushBegin(te_gmg *g, LevelHost &L, int kind)
{
	const ExPlan &pl = kind == 1 ? L.fx : L.tx_direct;
	uint64_t     &ep = kind == 1 ? L.face_epoch : L.blk_epoch;
	L.push_par       = (int) (ep & 1); // this exchange's buffer
	L.push_ep        = ++ep;
	const int slot   = 2 * L.index + (kind - 1);
	L.push_wait.n    = 0;
	for (size_t i = 0; i < pl.peers.size(); i++)
		if (pl.recv_cnt[i] > 0) L.push_wait.flag[L.push_wait.n++] = g->push.flags + (size_t) pl.peers[i] * g->push.nslot + slot;
}
int pushFinish(te_gmg *g, LevelHost &L, int kind, hipStream_t stream)
{
	const long long budget = (long long) (g->push.timeout_s * 1e8); // wall_clock64: 100 MHz
	if (L.push_wait.n > 0)
		hipLaunchKernelGGL(k_push_wait, dim3(1), dim3(64), 0, stream, L.push_wait, L.push_ep, budget, g->push.err, g->push.err_host);
	if (kind == 1) L.ghost_par = L.push_par; // what the kernels behind this exchange read
	HIPCHK(hipGetLastError());
	return TE_OK;
}
int pushExchange(te_gmg *g, LevelHost &L, int kind, const double *send, hipStream_t stream = nullptr)
{
	if (!stream) stream = g->stream;
	const ExPlan &pl = kind == 1 ? L.fx : L.tx_direct;
	if (pl.empty()) return TE_OK;
	if ((int) pl.peers.size() > PUSH_MAX_PEERS) return te::fail(TE_EUNSUPPORTED, "direct-store exchange: too many peers");
	std::unique_ptr<Timed> t(stream == g->stream ? new Timed(g, KC_EXCHANGE, 0) : nullptr);
	WatchdogArm            arm(g, stream, kind);
	pushBegin(g, L, kind);
	const int par  = L.push_par;
	const int slot = 2 * L.index + (kind - 1);
	PushPlan  pp;
	int64_t   most = 0;
	pp.n           = 0;
	for (size_t i = 0; i < pl.peers.size(); i++) {
		const int r = pl.peers[i];
		if (pl.send_cnt[i] > 0) {
			PushPeer &q = pp.p[pp.n++];
			q.dst       = (kind == 1 ? L.push_peer_ghost[par][i] : L.push_peer_cf[par][i] + pl.send_off[i]);
			q.src_off   = pl.send_off[i];
			q.cnt       = pl.send_cnt[i];
			q.flag      = g->push.peer_flags[r] + (size_t) g->rank * g->push.nslot + slot;
			most        = std::max(most, q.cnt);
		}
	}
	if (pp.n > 0) {
		const int bx = (int) std::min<int64_t>(64, std::max<int64_t>(1, most / 2 / 256 / 4));
		hipLaunchKernelGGL(k_push_ranges, dim3(bx, pp.n), dim3(256), 0, stream, send, pp, L.push_ep, L.push_done.p + (kind - 1), (const int *) g->push.err);
	}
	return pushFinish(g, L, kind, stream);
}
// the level's face exchange: remote slots of the current ghost buffer <- the peers' layers (`send` in send order)
int faceExchange(te_gmg *g, LevelHost &L, const double *send, hipStream_t stream = nullptr)
{
	if (g->push.on && L.push_faces && !g->recording && L.dim == 3) return pushExchange(g, L, 1, send, stream);
	return doExchange(g, 1, L.fx, send, L.ghostCur(), stream);
}
// make every ghost plane of `u` current: remote same-level faces (pack -> exchange -> ghost slots
// [0, nremote)), then the coarse/fine planes. Replaces SchurHelper.h:145-150 updateInterfaceDist.
// `ps`: the iterate is u + P(ps->coarse) (never stored): the faces are packed with the correction added.
// may_push: the exchange that follows is on the solver stream -- with the direct-store transport the pack kernel then stores the
// layers into the receivers' ghost slots itself and raises their flags (PackPush); returns true when it did (the caller finishes
// with pushFinish instead of an exchange of the send buffer)
template <int N> bool packFaces(te_gmg *g, LevelHost &L, const double *u, const ProlongSrc *ps, bool may_push = false)
{
	const bool push = may_push && g->push.on && L.push_faces && !g->recording && L.push_face_dst[0].p;
	Timed      t(g, push ? KC_EXCHANGE : KC_PACK, (size_t) L.nremote * L.nf);
	const dim3 grid(L.nremote), blk(N * N < 256 ? N * N : 256);
	PackPush   pp;
	if (push) {
		pushBegin(g, L, 1);
		pp.dst    = L.push_face_dst[L.push_par].p;
		pp.flags  = L.push_face_flags.p;
		pp.nflags = (int) L.push_face_flags.n;
		pp.epoch  = L.push_ep;
		pp.done   = L.push_done.p;
		pp.err    = g->push.err;
	}
	if (L.pack_f6) { // the iterate exists only as its face layers
		ProlongSrc none{nullptr, nullptr, nullptr};
		hipLaunchKernelGGL(k_pack_faces6_3d<N>, grid, blk, 0, g->stream, L.send_faces.p, L.pack_f6, ps ? *ps : none, L.sendbuf.p, L.f6Off(), pp);
	} else if (ps)
		hipLaunchKernelGGL(k_pack_faces_prolong3d<N>, grid, blk, 0, g->stream, L.send_faces.p, u, *ps, L.sendbuf.p, pp);
	else
		hipLaunchKernelGGL(k_pack_faces3d<N>, grid, blk, 0, g->stream, L.send_faces.p, u, L.sendbuf.p, pp);
	return push;
}
// ghost planes of the coarse/fine faces from the current iterate (u, or its face layers L.pack_f6 when it was never stored)
template <int N> void cfGhosts(te_gmg *g, LevelHost &L, const double *u, const ProlongSrc *ps)
{
	Timed      t(g, KC_CFGHOST, (size_t) L.ncf * L.nf);
	const dim3 grid(L.ncf), blk(N * N < 256 ? N * N : 256);
	if (L.pack_f6) {
		if (ps)
			hipLaunchKernelGGL((k_cf_ghost6_3d<N, true>), grid, blk, 0, g->stream, L.cf_desc.p, L.cf_slots.p, L.pack_f6, *ps, L.ghostCur(), L.f6Off());
		else
			hipLaunchKernelGGL((k_cf_ghost6_3d<N, false>), grid, blk, 0, g->stream, L.cf_desc.p, L.cf_slots.p, L.pack_f6, ProlongSrc(), L.ghostCur(), L.f6Off());
	} else if (ps)
		hipLaunchKernelGGL(k_cf_ghost_prolong3d<N>, grid, blk, 0, g->stream, L.cf_desc.p, L.cf_slots.p, u, *ps, L.ghostCur());
	else
		hipLaunchKernelGGL(k_cf_ghost3d<N>, grid, blk, 0, g->stream, L.cf_desc.p, L.cf_slots.p, u, L.ghostCur());
}
template <int N> int prepareGhosts(te_gmg *g, LevelHost &L, const double *u, const ProlongSrc *ps = nullptr)
{
	if (L.patch_local) return TE_OK; // the patch operator reads no neighbour
	L.ghost_has_v = false;
	if (L.nremote > 0) {
		// the face layers of an iterate that exists only as such already sit in send order (LevelHost::f6off): sent from there
		const bool direct = L.pack_f6 && !ps && L.f6Off();
		const bool pushed = !direct && packFaces<N>(g, L, u, ps, true);
		int        rc     = pushed ? pushFinish(g, L, 1, g->stream) : faceExchange(g, L, direct ? L.pack_f6 : L.sendbuf.p);
		if (rc) return rc;
	}
	if (L.ncf == 0) return TE_OK;
	cfGhosts<N>(g, L, u, ps);
	return TE_OK;
}
// Run `launch(subset)` over all patches of the level with current ghosts. With off-rank neighbours the
// exchange goes to the communication stream and the interior patches (no ghost-slot face) are computed
// underneath it; the boundary patches follow once the receive has landed. (north star: "ghost-cell
// exchange ... overlapped with interior smoothing")
template <int N, class F> int withGhosts(te_gmg *g, LevelHost &L, const double *u, F launch_, const double *xf_in = nullptr,
                                         double *xf_out = nullptr, const ProlongSrc *ps = nullptr)
{
	auto launch = [&](LevelDev D) {
		D.xf     = xf_in;
		D.xf_out = xf_out;
		launch_(D);
	};
	// (levels with few local patches: nothing worth hiding under the exchange, and the second stream and its two
	// events only add host calls and latency)
	// TE_OVERLAP_MIN (tests set 0 so that their small levels take the overlapped path). The split costs a second launch -- no
	// launch is shorter than one patch march, about 30 us -- and two cross-stream event hand-overs of about 20 us each, so it
	// pays only where the interior launch is much longer than that: more local patches than the chip holds workgroups at once
	// (3 x 256). Measured per rank with the exchanges in loop-back (tools/mr8_budget.py): at 512 local patches (512^3 on eight
	// ranks) the cycle is 525 us with the split and 475 us without it.
	L.ghost_has_v = false;
	const int mode = L.overlap_mode >= 0 ? L.overlap_mode : (L.P < g->cfg.num(O_OVERLAP_MIN, 768) ? 0 : (g->cfg.num(O_OVERLAP_MODE, 1) == 2 ? 2 : 1));
	if (L.patch_local || g->recording || L.nremote == 0 || !g->overlap || L.n_int == 0 || mode == 0) {
		int rc = prepareGhosts<N>(g, L, u, ps);
		if (rc) return rc;
		launch(L.dev());
		return TE_OK;
	}
	if (mode == 2) {
		// the interior patches go to the second stream (they wait for nothing but what the solver stream has done so far); pack,
		// exchange, coarse/fine ghosts and the boundary patches stay on the solver stream, which then waits for the interior
		HIPCHK(hipEventRecord(g->ev_pack, g->stream));
		HIPCHK(hipStreamWaitEvent(g->comm_stream, g->ev_pack, 0));
		std::swap(g->stream, g->comm_stream); // (every launch helper launches on g->stream)
		launch(L.devPart(false));
		hipError_t e = hipEventRecord(g->ev_recv, g->stream);
		std::swap(g->stream, g->comm_stream);
		HIPCHK(e);
		const bool pushed = packFaces<N>(g, L, u, ps, true);
		int        rc     = pushed ? pushFinish(g, L, 1, g->stream) : faceExchange(g, L, L.sendbuf.p);
		if (rc) return rc;
		if (L.ncf > 0) cfGhosts<N>(g, L, u, ps);
		launch(L.devPart(true));
		HIPCHK(hipStreamWaitEvent(g->stream, g->ev_recv, 0));
		return TE_OK;
	}
	packFaces<N>(g, L, u, ps);
	HIPCHK(hipEventRecord(g->ev_pack, g->stream));
	HIPCHK(hipStreamWaitEvent(g->comm_stream, g->ev_pack, 0));
	int rc = faceExchange(g, L, L.sendbuf.p, g->comm_stream);
	if (rc) return rc;
	HIPCHK(hipEventRecord(g->ev_recv, g->comm_stream));
	launch(L.devPart(false)); // interior, concurrent with the exchange
	HIPCHK(hipStreamWaitEvent(g->stream, g->ev_recv, 0));
	if (L.ncf > 0) cfGhosts<N>(g, L, u, ps);
	launch(L.devPart(true)); // boundary
	return TE_OK;
}
// one launch of k_stencil3d<N, MODE, ZS> with or without fused sums (RED: march3d.hpp StencilRed)
template <int N, int MODE, int ZS>
void launchStencilZS(te_gmg *g, dim3 grid, const LevelDev &D, const double *u, const double *f, double *out, double omega,
                     const RestrictDst &rd, int redmode, const RedSrc &rs)
{
	const dim3 blk(Tile3<N>::TPB);
	if constexpr (MODE == MODE_APPLY) {
		if (redmode == RED_OUT_A) {
			hipLaunchKernelGGL((k_stencil3d<N, MODE, ZS, RED_OUT_A>), grid, blk, 0, g->stream, D, u, f, out, omega, rd, rs);
			return;
		}
		if (redmode == RED_OUT_A_OUT) {
			hipLaunchKernelGGL((k_stencil3d<N, MODE, ZS, RED_OUT_A_OUT>), grid, blk, 0, g->stream, D, u, f, out, omega, rd, rs);
			return;
		}
	}
	if constexpr (MODE == MODE_RESID) {
		if (redmode == RED_OUT_OUT) {
			hipLaunchKernelGGL((k_stencil3d<N, MODE, ZS, RED_OUT_OUT>), grid, blk, 0, g->stream, D, u, f, out, omega, rd, rs);
			return;
		}
	}
	hipLaunchKernelGGL((k_stencil3d<N, MODE, ZS, RED_NONE>), grid, blk, 0, g->stream, D, u, f, out, omega, rd, rs);
}
// redmode != RED_NONE: the kernel leaves one pair of partial sums per work item in g->partial (red_a: the second operand
// of the dot product); *red_items = their number (the caller runs k_reduce_final2 over them)
template <int N, int MODE> int launchStencilN(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out,
                                              double omega, RestrictDst rd = RestrictDst(), const double *xf_in = nullptr,
                                              int redmode = RED_NONE, const double *red_a = nullptr, int *red_items = nullptr)
{
	// enough workgroups to fill 256 CUs a few times over: split patches into z-slabs when few
	int zs = 1;
	if (N >= 8) {
		while (zs < 4 && (g->cfg.has(O_ZS_FORCE) || (size_t) L.P * zs < 2048) && N / (zs * 2) >= 4) zs *= 2;
		if (zs == 4 && N == 32 && L.P <= 64 && !g->cfg.has(O_NO_ZS8)) zs = 8; // (see rbgsSlabs)
	}
	if (redmode != RED_NONE) {
		if ((size_t) 2 * L.P * zs > g->partial.n) return te::fail(TE_ESTATE, "launchStencil: partial-sum buffer too small");
		if (red_items) *red_items = L.P * zs;
	}
	auto launch = [&](LevelDev D) {
		if (D.count == 0) return;
		Timed t(g, zs > 1 ? KC_STENCIL_SLABS
		                  : (MODE == MODE_APPLY ? (redmode != RED_NONE ? KC_APPLY_DOT : KC_APPLY)
		                                        : (MODE == MODE_RESID ? KC_RESID : (MODE == MODE_JACOBI ? KC_JACOBI : KC_RESID_RESTRICT))),
		        (size_t) D.count * L.nc);
		auto         grid = [&](int z) { return dim3(8 * ((D.count * z + 7) / 8)); };
		const RedSrc rs{red_a, g->partial.p, D.first * zs}; // (interior patches are launched before the boundary patches)
		switch (zs) {
			case 1: launchStencilZS<N, MODE, 1>(g, grid(1), D, u, f, out, omega, rd, redmode, rs); break;
			case 2:
				if constexpr (N >= 8) launchStencilZS<N, MODE, 2>(g, grid(2), D, u, f, out, omega, rd, redmode, rs);
				break;
			case 8:
				if constexpr (N >= 32) launchStencilZS<N, MODE, 8>(g, grid(8), D, u, f, out, omega, rd, redmode, rs);
				break;
			default:
				if constexpr (N >= 16) launchStencilZS<N, MODE, 4>(g, grid(4), D, u, f, out, omega, rd, redmode, rs);
				break;
		}
	};
	int rc = withGhosts<N>(g, L, u, launch, xf_in);
	if (rc) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}
// ------------------------------------------------------------------------------ 2D launches
int prepareGhosts2d(te_gmg *g, LevelHost &L, const double *u)
{
	if (L.patch_local) return TE_OK;
	if (L.nremote > 0) {
		{
			Timed t(g, KC_PACK, (size_t) L.nremote * L.nf);
			hipLaunchKernelGGL(k_pack_faces2d, dim3(L.nremote), dim3(64), 0, g->stream, L.n, L.send_faces.p, u, L.sendbuf.p);
		}
		int rc = faceExchange(g, L, L.sendbuf.p);
		if (rc) return rc;
	}
	if (L.ncf == 0) return TE_OK;
	Timed t(g, KC_CFGHOST, (size_t) L.ncf * L.nf);
	hipLaunchKernelGGL(k_cf_ghost2d, dim3(L.ncf), dim3(64), 0, g->stream, L.n, L.cf_desc.p, L.cf_slots.p, u, L.ghostCur());
	return TE_OK;
}
template <int MODE> int launchStencil2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, double omega)
{
	int rc = prepareGhosts2d(g, L, u);
	if (rc) return rc;
	Timed t(g, MODE == MODE_APPLY ? KC_APPLY : (MODE == MODE_RESID ? KC_RESID : KC_JACOBI), (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_stencil2d<MODE>, dim3(gridFor((size_t) L.P * L.nc / 2, 256, 65536)), dim3(256), 0, g->stream, L.dev2(),
	                   u, f, out, omega);
	HIPCHK(hipGetLastError());
	return TE_OK;
}
// 64^2 patches: 512 threads per workgroup (four x-pairs per thread instead of eight: half the registers, twice the waves per
// CU at the same four resident patches)
static int tpb2d(const te_gmg *g) { return g->cfg.num(O_2D_TPB, 512) == 256 ? 256 : 512; }
// faces of u + P(coarse) for the neighbours on other ranks (u: the stored iterate, or e4: only its edge layers exist), and
// their values into this rank's ghost slots
int packProlongFaces2d(te_gmg *g, LevelHost &L, const double *u, const double *e4, const Prolong2D &ps)
{
	if (L.nremote == 0 || L.patch_local) return TE_OK;
	{
		Timed t(g, KC_PACK, (size_t) L.nremote * L.nf);
		hipLaunchKernelGGL(k_pack_faces_prolong2d, dim3(L.nremote), dim3(64), 0, g->stream, L.n, L.send_faces.p, u, e4, ps, L.sendbuf.p);
	}
	return faceExchange(g, L, L.sendbuf.p);
}
// zero_guess: levels with L.lds2d; prolong_from: levels with L.fuse2d && L.prolong_fusable (the caller checks)
int launchRbgs2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, bool zero_guess = false,
                 const double *prolong_from = nullptr)
{
	int rc;
	if (!zero_guess && !prolong_from && (rc = prepareGhosts2d(g, L, u))) return rc;
	if (prolong_from && (rc = packProlongFaces2d(g, L, u, nullptr, Prolong2D{L.parent.p, L.orth.p, prolong_from}))) return rc;
	if (L.P == 0) return TE_OK;
	if (L.n <= 64 && !g->cfg.has(O_2D_SIMPLE)) { // the patch and its halo ring fit in LDS: one pass
		const size_t lds = sizeof(double) * ((size_t) (L.n + 2) * (L.n + 2) + 16);
		Prolong2D    ps{L.parent.p, L.orth.p, prolong_from};
		Timed        t(g, zero_guess ? KC_RBGS_ZERO : (prolong_from ? KC_RBGS_PROLONG : KC_RBGS), (size_t) L.P * L.nc, true);
#define TE_RB2(Z, PR)                                                                                                          \
	if (L.n == 64 && tpb2d(g) == 512)                                                                                           \
		launchT(t, (k_rbgs2d_lds<Z, PR, 64, 512>), dim3(L.P), dim3(512), lds, g->stream, L.dev2(), u, f, out, ps);     \
	else if (L.n == 64)                                                                                                        \
		launchT(t, (k_rbgs2d_lds<Z, PR, 64>), dim3(L.P), dim3(256), lds, g->stream, L.dev2(), u, f, out, ps);          \
	else                                                                                                                       \
		launchT(t, (k_rbgs2d_lds<Z, PR, 0>), dim3(L.P), dim3(256), lds, g->stream, L.dev2(), u, f, out, ps)
		if (zero_guess)
			TE_RB2(true, false);
		else if (prolong_from)
			TE_RB2(false, true);
		else
			TE_RB2(false, false);
#undef TE_RB2
		HIPCHK(hipGetLastError());
		return TE_OK;
	}
	if (zero_guess || prolong_from) return te::fail(TE_ESTATE, "launchRbgs2d: fused variants need patches that fit in LDS");
	Timed      t(g, KC_RBGS, (size_t) L.P * L.nc);
	const dim3 grid(gridFor((size_t) L.P * L.nc, 256, 65536));
	hipLaunchKernelGGL(k_rbgs2d<0>, grid, dim3(256), 0, g->stream, L.dev2(), u, f, out);
	hipLaunchKernelGGL(k_rbgs2d<1>, grid, dim3(256), 0, g->stream, L.dev2(), u, f, out);
	HIPCHK(hipGetLastError());
	return TE_OK;
}
// coarse f = AvgRstr(f - A u) in one pass (levels with L.fuse2d)
int residRestrict2d(te_gmg *g, LevelHost &L, const double *u, const double *f, double *coarse)
{
	if (L.P == 0) return TE_OK; // a rank without patches on this level (fuse2d: it has no transfers either)
	int rc = prepareGhosts2d(g, L, u);
	if (rc) return rc;
	const size_t lds = sizeof(double) * ((size_t) (L.n + 2) * (L.n + 2) + 16);
	Timed        t(g, KC_RESID_RESTRICT, (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_resid_restrict2d_lds, dim3(L.P), dim3(256), lds, g->stream, L.dev2(), u, f,
	                   Prolong2D{L.parent.p, L.orth.p, nullptr}, coarse);
	HIPCHK(hipGetLastError());
	return TE_OK;
}
// *swapped: the result went to s1 (= L.t) instead of u: the caller exchanges the two vectors' buffers
int patchSolve2d(te_gmg *g, LevelHost &L, const double *f, double *u, double *s0, double *s1, bool zero_guess, bool *swapped)
{
	*swapped = false;
	int          rc;
	const size_t total = (size_t) L.P * L.nc;
	if (L.n == 64 && L.matsT.p && !g->cfg.has(O_2D_SIMPLE) && !g->cfg.has(O_2D_NO_MFMA)) { // 64^2 patches: the four products on the matrix cores
		if (!zero_guess && (rc = prepareGhosts2d(g, L, u))) return rc;
		const size_t lds = sizeof(double) * 64 * PS2D_LD;
		bool        &attr = g->ps2d_attr;
		const bool   pf = L.P <= 256 && !g->cfg.has(O_2D_NO_PF); // few patches: a workgroup has its CU to itself anyway
		Timed        t(g, KC_PS_MFMA, total, true);
		auto         launch = [&](auto kern) -> int {
            launchT(t, kern, dim3(L.P), dim3(256), lds, g->stream, L.dev2(), L.plan.p, L.matsT.p, L.lam.p, L.zero_mode.p, f, u, s1);
            return TE_OK;
		};
		if (!attr) { // all four once, so that the attribute is set whichever runs first
			const void *ks[4] = {reinterpret_cast<const void *>(k_patch_solve2d_mfma<true, true>), reinterpret_cast<const void *>(k_patch_solve2d_mfma<true, false>),
			                     reinterpret_cast<const void *>(k_patch_solve2d_mfma<false, true>), reinterpret_cast<const void *>(k_patch_solve2d_mfma<false, false>)};
			for (const void *k : ks) HIPCHK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
			attr = true;
		}
		if (zero_guess)
			rc = pf ? launch(k_patch_solve2d_mfma<true, true>) : launch(k_patch_solve2d_mfma<true, false>);
		else
			rc = pf ? launch(k_patch_solve2d_mfma<false, true>) : launch(k_patch_solve2d_mfma<false, false>);
		if (rc) return rc;
		HIPCHK(hipGetLastError());
		*swapped = true;
		return TE_OK;
	}
	if (L.n <= 64 && L.matsT.p && !g->cfg.has(O_2D_SIMPLE)) { // one launch, the patch in LDS
		if (!zero_guess && (rc = prepareGhosts2d(g, L, u))) return rc;
		const size_t lds = sizeof(double) * 2 * L.nc;
		Timed        t(g, KC_PS_MFMA, total);
#define TE_PS2(Z, NC, T)                                                                                                     \
	hipLaunchKernelGGL((k_patch_solve2d_lds<Z, NC, T>), dim3(L.P), dim3(T), lds, g->stream, L.dev2(), L.plan.p, L.mats.p, \
	                   L.matsT.p, L.lam.p, L.zero_mode.p, f, u, s1)
		const bool wide = L.n == 64 && L.P <= 128; // few patches: sixteen waves per patch
		if (zero_guess) {
			if (wide)
				TE_PS2(true, 64, 1024);
			else if (L.n == 64)
				TE_PS2(true, 64, 256);
			else
				TE_PS2(true, 0, 256);
		} else {
			if (wide)
				TE_PS2(false, 64, 1024);
			else if (L.n == 64)
				TE_PS2(false, 64, 256);
			else
				TE_PS2(false, 0, 256);
		}
#undef TE_PS2
		HIPCHK(hipGetLastError());
		*swapped = true;
		return TE_OK;
	}
	if (zero_guess) {
		Timed t(g, KC_VECOP, total);
		HIPCHK(hipMemsetAsync(u, 0, sizeof(double) * total, g->stream));
	}
	if ((rc = prepareGhosts2d(g, L, u))) return rc;
	const dim3   grid(gridFor(total, 256, 65536)), blk(256);
	{
		Timed t(g, KC_PATCH_RHS, total);
		hipLaunchKernelGGL(k_patch_rhs2d, grid, blk, 0, g->stream, L.dev2(), u, f, s0);
	}
	const bool mfma2d = L.n % 16 == 0 && !g->cfg.has(O_2D_SIMPLE); // large patches: the passes on the matrix cores
	const dim3 gridm(((size_t) L.P * (L.n / 16) * (L.n / 16) + 3) / 4);
#define TE_DST2(STAGE, IN, OUT)                                                                                          \
	{                                                                                                                    \
		Timed t(g, KC_DST, total);                                                                                       \
		if (mfma2d)                                                                                                      \
			hipLaunchKernelGGL(k_dst_axis2d_mfma<STAGE>, gridm, blk, 0, g->stream, L.n, L.P, L.plan.p, L.mats.p, L.lam.p, \
			                   L.zero_mode.p, L.rh2.p, IN, OUT);                                                         \
		else                                                                                                             \
			hipLaunchKernelGGL(k_dst_axis2d<STAGE>, grid, blk, 0, g->stream, L.n, L.P, L.plan.p, L.mats.p, L.lam.p,      \
			                   L.zero_mode.p, L.rh2.p, IN, OUT);                                                         \
	}
	TE_DST2(0, s0, s1)
	TE_DST2(1, s1, s0)
	TE_DST2(2, s0, s1)
	TE_DST2(3, s1, u)
#undef TE_DST2
	HIPCHK(hipGetLastError());
	return TE_OK;
}
int restrict2d(te_gmg *g, LevelHost &L, const double *fine, double *coarse)
{
	if (L.n_up > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_up * L.nc / 4);
		hipLaunchKernelGGL(k_restrict_pack2d, dim3(L.n_up), dim3(256), 0, g->stream, L.n, L.up_desc.p, L.up_off.p, fine, L.upbuf.p);
	}
	int rc = doExchange(g, 2, L.tx_up, L.upbuf.p, L.downbuf.p);
	if (rc) return rc;
	if (L.Pc == 0) return TE_OK;
	Timed t(g, KC_RESTRICT, (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_restrict2d, dim3(gridFor((size_t) L.Pc * L.nc, 256, 65536)), dim3(256), 0, g->stream, L.n, L.Pc,
	                   L.child.p, L.copy.p, fine, L.downbuf.p, L.down_off.p, coarse);
	HIPCHK(hipGetLastError());
	return TE_OK;
}
int prolong2d(te_gmg *g, LevelHost &L, const double *coarse, double *fine)
{
	if (L.n_down > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_down * L.nc / 4);
		hipLaunchKernelGGL(k_prolong_pack2d, dim3(L.n_down), dim3(256), 0, g->stream, L.n, L.down_desc.p, L.down_off.p, coarse,
		                   L.downbuf.p);
	}
	int rc = doExchange(g, 3, L.tx_down, L.downbuf.p, L.upbuf.p);
	if (rc) return rc;
	if (L.P == 0) return TE_OK;
	Timed t(g, KC_PROLONG, (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_prolong2d, dim3(gridFor((size_t) L.P * L.nc, 256, 65536)), dim3(256), 0, g->stream, L.n, L.P, L.parent.p,
	                   L.orth.p, coarse, L.upbuf.p, L.up_off.p, fine);
	HIPCHK(hipGetLastError());
	return TE_OK;
}

// opts.fuse = 2 / 3 in 2D (levels with L.fuse2_ok: patches in LDS, every parent and neighbour local): see kernels2d.hpp
int zeroSweepResid2d(te_gmg *g, LevelHost &L, const double *f, double *out, double *coarse, bool store_u)
{
	const size_t lds = sizeof(double) * ((size_t) (L.n + 2) * (L.n + 2) + 16);
	Prolong2D    dst{L.parent.p, L.orth.p, nullptr};
	int          rc;
	if (L.P > 0) {
		Timed t(g, store_u ? KC_ZERO_RESID : KC_ZERO_RESID_FACES, (size_t) L.P * L.nc, true);
#define TE_ZR2(S, NC)                                                                                                             \
	launchT(t, (k_rbgs_zero_resid2d_lds<S, NC>), dim3(L.P), dim3(256), lds, g->stream, L.dev2(), f, out, L.e4buf.p, dst, \
	                   coarse, L.upbuf.p, L.up_off.p)
		if (L.n == 64 && tpb2d(g) == 512) {
			if (store_u)
				launchT(t, (k_rbgs_zero_resid2d_lds<true, 64, 512>), dim3(L.P), dim3(512), lds, g->stream, L.dev2(), f, out, L.e4buf.p, dst,
				                   coarse, L.upbuf.p, L.up_off.p);
			else
				launchT(t, (k_rbgs_zero_resid2d_lds<false, 64, 512>), dim3(L.P), dim3(512), lds, g->stream, L.dev2(), f, out, L.e4buf.p, dst,
				                   coarse, L.upbuf.p, L.up_off.p);
		} else if (store_u && L.n == 64)
			TE_ZR2(true, 64);
		else if (store_u)
			TE_ZR2(true, 0);
		else if (L.n == 64)
			TE_ZR2(false, 64);
		else
			TE_ZR2(false, 0);
#undef TE_ZR2
	}
	if (L.nremote > 0) { // the new edge layers of neighbours on other ranks
		{
			Timed t(g, KC_PACK, (size_t) L.nremote * L.nf);
			if (store_u)
				hipLaunchKernelGGL(k_pack_faces2d, dim3(L.nremote), dim3(64), 0, g->stream, L.n, L.send_faces.p, out, L.sendbuf.p);
			else
				hipLaunchKernelGGL(k_pack_edges2d, dim3(L.nremote), dim3(64), 0, g->stream, L.n, L.send_faces.p, L.e4buf.p, L.sendbuf.p);
		}
		if ((rc = faceExchange(g, L, L.sendbuf.p))) return rc;
	}
	if (L.P > 0) {
		Timed t(g, KC_FIXUP, (size_t) L.P * 4 * L.nf);
		hipLaunchKernelGGL(k_restrict_fixup2d, dim3(L.P), dim3(128), 0, g->stream, L.dev2(), out,
		                   store_u ? (const double *) nullptr : (const double *) L.e4buf.p, dst, coarse, L.upbuf.p, L.up_off.p);
	}
	// children whose parent lives on another rank: ship the finished blocks
	if ((rc = doExchange(g, 2, L.tx_up, L.upbuf.p, L.downbuf.p))) return rc;
	if (L.n_down > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_down * L.nc / 4);
		hipLaunchKernelGGL(k_restrict_unpack2d, dim3(L.n_down), dim3(256), 0, g->stream, L.n, L.down_desc.p, L.down_off.p, L.downbuf.p, coarse);
	}
	HIPCHK(hipGetLastError());
	return TE_OK;
}
int resweepProlong2d(te_gmg *g, LevelHost &L, const double *f, double *out, const double *prolong_from)
{
	int rc = packProlongFaces2d(g, L, nullptr, L.e4buf.p, Prolong2D{L.parent.p, L.orth.p, prolong_from});
	if (rc) return rc;
	if (L.P == 0) return TE_OK;
	const size_t lds = sizeof(double) * ((size_t) (L.n + 2) * (L.n + 2) + 16);
	Timed        t(g, KC_RESWEEP, (size_t) L.P * L.nc, true);
	if (L.n == 64 && tpb2d(g) == 512)
		launchT(t, (k_rbgs_resweep_prolong2d_lds<64, 512>), dim3(L.P), dim3(512), lds, g->stream, L.dev2(), f, L.e4buf.p, out,
		                   Prolong2D{L.parent.p, L.orth.p, prolong_from});
	else if (L.n == 64)
		launchT(t, k_rbgs_resweep_prolong2d_lds<64>, dim3(L.P), dim3(256), lds, g->stream, L.dev2(), f, L.e4buf.p, out,
		                   Prolong2D{L.parent.p, L.orth.p, prolong_from});
	else
		launchT(t, k_rbgs_resweep_prolong2d_lds<0>, dim3(L.P), dim3(256), lds, g->stream, L.dev2(), f, L.e4buf.p, out,
		                   Prolong2D{L.parent.p, L.orth.p, prolong_from});
	HIPCHK(hipGetLastError());
	return TE_OK;
}

template <int MODE> int launchStencil(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, double omega,
                                      RestrictDst rd = RestrictDst(), const double *xf_in = nullptr, int redmode = RED_NONE,
                                      const double *red_a = nullptr, int *red_items = nullptr)
{
	if (red_items) *red_items = 0;
	if (L.P == 0) return TE_OK;
	if (L.dim == 2) {
		if (redmode != RED_NONE) return te::fail(TE_EUNSUPPORTED, "fused sums exist for the 3D stencil kernel only");
		if constexpr (MODE == MODE_RESID_RESTRICT)
			return te::fail(TE_EUNSUPPORTED, "fused residual+restrict has no 2D kernel");
		else
			return launchStencil2d<MODE>(g, L, u, f, out, omega);
	}
	switch (L.n) {
		case 4: return launchStencilN<4, MODE>(g, L, u, f, out, omega, rd, xf_in, redmode, red_a, red_items);
		case 8: return launchStencilN<8, MODE>(g, L, u, f, out, omega, rd, xf_in, redmode, red_a, red_items);
		case 16: return launchStencilN<16, MODE>(g, L, u, f, out, omega, rd, xf_in, redmode, red_a, red_items);
		default: return launchStencilN<32, MODE>(g, L, u, f, out, omega, rd, xf_in, redmode, red_a, red_items);
	}
}
// z-slabs per patch for the RB-GS kernels: enough workgroups to occupy 256 CUs x 4 when the level has few patches
template <int N> inline int rbgsSlabs(const te_gmg *g, int count)
{
	int zs = 1;
	while (zs < 4 && !g->cfg.has(O_RBGS_NOSLAB) && (size_t) count * zs < 1024 && N / (zs * 2) >= 4) zs *= 2;
	// very few patches (the coarsest levels of a cycle): a kernel is one patch's march, a dependent chain of plane steps of
	// ~1-2 us each that nothing hides -- eight slabs of four planes (six steps) instead of four of eight (ten steps)
	if (zs == 4 && N == 32 && count <= 64 && !g->cfg.has(O_NO_ZS8)) zs = 8;
	return zs;
}
template <int N, bool ZERO, bool PROLONG>
void launchRbgsKernel(te_gmg *g, const LevelDev &D, const double *u, const double *f, double *out, const ProlongSrc &ps)
{
	const int  zs = rbgsSlabs<N>(g, D.count);
	const dim3 grid(8 * ((D.count * zs + 7) / 8)), blk(Tile3<N>::TPB);
	if (zs == 8) {
		if constexpr (N >= 32) hipLaunchKernelGGL((k_rbgs3d<N, ZERO, PROLONG, 8>), grid, blk, 0, g->stream, D, u, f, out, ps);
	} else if (zs == 4) {
		if constexpr (N >= 16) hipLaunchKernelGGL((k_rbgs3d<N, ZERO, PROLONG, 4>), grid, blk, 0, g->stream, D, u, f, out, ps);
	} else if (zs == 2) {
		if constexpr (N >= 8) hipLaunchKernelGGL((k_rbgs3d<N, ZERO, PROLONG, 2>), grid, blk, 0, g->stream, D, u, f, out, ps);
	} else {
		hipLaunchKernelGGL((k_rbgs3d<N, ZERO, PROLONG, 1>), grid, blk, 0, g->stream, D, u, f, out, ps);
	}
}
template <int N> int launchRbgsN(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, bool zero_guess,
                                 const double *prolong_from, const double *xf_in, double *xf_out)
{
	if (prolong_from) { // u + P(coarse) is formed on the fly: only for levels without coarse/fine faces (checked by the caller)
		ProlongSrc ps;
		ps.parent = L.parent.p;
		ps.orth   = L.orth.p;
		ps.coarse = prolong_from;
		const bool cfp = (L.ncf > 0 || L.has_copy); // refined level: copy-through patches / coarse-fine ghost slots
		auto launch = [&](LevelDev D) {
			if (D.count == 0) return;
			if (cfp) {
				Timed t(g, KC_RBGS_PROLONG, (size_t) D.count * L.nc);
				hipLaunchKernelGGL((k_rbgs3d<N, false, true, 1, true>), dim3(8 * ((D.count + 7) / 8)), dim3(Tile3<N>::TPB), 0, g->stream,
				                   D, u, f, out, ps);
				return;
			}
			Timed t(g, rbgsSlabs<N>(g, D.count) > 1 ? KC_RBGS_SLABS : KC_RBGS_PROLONG, (size_t) D.count * L.nc);
			launchRbgsKernel<N, false, true>(g, D, u, f, out, ps);
		};
		// neighbours on other ranks receive this rank's face layers of u + P(coarse) (exchange under the interior)
		int rc = withGhosts<N>(g, L, u, launch, xf_in, xf_out, &ps);
		if (rc) return rc;
		HIPCHK(hipGetLastError());
		return TE_OK;
	}
	if (zero_guess) { // a zero iterate has zero ghosts everywhere: nothing to exchange or build
		Timed    t(g, rbgsSlabs<N>(g, L.P) > 1 ? KC_RBGS_SLABS : KC_RBGS_ZERO, (size_t) L.P * L.nc);
		LevelDev D = L.dev();
		D.xf_out   = xf_out;
		launchRbgsKernel<N, true, false>(g, D, u, f, out, ProlongSrc());
		HIPCHK(hipGetLastError());
		return TE_OK;
	}
	auto launch = [&](LevelDev D) {
		if (D.count == 0) return;
		Timed t(g, rbgsSlabs<N>(g, D.count) > 1 ? KC_RBGS_SLABS : KC_RBGS, (size_t) D.count * L.nc);
		launchRbgsKernel<N, false, false>(g, D, u, f, out, ProlongSrc());
	};
	int rc = withGhosts<N>(g, L, u, launch, xf_in, xf_out);
	if (rc) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}
// The restricted blocks of this rank's patches to the other ranks (the kernels before this have written them into the local
// coarse patches or into upbuf), the other ranks' blocks into the local coarse patches. repl_up (the coarse level lives on
// every rank): the finished octants are copied out of the coarse patches first -- one copy, sent to everybody.
template <int N> int shipRestricted(te_gmg *g, LevelHost &L, double *coarse)
{
	// every rank's blocks are whole coarse patches in one run of the coarse vector: run to run, in place
	if (L.repl_up && L.repl_direct && !g->cfg.has(O_REPL_BLOCKS)) {
		if (g->push.on && L.push_blocks && !g->recording) return pushExchange(g, L, 2, coarse);
		return doExchange(g, 2, L.tx_direct, coarse, coarse);
	}
	if (L.repl_up && L.n_up > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_up * L.nc / 8);
		hipLaunchKernelGGL(k_prolong_pack3d<N>, dim3(L.n_up), dim3(256), 0, g->stream, L.bc_desc.p, L.up_off.p, coarse, L.upbuf.p);
	}
	int rc = doExchange(g, 2, L.tx_up, L.upbuf.p, L.downbuf.p);
	if (rc) return rc;
	if (L.n_down > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_down * L.nc / 8);
		hipLaunchKernelGGL(k_restrict_unpack3d<N>, dim3(L.n_down), dim3(256), 0, g->stream, L.down_desc.p, L.down_off.p,
		                   L.downbuf.p, coarse);
	}
	return TE_OK;
}
// opts.fuse = 2: first pre-smoothing sweep from a zero iterate + residual + restriction (march3d.hpp,
// k_rbgs_zero_resid3d / k_restrict_fixup3d). out = S(0, f) with its x faces in xf_out, coarse = AvgRstr(f - A out).
// store_u = false (opts.fuse = 3): the new iterate is left in L.f6buf as its six face layers only
template <int N>
int zeroSweepResidN(te_gmg *g, LevelHost &L, const double *f, double *out, double *coarse, double *xf_out, bool store_u,
                    double *fcorr_out, const double *fcorr_in, const PendingRhs *fs)
{
	RestrictDst rd = RestrictDst();
	rd.parent     = L.parent.p;
	rd.orth       = L.orth.p;
	rd.coarse     = coarse;
	rd.remote     = L.upbuf.p;
	rd.remote_off = L.up_off.p;
	// the patches export their 2x2 face sums only on uniformly refined levels; a refined level's terms are formed by the
	// gather kernel from the face layers
	const bool export_rs6 = fcorr_out && L.prolong_fusable;
	rd.rs6                = export_rs6 ? L.rs6.p : nullptr;
	int rc;
	if (L.P > 0) {
		Timed      t(g, store_u ? KC_ZERO_RESID : (fcorr_in ? KC_ZERO_RESID_FACES_FCORR : KC_ZERO_RESID_FACES), (size_t) L.P * L.nc, true);
		if (!store_u) L.f6_tab = L.f6off.p != nullptr && !g->cfg.has(O_PACK_FACES); // the face layers go where the level's table puts them
		LevelDev   D = L.dev();
		const dim3 grid(8 * ((L.P + 7) / 8)), blk(Tile3<N>::TPB);
		if (store_u) {
			D.xf_out = xf_out;
			launchT(t, (k_rbgs_zero_resid3d<N, true>), grid, blk, 0, g->stream, D, f, out, rd, FSrc());
		} else {
			D.f6_out = L.f6buf.p;
			D.fcorr  = fcorr_in;
			// TE_ZR_AHEAD = 1: the right-hand side requested one plane ahead only (the form before round 3; bit-identical)
			const bool ah1 = g->cfg.num(O_ZR_AHEAD, 4) == 1;
#define TE_ZR(EXP, FC)                                                                                                   \
	if (ah1)                                                                                                             \
		launchT(t, (k_rbgs_zero_resid3d<N, false, EXP, FC, 1>), grid, blk, 0, g->stream, D, f, out, rd, FSrc());         \
	else                                                                                                                 \
		launchT(t, (k_rbgs_zero_resid3d<N, false, EXP, FC, 4>), grid, blk, 0, g->stream, D, f, out, rd, FSrc())
			if (fs) { // the right-hand side is a pending vector statement of te_bicgstab (march3d.hpp FSrc): formed and stored here
				const FSrc a = fs->args;
				if (fs->kind == 1 && export_rs6)
					launchT(t, (k_rbgs_zero_resid3d<N, false, true, false, 4, 1>), grid, blk, 0, g->stream, D, f, out, rd, a);
				else if (fs->kind == 1)
					launchT(t, (k_rbgs_zero_resid3d<N, false, false, false, 4, 1>), grid, blk, 0, g->stream, D, f, out, rd, a);
				else if (export_rs6)
					launchT(t, (k_rbgs_zero_resid3d<N, false, true, false, 4, 2>), grid, blk, 0, g->stream, D, f, out, rd, a);
				else
					launchT(t, (k_rbgs_zero_resid3d<N, false, false, false, 4, 2>), grid, blk, 0, g->stream, D, f, out, rd, a);
			} else if (export_rs6 && fcorr_in) {
				TE_ZR(true, true);
			} else if (export_rs6) {
				TE_ZR(true, false);
			} else if (fcorr_in) {
				TE_ZR(false, true);
			} else {
				TE_ZR(false, false);
			}
#undef TE_ZR
		}
	}
	// the new face layers of neighbours on other ranks (no-op on one rank)
	L.pack_f6 = store_u ? nullptr : L.f6buf.p;
	rc        = prepareGhosts<N>(g, L, out);
	L.pack_f6 = nullptr;
	if (rc) return rc;
	L.ghost_has_v = !store_u; // (the slots of neighbours on other ranks hold their face layers of v until the next exchange of the level)
	if (fcorr_out) { // the ghost terms were formed by the patches that own the face values: sort them into the coarse
		// level's side array (a permutation copy of 6/128 of a vector instead of the fix-up pass)
		if (L.Pc > 0) {
			const bool use_gtab = export_rs6 && !g->cfg.has(O_NO_GTAB);
			if (use_gtab && !L.gtab.p && (size_t) L.P * 6 * (N / 2) * (N / 2) < ((size_t) 1 << 31)) { // once per level
				int rc2 = L.gtab.alloc((size_t) L.Pc * 48);
				if (rc2) return rc2;
				hipLaunchKernelGGL(k_gather_table3d<N>, dim3(L.Pc), dim3(64), 0, g->stream, L.dev(), L.child.p, L.copy.p, L.gtab.p);
			}
			Timed t(g, KC_FCORR_GATHER, (size_t) L.P * 6 * L.nf / 4);
			hipLaunchKernelGGL(k_fcorr_gather3d<N>, dim3(L.Pc * 12), dim3(256), 0, g->stream, L.dev(), L.child.p, L.copy.p,
			                   export_rs6 ? (const double *) L.rs6.p : (const double *) nullptr, (const double *) L.f6buf.p, coarse, fcorr_out,
			                   use_gtab ? (const int32_t *) L.gtab.p : (const int32_t *) nullptr);
		}
	} else if (L.P > 0) {
		Timed    t(g, KC_FIXUP, (size_t) L.P * 6 * L.nf);
		LevelDev D = L.dev();
		if (store_u)
			D.xf = xf_out;
		else
			D.f6 = L.f6buf.p;
		hipLaunchKernelGGL((k_restrict_fixup3d<N, false>), dim3(L.P), dim3(256), 0, g->stream, D, out, rd);
	}
	// children whose parent lives on another rank: ship the finished blocks (as residRestrictN)
	if ((rc = shipRestricted<N>(g, L, coarse))) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}
// opts.fuse = 3, post-smoothing: out = S(v + P(prolong_from), f) with v = S(0, f) recomputed (its faces in L.f6buf)
template <int N>
int resweepProlongN(te_gmg *g, LevelHost &L, const double *f, double *out, const double *prolong_from, double *xf_out, const double *fcorr_in)
{
	ProlongSrc ps;
	ps.parent = L.parent.p;
	ps.orth   = L.orth.p;
	ps.coarse = prolong_from;
	auto launch = [&](LevelDev D) {
		if (D.count == 0) return;
		Timed t(g, fcorr_in ? KC_RESWEEP_FCORR : KC_RESWEEP, (size_t) D.count * L.nc, true);
		D.f6    = L.f6buf.p;
		D.fcorr = fcorr_in;
		if constexpr (N >= 4) {
			const dim3  grid(8 * ((D.count + 7) / 8)), blk(Tile3<N>::TPB);
			// tuning variants (march3d.hpp), all bit-identical. Defaults, each measured: the finest level stores u non-temporally
			// (nobody re-reads it) and loads f non-temporally unless f can still be in the Infinity Cache from the pre-sweep
			// that read it (19 instead of 27: 256^3, a rank's share at eight ranks; 53.4 -> 49.0 us at 256^3); a large finest
			// level runs two workgroups per CU with four planes of f in flight each instead of three with two (59: 450 -> 443 us
			// at 512^3, same box; slower at 256^3); a coarser level with exported ghost terms (level 1 of 512^3) keeps ordinary
			// stores as well -- its u is the correction the finer level's post-sweep reads next (3: 64.5 us, against 68.7 with
			// non-temporal stores)
			const char *ve    = g->cfg.str(O_RESWEEP_V);
			const bool  small = (size_t) L.P * L.nc * sizeof(double) <= ((size_t) 160 << 20);
			const int   v     = ve ? atoi(ve) : (fcorr_in ? 3 : (g->cur_level != 0 ? 27 : (small ? 19 : 59)));
			if (L.ncf > 0 || L.has_copy) { // refined level: copy-through patches / coarse-fine ghost slots
				if (v == 3)
					launchT(t, (k_rbgs_resweep_prolong3d<N, 3, false, true>), grid, blk, 0, g->stream, D, f, out, ps);
				else
					launchT(t, (k_rbgs_resweep_prolong3d<N, 27, false, true>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (fcorr_in) {
				if (v == 0)
					launchT(t, (k_rbgs_resweep_prolong3d<N, 0, true>), grid, blk, 0, g->stream, D, f, out, ps);
				else if (v == 27)
					launchT(t, (k_rbgs_resweep_prolong3d<N, 27, true>), grid, blk, 0, g->stream, D, f, out, ps);
				else if (v == 19)
					launchT(t, (k_rbgs_resweep_prolong3d<N, 19, true>), grid, blk, 0, g->stream, D, f, out, ps);
				else
					launchT(t, (k_rbgs_resweep_prolong3d<N, 3, true>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 0) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 0, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 7) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 7, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 11) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 11, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 19) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 19, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 59) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 59, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 63) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 63, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 23) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 23, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 31) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 31, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 27) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 27, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else if (v == 3) {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 3, false>), grid, blk, 0, g->stream, D, f, out, ps);
			} else {
				launchT(t, (k_rbgs_resweep_prolong3d<N, 27, false>), grid, blk, 0, g->stream, D, f, out, ps);
			}
		}
	};
	if (L.post_exchange_free && L.ghost_has_v && L.ncf == 0 && !L.has_copy && !g->cfg.has(O_POST_EXCHANGE)) {
		// every neighbour's parent is local (the coarser level lives on every rank) and the ghost slots still hold the neighbours'
		// face layers of v: the kernel forms v + P e for them as for local neighbours; nothing travels (a global decision: all
		// ranks of the level take it together)
		ps.gparent    = L.slot_parent.p;
		ps.gorth      = L.slot_orth.p;
		L.ghost_has_v = false;
		LevelDev D    = L.dev();
		D.xf_out      = xf_out;
		launch(D);
		HIPCHK(hipGetLastError());
		return TE_OK;
	}
	L.pack_f6 = L.f6buf.p; // neighbours on other ranks receive the face layers of v + P(coarse)
	int rc    = withGhosts<N>(g, L, out /* unused: the faces come from pack_f6 */, launch, nullptr, xf_out, &ps);
	L.pack_f6 = nullptr;
	if (rc) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}
int resweepProlong(te_gmg *g, LevelHost &L, const double *f, double *out, const double *prolong_from, double *xf_out,
                   const double *fcorr_in)
{
	if (L.dim == 2) return resweepProlong2d(g, L, f, out, prolong_from);
	switch (L.n) {
		case 4: return resweepProlongN<4>(g, L, f, out, prolong_from, xf_out, fcorr_in);
		case 8: return resweepProlongN<8>(g, L, f, out, prolong_from, xf_out, fcorr_in);
		case 16: return resweepProlongN<16>(g, L, f, out, prolong_from, xf_out, fcorr_in);
		default: return resweepProlongN<32>(g, L, f, out, prolong_from, xf_out, fcorr_in);
	}
}
// opts.fuse = 2 with the block-Jacobi smoother: after an exact patch solve from the zero iterate the residual
// vanishes inside every patch (A_patch u = f is what was solved) and equals -(g + m)/h^2 = -2 gamma/h^2 on the face
// layers (the patch operator closes interface faces with ghost = -m, the level operator with the neighbour's g), so
// coarse f = AvgRstr(f - A u) is k_restrict_fixup3d<OWN> applied to a zeroed coarse vector: no pass over u and f
// at all. (What is dropped is the rounding noise of the solve, ~1e-13 |f|.) u: the new iterate, xf: its
// compact x faces or null; coarse: the coarse level's f with `coarse_n` entries.
template <int N> int interfaceResidRestrictN(te_gmg *g, LevelHost &L, const double *u, const double *xf, double *coarse, size_t coarse_n)
{
	RestrictDst rd = RestrictDst();
	rd.parent     = L.parent.p;
	rd.orth       = L.orth.p;
	rd.coarse     = coarse;
	rd.remote     = L.upbuf.p;
	rd.remote_off = L.up_off.p;
	int rc;
	L.pack_f6 = L.ps_faces ? L.f6buf.p : nullptr; // (the iterate exists only as its face layers)
	rc        = prepareGhosts<N>(g, L, u);
	L.pack_f6 = nullptr;
	if (rc) return rc;
	{
		Timed t(g, KC_VECOP, coarse_n);
		// (blocks exchanged in place: only this rank's run -- the others' runs are received, and with the direct-store transport a
		// peer that is ahead may have stored its run already)
		if (L.repl_up && L.repl_direct && !g->cfg.has(O_REPL_BLOCKS) && !L.tx_direct.empty()) {
			if (L.tx_direct.send_cnt[0] > 0)
				HIPCHK(hipMemsetAsync(coarse + L.tx_direct.send_off[0], 0, sizeof(double) * (size_t) L.tx_direct.send_cnt[0], g->stream));
		} else
			HIPCHK(hipMemsetAsync(coarse, 0, sizeof(double) * coarse_n, g->stream));
		if (L.n_up > 0) HIPCHK(hipMemsetAsync(L.upbuf.p, 0, sizeof(double) * L.upbuf.n, g->stream));
	}
	if (L.P > 0) {
		Timed    t(g, KC_FIXUP, (size_t) L.P * 6 * L.nf);
		LevelDev D = L.dev();
		D.xf       = L.ps_faces ? nullptr : xf;
		D.f6       = L.ps_faces ? L.f6buf.p : nullptr;
		hipLaunchKernelGGL((k_restrict_fixup3d<N, true>), dim3(L.P), dim3(256), 0, g->stream, D, u, rd);
	}
	if ((rc = shipRestricted<N>(g, L, coarse))) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}
int interfaceResidRestrict(te_gmg *g, LevelHost &L, const double *u, const double *xf, double *coarse, size_t coarse_n)
{
	switch (L.n) {
		case 4: return interfaceResidRestrictN<4>(g, L, u, xf, coarse, coarse_n);
		case 8: return interfaceResidRestrictN<8>(g, L, u, xf, coarse, coarse_n);
		case 16: return interfaceResidRestrictN<16>(g, L, u, xf, coarse, coarse_n);
		default: return interfaceResidRestrictN<32>(g, L, u, xf, coarse, coarse_n);
	}
}
int zeroSweepResid(te_gmg *g, LevelHost &L, const double *f, double *out, double *coarse, double *xf_out, bool store_u,
                   double *fcorr_out = nullptr, const double *fcorr_in = nullptr, const PendingRhs *fs = nullptr)
{
	if (L.dim == 2) return zeroSweepResid2d(g, L, f, out, coarse, store_u);
	switch (L.n) {
		case 4: return zeroSweepResidN<4>(g, L, f, out, coarse, xf_out, store_u, fcorr_out, fcorr_in, fs);
		case 8: return zeroSweepResidN<8>(g, L, f, out, coarse, xf_out, store_u, fcorr_out, fcorr_in, fs);
		case 16: return zeroSweepResidN<16>(g, L, f, out, coarse, xf_out, store_u, fcorr_out, fcorr_in, fs);
		default: return zeroSweepResidN<32>(g, L, f, out, coarse, xf_out, store_u, fcorr_out, fcorr_in, fs);
	}
}
int launchRbgs(te_gmg *g, LevelHost &L, const double *u, const double *f, double *out, bool zero_guess = false,
               const double *prolong_from = nullptr, const double *xf_in = nullptr, double *xf_out = nullptr)
{
	if (L.P == 0) return TE_OK;
	if (L.dim == 2) return launchRbgs2d(g, L, u, f, out, zero_guess, prolong_from);
	switch (L.n) {
		case 4: return launchRbgsN<4>(g, L, u, f, out, zero_guess, prolong_from, xf_in, xf_out);
		case 8: return launchRbgsN<8>(g, L, u, f, out, zero_guess, prolong_from, xf_in, xf_out);
		case 16: return launchRbgsN<16>(g, L, u, f, out, zero_guess, prolong_from, xf_in, xf_out);
		default: return launchRbgsN<32>(g, L, u, f, out, zero_guess, prolong_from, xf_in, xf_out);
	}
}
// Cycle.h:59-65 in one pass: coarse f = AvgRstr(f - A u), r never stored. Children whose parent is
// on another rank write their block into upbuf; received blocks are placed by k_restrict_unpack3d.
template <int N> int residRestrictN(te_gmg *g, LevelHost &L, const double *u, const double *f, double *coarse, const double *xf_in)
{
	RestrictDst rd = RestrictDst();
	rd.parent     = L.parent.p;
	rd.orth       = L.orth.p;
	rd.coarse     = coarse;
	rd.remote     = L.upbuf.p;
	rd.remote_off = L.up_off.p;
	int rc        = TE_OK;
	if (L.P > 0) rc = launchStencilN<N, MODE_RESID_RESTRICT>(g, L, u, f, L.r->d, 0.0, rd, xf_in);
	if (rc) return rc;
	if ((rc = shipRestricted<N>(g, L, coarse))) return rc;
	HIPCHK(hipGetLastError());
	return TE_OK;
}
int residRestrict(te_gmg *g, LevelHost &L, const double *u, const double *f, double *coarse, const double *xf_in = nullptr)
{
	if (L.dim == 2) return residRestrict2d(g, L, u, f, coarse);
	switch (L.n) {
		case 4: return residRestrictN<4>(g, L, u, f, coarse, xf_in);
		case 8: return residRestrictN<8>(g, L, u, f, coarse, xf_in);
		case 16: return residRestrictN<16>(g, L, u, f, coarse, xf_in);
		default: return residRestrictN<32>(g, L, u, f, coarse, xf_in);
	}
}
// x-face columns of `d`, if the level still holds them (RB-GS sweeps and the single-pass patch solve produce them, inside te_vcycle)
inline const double *xfFor(LevelHost &L, const double *d) { return (d && L.xf_valid_for == d) ? L.xfbuf[L.xf_cur].p : nullptr; }
// the sweep wrote `out` together with its x-face columns into the other buffer: make them current
inline void xfProduced(LevelHost &L, const double *out)
{
	L.xf_cur ^= 1;
	L.xf_valid_for = out;
}
template <int N> int patchSolveN(te_gmg *g, LevelHost &L, const double *f, double *u, double *s0, double *s1,
                                 bool zero_guess, const double *prolong_from)
{
	const size_t total = (size_t) L.P * L.nc;
	int          rc;
	const bool   faces_req = L.ps_faces_req && zero_guess; // (a request holds for the very next sweep only)
	L.ps_faces_req         = false;
	if (!g->cfg.has(O_PS_SLOW)) { // (3D patches are 4, 8, 16 or 32 cells wide)
		// matrix-core path (patchsolve32.hpp; 16^3 patches: patchsolve16.hpp): interface terms on the face layers only, then x,y forward
		// per plane; z forward + eigenvalue divide + z inverse; x,y inverse. A zero initial guess has no
		// interface term (gamma = 0) and u is overwritten without being read.
		// few patches: the three-pass kernels, each patch spread over `seg` workgroups (one patch per CU would
		// leave most of the chip idle and a single solve takes ~80 us); otherwise the single-pass kernel
		// (TE_PS_MODE = 1pass | 1pass-dense | 3pass pins the choice; 3pass also pins one workgroup per patch: tests)
		const char *mode     = g->cfg.str(O_PS_MODE);
		// (the GLOBAL patch count decides: k_ps_sym and the three-pass kernels differ in the last bits, and a sharded run must
		// take the arithmetic path of the single-rank run -- 512^3 on 8 ranks has 64 local patches of 512 on level 1)
		const bool  one_pass = mode ? !strncmp(mode, "1pass", 5) : L.P_global >= 256;
		const int   seg      = (one_pass || mode) ? 1 : (L.P >= 128 ? 2 : (L.P >= 64 ? 4 : 8));
		const dim3 gp(L.P, seg), b256(256);
		// x-face columns of the old iterate, if its producer exported them (te_vcycle only: see xfFor)
		const double *xf_in = (g->in_cycle && !zero_guess) ? xfFor(L, u) : nullptr;
		L.xf_valid_for      = nullptr; // u is rewritten in place
		if (!zero_guess) {
			ProlongSrc ps;
			ps.parent = L.parent.p;
			ps.orth   = L.orth.p;
			ps.coarse = prolong_from;
			L.pack_f6 = L.ps_faces ? L.f6buf.p : nullptr;
			rc        = prepareGhosts<N>(g, L, u, prolong_from ? &ps : nullptr);
			L.pack_f6 = nullptr;
			if (rc) return rc;
			Timed    t(g, KC_PATCH_RHS, (size_t) L.P * 6 * L.nf);
			LevelDev D = L.dev();
			D.xf       = L.ps_faces ? nullptr : xf_in;
			D.f6       = L.ps_faces ? L.f6buf.p : nullptr;
			L.ps_faces = false; // (this sweep rewrites the whole iterate)
			if (prolong_from)
				hipLaunchKernelGGL((k_face_corr3d<N, true>), dim3(L.P * 6), b256, 0, g->stream, D, u, L.corr.p, ps);
			else
				hipLaunchKernelGGL((k_face_corr3d<N, false>), dim3(L.P * 6), b256, 0, g->stream, D, u, L.corr.p, ps);
		}
		if constexpr (N <= 8) { // 4^3 / 8^3 patches: one launch as well (vector units: k_ps_small)
			Timed t(g, KC_DST, total, true);
			if (zero_guess)
				launchT(t, (k_ps_small<N, false>), dim3(L.P), b256, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, L.zero_mode.p, L.rh2.p, f,
				        (const double *) nullptr, u);
			else
				launchT(t, (k_ps_small<N, true>), dim3(L.P), b256, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, L.zero_mode.p, L.rh2.p, f,
				        (const double *) L.corr.p, u);
			HIPCHK(hipGetLastError());
			return TE_OK;
		}
		if constexpr (N == 16) { // the whole solve of a 16^3 patch in one launch, the patch in LDS (k_ps16)
			Timed t(g, KC_PS_MFMA, total, true);
			if (zero_guess)
				launchT(t, k_ps16<false>, dim3(L.P), b256, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, L.zero_mode.p, L.rh2.p, f,
				        (const double *) nullptr, u);
			else
				launchT(t, k_ps16<true>, dim3(L.P), b256, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, L.zero_mode.p, L.rh2.p, f,
				        (const double *) L.corr.p, u);
			HIPCHK(hipGetLastError());
			return TE_OK;
		}
		if (one_pass) { // the whole solve in one pass over HBM (k_ps_fused)
			bool &lds_ok = g->ps_lds_ok;
			int  &ncu    = g->ncu;
			if (!lds_ok) {
				int dev = 0;
				HIPCHK(hipGetDevice(&dev));
				HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
				HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_fused<false>),
				                           hipFuncAttributeMaxDynamicSharedMemorySize, PSF_LDS_BYTES));
				HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_fused<true>),
				                           hipFuncAttributeMaxDynamicSharedMemorySize, PSF_LDS_BYTES));
				HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_sym<false>),
				                           hipFuncAttributeMaxDynamicSharedMemorySize, PSS_LDS_BYTES));
				HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_sym<true>),
				                           hipFuncAttributeMaxDynamicSharedMemorySize, PSS_LDS_BYTES));
				HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_sym<false, true>),
				                           hipFuncAttributeMaxDynamicSharedMemorySize, PSS_LDS_BYTES));
				lds_ok = true;
			}
			// (decided here as well as below: the class of the launch is part of its timing scope)
			const bool dense_only0 = mode && !strcmp(mode, "1pass-dense");
			const bool faces0 = faces_req && L.f6buf.p && !dense_only0 && (L.sym_ok ? L.P : L.n_pure) == L.P;
			Timed         t(g, faces0 ? KC_PS_MFMA_FACES : KC_PS_MFMA, total, true);
			const dim3    b512(512);
			const double *cp = zero_guess ? (const double *) nullptr : (const double *) L.corr.p;
			// pure axes: half-size transforms, one resident workgroup per CU walks over the patches (k_ps_sym);
			// patches with a mixed Dirichlet/Neumann axis: full transforms, one workgroup per patch (k_ps_fused)
			const bool dense_only = mode && !strcmp(mode, "1pass-dense");
			const int  n_sym = dense_only ? 0 : (L.sym_ok ? L.P : L.n_pure), n_mix = L.P - n_sym;
			const int32_t *lst_sym = (n_sym > 0 && n_mix > 0) ? L.ps_list.p : nullptr;
			const int32_t *lst_mix = (n_sym > 0 && n_mix > 0) ? L.ps_list.p + n_sym : nullptr;
			if (n_sym > 0) {
				const dim3 gs(std::min(n_sym, ncu));
				double    *xo = (g->in_cycle && n_mix == 0 && !g->no_xf_export) ? L.xfbuf[L.xf_cur ^ 1].p : nullptr; // (k_ps_fused does not export)
				const bool faces = faces_req && n_mix == 0 && L.f6buf.p;
				if (faces) { // only the face layers of the result: see k_ps_sym<CORR, FACES>
					L.f6_tab = false; // (written as [p][6])
					launchT(t, (k_ps_sym<false, true>), gs, b512, PSS_LDS_BYTES, g->stream, n_sym, L.plan.p, L.matsym.p, L.lam.p,
					        L.zero_mode.p, L.rh2.p, f, cp, u, (double *) nullptr, lst_sym, L.f6buf.p);
					L.ps_faces = true;
					xo         = nullptr;
				} else if (zero_guess)
					launchT(t, (k_ps_sym<false, false>), gs, b512, PSS_LDS_BYTES, g->stream, n_sym, L.plan.p, L.matsym.p, L.lam.p,
					        L.zero_mode.p, L.rh2.p, f, cp, u, xo, lst_sym, (double *) nullptr);
				else
					launchT(t, (k_ps_sym<true, false>), gs, b512, PSS_LDS_BYTES, g->stream, n_sym, L.plan.p, L.matsym.p, L.lam.p,
					        L.zero_mode.p, L.rh2.p, f, cp, u, xo, lst_sym, (double *) nullptr);
				if (xo) xfProduced(L, u);
			}
			if (n_mix > 0) {
				const dim3 gf(8 * ((n_mix + 7) / 8));
				if (zero_guess)
					launchT(t, k_ps_fused<false>, gf, b512, PSF_LDS_BYTES, g->stream, n_mix, L.plan.p, L.mats.p, L.lam.p,
					                   L.zero_mode.p, L.rh2.p, f, cp, u, lst_mix);
				else
					launchT(t, k_ps_fused<true>, gf, b512, PSF_LDS_BYTES, g->stream, n_mix, L.plan.p, L.mats.p, L.lam.p,
					                   L.zero_mode.p, L.rh2.p, f, cp, u, lst_mix);
			}
			HIPCHK(hipGetLastError());
			return TE_OK;
		}
		{
			Timed t(g, KC_PS_3PASS, total);
			if (zero_guess)
				hipLaunchKernelGGL((k_ps_xy<false, false>), gp, b256, 0, g->stream, L.P, L.plan.p, L.mats.p, f,
				                   (const double *) nullptr, s1);
			else
				hipLaunchKernelGGL((k_ps_xy<false, true>), gp, b256, 0, g->stream, L.P, L.plan.p, L.mats.p, f,
				                   (const double *) L.corr.p, s1);
		}
		{
			Timed t(g, KC_PS_3PASS, total);
			hipLaunchKernelGGL(k_ps_z, gp, b256, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, L.zero_mode.p, L.rh2.p, s1, s0);
		}
		{
			Timed t(g, KC_PS_3PASS, total);
			hipLaunchKernelGGL(k_ps_xy<true>, gp, b256, 0, g->stream, L.P, L.plan.p, L.mats.p, s0, (const double *) nullptr, u);
		}
		HIPCHK(hipGetLastError());
		return TE_OK;
	}
	L.xf_valid_for = nullptr; // u is rewritten in place
	if (zero_guess) {
		Timed t(g, KC_VECOP, total);
		HIPCHK(hipMemsetAsync(u, 0, sizeof(double) * total, g->stream));
	}
	if ((rc = prepareGhosts<N>(g, L, u))) return rc;
	{
		Timed t(g, KC_PATCH_RHS, total);
		hipLaunchKernelGGL(k_patch_rhs3d<N>, dim3(gridFor(total, 256)), dim3(256), 0, g->stream, L.dev(), u, f, s0);
	}
	constexpr int BPP = (N * N * N + 255) / 256;
	const dim3    grid(L.P * BPP), blk(256);
#define TE_DST(STAGE, IN, OUT)                                                                                \
	{                                                                                                         \
		Timed t(g, KC_DST, total);                                                                                 \
		hipLaunchKernelGGL((k_dst_axis3d<N, STAGE>), grid, blk, 0, g->stream, L.P, L.plan.p, L.mats.p, L.lam.p, \
		                   L.zero_mode.p, L.rh2.p, IN, OUT);                                                  \
	}
	TE_DST(0, s0, s1)
	TE_DST(1, s1, s0)
	TE_DST(2, s0, s1)
	TE_DST(3, s1, s0)
	TE_DST(4, s0, s1)
	TE_DST(5, s1, u)
#undef TE_DST
	HIPCHK(hipGetLastError());
	return TE_OK;
}
int patchSolve(te_gmg *g, LevelHost &L, const double *f, double *u, bool zero_guess = false,
               const double *prolong_from = nullptr, bool *swapped = nullptr)
{
	bool dummy;
	if (!swapped) swapped = &dummy;
	*swapped = false;
	if (L.P == 0) return TE_OK;
	double *s0 = L.r->d, *s1 = L.t->d;
	if (L.dim == 2) {
		int rc = patchSolve2d(g, L, f, u, s0, s1, zero_guess, swapped);
		if (rc == TE_OK && *swapped && swapped == &dummy) return te::fail(TE_ESTATE, "patchSolve: 2D result left in scratch");
		return rc;
	}
	switch (L.n) {
		case 4: return patchSolveN<4>(g, L, f, u, s0, s1, zero_guess, prolong_from);
		case 8: return patchSolveN<8>(g, L, f, u, s0, s1, zero_guess, prolong_from);
		case 16: return patchSolveN<16>(g, L, f, u, s0, s1, zero_guess, prolong_from);
		default: return patchSolveN<32>(g, L, f, u, s0, s1, zero_guess, prolong_from);
	}
}
template <int N> int restrictN(te_gmg *g, LevelHost &L, const double *fine, double *coarse)
{
	if (L.n_up > 0) {
		Timed t(g, KC_PACK, (size_t) L.n_up * L.nc / 8);
		hipLaunchKernelGGL(k_restrict_pack3d<N>, dim3(L.n_up), dim3(256), 0, g->stream, L.up_desc.p, L.up_off.p, fine,
		                   L.upbuf.p);
	}
	int rc = doExchange(g, 2, L.tx_up, L.upbuf.p, L.downbuf.p);
	if (rc) return rc;
	if (L.Pc == 0) return TE_OK;
	Timed t(g, KC_RESTRICT, (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_restrict3d<N>, dim3(gridFor((size_t) L.Pc * L.nc, 256)), dim3(256), 0, g->stream, L.Pc,
	                   L.child.p, L.copy.p, fine, L.downbuf.p, L.down_off.p, coarse);
	HIPCHK(hipGetLastError());
	return TE_OK;
}
template <int N> int prolongN(te_gmg *g, LevelHost &L, const double *coarse, double *fine)
{
	if (L.n_down > 0 && !L.repl_up) { // (repl_up: those blocks were received for the restriction; every parent is local)
		Timed t(g, KC_PACK, (size_t) L.n_down * L.nc / 8);
		hipLaunchKernelGGL(k_prolong_pack3d<N>, dim3(L.n_down), dim3(256), 0, g->stream, L.down_desc.p, L.down_off.p,
		                   coarse, L.downbuf.p);
	}
	int rc = doExchange(g, 3, L.tx_down, L.downbuf.p, L.upbuf.p);
	if (rc) return rc;
	if (L.P == 0) return TE_OK;
	Timed t(g, KC_PROLONG, (size_t) L.P * L.nc);
	hipLaunchKernelGGL(k_prolong3d<N>, dim3(gridFor((size_t) L.P * L.nc / 2, 256)), dim3(256), 0, g->stream, L.P,
	                   L.parent.p, L.orth.p, coarse, L.upbuf.p, L.up_off.p, fine);
	HIPCHK(hipGetLastError());
	return TE_OK;
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

void swapData(te_vec *a, te_vec *b) { std::swap(a->d, b->d); }

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
		default: return te::fail(TE_EINVAL, "te_smooth: unknown smoother");
	}
}
int doRestrict(te_gmg *g, int fine_level, const double *fine, double *coarse)
{
	LevelHost &L = *g->levels[fine_level];
	if (L.dim == 2) return restrict2d(g, L, fine, coarse);
	switch (L.n) {
		case 4: return restrictN<4>(g, L, fine, coarse);
		case 8: return restrictN<8>(g, L, fine, coarse);
		case 16: return restrictN<16>(g, L, fine, coarse);
		default: return restrictN<32>(g, L, fine, coarse);
	}
}
int doProlong(te_gmg *g, int fine_level, const double *coarse, double *fine)
{
	LevelHost &L = *g->levels[fine_level];
	if (L.dim == 2) return prolong2d(g, L, coarse, fine);
	switch (L.n) {
		case 4: return prolongN<4>(g, L, coarse, fine);
		case 8: return prolongN<8>(g, L, coarse, fine);
		case 16: return prolongN<16>(g, L, coarse, fine);
		default: return prolongN<32>(g, L, coarse, fine);
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
					g->no_xf_export = last;
					r               = patchSolve(g, L, f->d, u->d, false, c);
					g->no_xf_export = false;
					if (r) return r;
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
                || (o->smoother == TE_SMOOTH_PATCH_SOLVE && L.dim == 3 && !g->cfg.has(O_PS_SLOW)))) {
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
		// (the kernel variants that form a pending right-hand side exist for the 3D path that does not store the iterate)
		const PendingRhs *fs = (pend && u_unstored && L.dim == 3 && !fcorr_in && L.P > 0 && !g->recording) ? pend : nullptr;
		if (fs)
			pend = nullptr;
		else if ((rc = formRhs()))
			return rc;
		if ((rc = zeroSweepResid(g, L, f->d, L.t->d, C.f->d, L.xfbuf[L.xf_cur ^ 1].p, !u_unstored, fcorr_out, fcorr_in, fs))) return rc;
		C.f_has_corr = fcorr_out != nullptr;
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
	} else if (fcorr_in) {
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
} // namespace

// Nothing may unwind into a C caller (ctypes, the reference's C++ built with other flags): every int-returning entry
// point below runs inside this barrier. std::bad_alloc and friends come from the std::vector / std::map set-up code.
template <class F> static int guarded(F body) noexcept
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

extern "C" {
void te_cycle_opts_default(te_cycle_opts *o)
{
	if (!o) return;
	o->pre_sweeps = o->post_sweeps = o->coarse_sweeps = o->mid_sweeps = 1; // CycleOpts.h:64-79
	o->cycle_type   = 0;
	o->smoother     = TE_SMOOTH_PATCH_SOLVE;
	o->omega        = 6.0 / 7.0;
	o->exact_coarse = 1;
	o->fuse         = 3;
}

int te_gmg_create(const te_hier *h, int device, te_gmg **out)
{
	return guarded([&]() -> int {
		if (!h || !out) return te::fail(TE_EINVAL, "te_gmg_create: null argument");
		int ndev = 0;
		if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
			return te::fail(TE_EHIP, "te_gmg_create: no HIP device visible (this library has no CPU fallback)");
		if (device < 0) HIPCHK(hipGetDevice(&device));
		HIPCHK(hipSetDevice(device));
		auto g    = std::make_unique<te_gmg>();
		g->device = device;
		g->dim    = h->h.dim;
		g->n      = h->h.n;
		g->rank   = h->h.rank;
		g->nranks = h->h.nranks;
		g->placement[0] = h->h.agglomerate, g->placement[1] = h->h.agglomerate_max, g->placement[2] = h->h.replicate;
		g->placement[3] = (double) h->h.levels.size();
		memset(g->calls, 0, sizeof(g->calls));
		memset(g->cells, 0, sizeof(g->cells));
		memset(g->total_ms, 0, sizeof(g->total_ms));
		HIPCHK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
		HIPCHK(hipStreamCreateWithFlags(&g->comm_stream, hipStreamNonBlocking));
		HIPCHK(hipEventCreateWithFlags(&g->ev_pack, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&g->ev_recv, hipEventDisableTiming));
		g->cfg.fromEnv();
		g->overlap = !g->cfg.has(O_NO_OVERLAP);
		int rc;
		for (int li = 0; li < (int) h->h.levels.size(); li++)
			if ((rc = buildLevel(g.get(), h->h, li))) return rc;
		{ // partial sums: the reduction kernels' blocks, or one pair per work item of a stencil launch with fused sums (<= 8 slabs per patch)
			size_t items = (size_t) g->red_blocks;
			for (auto &L : g->levels) items = std::max(items, (size_t) L->P * (L->P <= 64 ? 8 : (L->P < 2048 ? 4 : 1)));
			if ((rc = g->partial.alloc(2 * items)) || (rc = g->result.alloc(8))) return rc;
		}
		HIPCHK(hipHostMalloc((void **) &g->result_host, 8 * sizeof(double), hipHostMallocDefault));
		for (int li = 0; li < (int) g->levels.size(); li++) {
			LevelHost &L = *g->levels[li];
			te_vec    *v;
			if ((rc = newVec(g.get(), li, &v))) return rc;
			L.r.reset(v);
			if ((rc = newVec(g.get(), li, &v))) return rc;
			L.t.reset(v);
			if (li > 0) {
				if ((rc = newVec(g.get(), li, &v))) return rc;
				L.u.reset(v);
				if ((rc = newVec(g.get(), li, &v))) return rc;
				L.f.reset(v);
			}
		}
		HIPCHK(hipStreamSynchronize(g->stream));
		*out = g.release();
		return TE_OK;
	});
}
void te_gmg_destroy(te_gmg *g)
{
	if (!g) return;
	watchdogStop(g);
	(void) hipStreamSynchronize(g->stream);
	if (g->comm_stream) (void) hipStreamSynchronize(g->comm_stream);
	for (void *m : g->push.opened) (void) hipIpcCloseMemHandle(m);
	for (size_t l = 0; l + 1 < g->levels.size(); l++) // (the coarse vectors own their first buffer, the level its second)
		if (g->levels[l]->cf_buf[0]) g->levels[l + 1]->f->d = g->levels[l]->cf_buf[0];
	if (g->push.flags) (void) hipFree(g->push.flags);
	if (g->push.err) (void) hipFree(g->push.err);
	if (g->push.err_host) (void) hipHostFree(g->push.err_host);
	for (auto &L : g->levels) {
		for (te_vec *v : {L->u.get(), L->f.get(), L->r.get(), L->t.get()})
			if (v && v->d) (void) hipFree(v->d);
	}
	for (te_vec *v : g->bicg_work)
		if (v) te_vec_destroy(v);
	for (auto &e : g->ev_pool) {
		(void) hipEventDestroy(e.a);
		(void) hipEventDestroy(e.b);
	}
	if (g->rccl.comm && g->rccl.CommDestroy) (void) g->rccl.CommDestroy(g->rccl.comm);
	if (g->result_host) (void) hipHostFree(g->result_host);
	if (g->ev_pack) (void) hipEventDestroy(g->ev_pack);
	if (g->ev_recv) (void) hipEventDestroy(g->ev_recv);
	if (g->comm_stream) (void) hipStreamDestroy(g->comm_stream);
	(void) hipStreamDestroy(g->stream);
	delete g;
}
int   te_gmg_num_levels(const te_gmg *g) { return guarded([&]() -> int { return g ? (int) g->levels.size() : TE_EINVAL; }); }
int   te_gmg_sync(te_gmg *g)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_sync: null");
		HIPCHK(hipStreamSynchronize(g->stream));
		return TE_OK;
	});
}
void *te_gmg_stream(te_gmg *g) { return g ? (void *) g->stream : nullptr; }
int   te_gmg_set_exchange(te_gmg *g, te_exchange_fn fn, void *user)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_set_exchange: null");
		if (fn && g->rccl.comm) { // an explicit callback replaces the native RCCL back-end
			(void) g->rccl.CommDestroy(g->rccl.comm);
			g->rccl.comm = nullptr;
		}
		g->exchange      = fn;
		g->exchange_user = user;
		if (fn) watchdogStart(g);
		return TE_OK;
	});
}
int te_gmg_set_allreduce(te_gmg *g, te_allreduce_fn fn, void *user)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_set_allreduce: null");
		g->allreduce      = fn;
		g->allreduce_user = user;
		return TE_OK;
	});
}

static void *rcclSym(void *lib, const char *name) { return dlsym(lib, name); }
int te_rccl_unique_id(const char *libpath, char *id128)
{
	return guarded([&]() -> int {
		if (!libpath || !id128) return te::fail(TE_EINVAL, "te_rccl_unique_id: null argument");
		void *lib = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
		if (!lib) return te::fail(TE_EIO, std::string("te_rccl_unique_id: dlopen failed: ") + dlerror());
		auto get = (int (*)(void *)) rcclSym(lib, "ncclGetUniqueId");
		if (!get) return te::fail(TE_EIO, "te_rccl_unique_id: ncclGetUniqueId not found");
		int rc = get(id128);
		if (rc) return te::fail(TE_ESTATE, "ncclGetUniqueId failed");
		return TE_OK;
	});
}
int te_gmg_use_rccl(te_gmg *g, const char *libpath, const char *id128, int rank, int nranks)
{
	return guarded([&]() -> int {
		if (!g || !libpath || !id128) return te::fail(TE_EINVAL, "te_gmg_use_rccl: null argument");
		HIPCHK(hipSetDevice(g->device));
		void *lib = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
		if (!lib) return te::fail(TE_EIO, std::string("te_gmg_use_rccl: dlopen failed: ") + dlerror());
		static_assert(sizeof(ncclUniqueId) == 128, "te_rccl_unique_id hands out 128 bytes");
		ncclUniqueId id;
		memcpy(&id, id128, 128);
		auto init = (int (*)(void **, int, ncclUniqueId, int)) rcclSym(lib, "ncclCommInitRank");
		te_gmg::Rccl r;
		r.lib            = lib;
		r.GroupStart     = (int (*)()) rcclSym(lib, "ncclGroupStart");
		r.GroupEnd       = (int (*)()) rcclSym(lib, "ncclGroupEnd");
		r.Send           = (int (*)(const void *, size_t, int, int, void *, hipStream_t)) rcclSym(lib, "ncclSend");
		r.Recv           = (int (*)(void *, size_t, int, int, void *, hipStream_t)) rcclSym(lib, "ncclRecv");
		r.CommDestroy    = (int (*)(void *)) rcclSym(lib, "ncclCommDestroy");
		r.AllReduce      = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)) rcclSym(lib, "ncclAllReduce");
		r.GetErrorString = (const char *(*) (int) ) rcclSym(lib, "ncclGetErrorString");
		r.CommCount      = (int (*)(void *, int *)) rcclSym(lib, "ncclCommCount");
		r.CommUserRank   = (int (*)(void *, int *)) rcclSym(lib, "ncclCommUserRank");
		if (!init || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.CommDestroy || !r.GetErrorString || !r.AllReduce)
			return te::fail(TE_EIO, "te_gmg_use_rccl: RCCL symbols missing in " + std::string(libpath));
		if (nranks > 1 && (rank != g->rank || nranks != g->nranks)) // (before the communicator exists: nothing to leak)
			return te::fail(TE_EINVAL, "te_gmg_use_rccl: rank / nranks differ from the hierarchy's");
		int rc = init(&r.comm, nranks, id, rank);
		if (rc) return te::fail(TE_ESTATE, std::string("ncclCommInitRank failed: ") + r.GetErrorString(rc));
		if (g->rccl.comm && g->rccl.CommDestroy) (void) g->rccl.CommDestroy(g->rccl.comm); // a second call replaces the first communicator
		g->rccl = r;
		watchdogStart(g);
		return TE_OK;
	});
}
// The direct-store transport (pushkernels.hpp). Collective over the ranks of the hierarchy; needs a working reduction over the
// ranks (te_gmg_use_rccl or te_gmg_set_allreduce) to publish, once, every rank's IPC handles and receive offsets: a directory of
// 32-bit words, one slice per rank, summed over the ranks eight words at a time (each word has one contributor).
// Ranks that live in this very process (the tests' virtual ranks) are reached through their raw pointers.
static int pushSetup(te_gmg *g)
{
	auto &P = g->push;
	if (P.ready) return TE_OK;
	const int R = g->nranks, NL = (int) g->levels.size();
	if (g->dim != 3) return te::fail(TE_EUNSUPPORTED, "te_gmg_use_push: 3D hierarchies only");
	if (R < 2) return te::fail(TE_ESTATE, "te_gmg_use_push: one rank has nobody to push to");
	if (!g->rccl.comm && !g->allreduce) return te::fail(TE_ESTATE, "te_gmg_use_push: needs te_gmg_use_rccl or te_gmg_set_allreduce first (the handles travel through it)");
	if (R > PUSH_MAX_PEERS) return te::fail(TE_EUNSUPPORTED, "te_gmg_use_push: too many ranks");
	const bool self = g->cfg.has(O_RCCL_LOOPBACK); // diagnostic: every peer is this rank itself (tools/mr8_budget.py)
	P.nslot = 2 * NL;
	const size_t fbytes = sizeof(unsigned long long) * (size_t) R * P.nslot;
	HIPCHK(hipExtMallocWithFlags((void **) &P.flags, fbytes, hipDeviceMallocFinegrained));
	HIPCHK(hipMemset(P.flags, 0, fbytes));
	HIPCHK(hipMalloc((void **) &P.err, 64));
	HIPCHK(hipMemset(P.err, 0, 64));
	HIPCHK(hipHostMalloc((void **) &P.err_host, 64, hipHostMallocMapped));
	*P.err_host = 0;
	P.timeout_s = std::max(0.1, g->cfg.real(O_PUSH_TIMEOUT, P.timeout_s));
	int rc;
	for (int l = 0; l < NL; l++) {
		LevelHost &L = *g->levels[l];
		if ((rc = L.push_done.alloc(2))) return rc;
		HIPCHK(hipMemset(L.push_done.p, 0, 2 * sizeof(unsigned)));
		if (L.nremote > 0) {
			if ((rc = L.ghost_alt.alloc(L.ghost.n))) return rc;
			HIPCHK(hipMemset(L.ghost_alt.p, 0, sizeof(double) * L.ghost.n));
		}
		if (L.repl_up && L.repl_direct && l + 1 < NL) {
			te_vec *cf = g->levels[l + 1]->f.get();
			if ((rc = L.cf_alt.alloc(std::max<size_t>(cf->n, 2)))) return rc;
			HIPCHK(hipMemset(L.cf_alt.p, 0, sizeof(double) * std::max<size_t>(cf->n, 2)));
			L.cf_buf[0] = cf->d, L.cf_buf[1] = L.cf_alt.p;
		}
	}
	HIPCHK(hipDeviceSynchronize());
	// ---- the directory: per rank [pid, flags (handle 16 + pointer 2), per level: 4 x (handle 16 + pointer 2), R x recv offset 2]
	const int      HW = 18, LW = 4 * HW + 2 * R, W = 1 + HW + NL * LW;
	std::vector<double> dir((size_t) R * W, 0.0);
	auto put = [&](double *dst, const void *devptr) { // handle + raw pointer of one allocation (null: zeros)
		if (!devptr) return TE_OK;
		hipIpcMemHandle_t h;
		HIPCHK(hipIpcGetMemHandle(&h, const_cast<void *>(devptr)));
		uint32_t w[16];
		static_assert(sizeof h == 64, "hipIpcMemHandle_t is 64 bytes");
		memcpy(w, &h, 64);
		for (int k = 0; k < 16; k++) dst[k] = (double) w[k];
		const uint64_t a = (uint64_t) (uintptr_t) devptr;
		dst[16] = (double) (uint32_t) (a & 0xFFFFFFFFu), dst[17] = (double) (uint32_t) (a >> 32);
		return TE_OK;
	};
	double *mine = &dir[(size_t) g->rank * W];
	mine[0]      = (double) (uint32_t) getpid();
	if ((rc = put(mine + 1, P.flags))) return rc;
	for (int l = 0; l < NL; l++) {
		LevelHost &L = *g->levels[l];
		double    *q = mine + 1 + HW + (size_t) l * LW;
		if (L.nremote > 0 && ((rc = put(q, L.ghost.p)) || (rc = put(q + HW, L.ghost_alt.p)))) return rc;
		if (L.cf_buf[0] && ((rc = put(q + 2 * HW, L.cf_buf[0])) || (rc = put(q + 3 * HW, L.cf_buf[1])))) return rc;
		for (int r = 0; r < R; r++) q[4 * HW + 2 * r] = q[4 * HW + 2 * r + 1] = 0.0;
		for (size_t i = 0; i < L.fx.peers.size(); i++) { // where rank fx.peers[i]'s layers land in my ghost buffers (+1: 0 = nothing)
			const uint64_t o = (uint64_t) L.fx.recv_off[i] + 1;
			q[4 * HW + 2 * L.fx.peers[i]] = (double) (uint32_t) (o & 0xFFFFFFFFu), q[4 * HW + 2 * L.fx.peers[i] + 1] = (double) (uint32_t) (o >> 32);
		}
	}
	for (size_t i = 0; i < dir.size(); i += 8) {
		const int n = (int) std::min<size_t>(8, dir.size() - i);
		HIPCHK(hipMemcpyAsync(g->result.p, &dir[i], n * sizeof(double), hipMemcpyHostToDevice, g->stream));
		if ((rc = finishReduce(g, n, 0, true))) return rc;
		for (int k = 0; k < n; k++) dir[i + k] = g->result_host[k];
	}
	// ---- map the peers
	const uint32_t mypid = (uint32_t) getpid();
	auto open = [&](const double *src, void **out) -> int { // handle + pointer words of a peer's allocation -> a pointer usable here
		const uint64_t raw = (uint64_t) (uint32_t) src[16] | ((uint64_t) (uint32_t) src[17] << 32);
		*out               = nullptr;
		if (raw == 0) return TE_OK;
		hipIpcMemHandle_t h;
		uint32_t          w[16];
		for (int k = 0; k < 16; k++) w[k] = (uint32_t) src[k];
		memcpy(&h, w, 64);
		void *m = nullptr;
		HIPCHK(hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess));
		P.opened.push_back(m);
		*out = m;
		return TE_OK;
	};
	auto peerPtr = [&](int r, const double *src, void *own, void **out) -> int {
		const double *slice = &dir[(size_t) r * W];
		if (self || r == g->rank) { // (loop-back: this rank's own buffer stands in for the peer's)
			*out = own;
			return TE_OK;
		}
		if ((uint32_t) slice[0] == mypid) { // a virtual rank in this process: its pointer as it is
			*out = (void *) (uintptr_t) ((uint64_t) (uint32_t) src[16] | ((uint64_t) (uint32_t) src[17] << 32));
			return TE_OK;
		}
		return open(src, out);
	};
	// (a rank whose mapping fails still takes part in the closing reduction: all ranks succeed, or all fail)
	auto mapPeers = [&]() -> int {
	P.peer_flags.assign(R, nullptr);
	for (int r = 0; r < R; r++) {
		void *m = nullptr;
		if ((rc = peerPtr(r, &dir[(size_t) r * W + 1], P.flags, &m))) return rc;
		// loop-back: my own table stands in, shifted so that "my row of the peer's table" is the peer's row of mine
		P.peer_flags[r] = self ? P.flags + ((ptrdiff_t) r - g->rank) * P.nslot : (unsigned long long *) m;
		if (!P.peer_flags[r]) return te::fail(TE_ESTATE, "te_gmg_use_push: rank " + std::to_string(r) + " published no flag table");
	}
	for (int l = 0; l < NL; l++) {
		LevelHost &L = *g->levels[l];
		if (L.nremote > 0 && !L.fx.empty()) {
			for (int b = 0; b < 2; b++) L.push_peer_ghost[b].assign(L.fx.peers.size(), nullptr);
			bool ok = true;
			for (size_t i = 0; i < L.fx.peers.size() && ok; i++) {
				const int     r = L.fx.peers[i];
				const double *q = &dir[(size_t) r * W + 1 + HW + (size_t) l * LW];
				const uint64_t o1 = (uint64_t) (uint32_t) q[4 * HW + 2 * g->rank] | ((uint64_t) (uint32_t) q[4 * HW + 2 * g->rank + 1] << 32);
				const int64_t off = self ? L.fx.recv_off[i] : (int64_t) o1 - 1;
				if (off < 0) { // the peer expects nothing from me here although I send: the plans disagree
					ok = false;
					break;
				}
				for (int b = 0; b < 2; b++) {
					void *m = nullptr;
					// (loop-back: the OTHER buffer of this rank, so that what is being read is not overwritten)
					if ((rc = peerPtr(r, q + b * HW, b ? L.ghost.p : L.ghost_alt.p, &m))) return rc;
					if (!m) ok = false;
					L.push_peer_ghost[b][i] = m ? (double *) m + off : nullptr;
				}
			}
			if (!ok) return te::fail(TE_ESTATE, "te_gmg_use_push: a neighbour rank published no receive buffer for level " + std::to_string(l));
			// where every face of the send order goes (pack + push in one launch), and the flags that launch raises
			std::vector<unsigned long long *> fl;
			for (int b = 0; b < 2; b++) {
				std::vector<double *> dst((size_t) L.nremote, nullptr);
				for (size_t i = 0; i < L.fx.peers.size(); i++)
					for (int64_t k = 0; k < L.fx.send_cnt[i] / (int64_t) L.nf; k++)
						dst[(size_t) (L.fx.send_off[i] / (int64_t) L.nf + k)] = L.push_peer_ghost[b][i] + k * (int64_t) L.nf;
				int rc2 = L.push_face_dst[b].upload(dst);
				if (rc2) return rc2;
			}
			for (size_t i = 0; i < L.fx.peers.size(); i++)
				if (L.fx.send_cnt[i] > 0) fl.push_back(P.peer_flags[L.fx.peers[i]] + (size_t) g->rank * P.nslot + 2 * l);
			int rc3 = L.push_face_flags.upload(fl);
			if (rc3) return rc3;
			L.push_faces = true;
		}
		if (L.cf_buf[0] && !L.tx_direct.empty()) {
			for (int b = 0; b < 2; b++) L.push_peer_cf[b].assign(L.tx_direct.peers.size(), nullptr);
			for (size_t i = 0; i < L.tx_direct.peers.size(); i++) {
				const int     r = L.tx_direct.peers[i];
				const double *q = &dir[(size_t) r * W + 1 + HW + (size_t) l * LW];
				for (int b = 0; b < 2; b++) {
					void *m = nullptr;
					if ((rc = peerPtr(r, q + (2 + b) * HW, L.cf_buf[b ^ 1], &m))) return rc;
					if (!m) return te::fail(TE_ESTATE, "te_gmg_use_push: a rank published no coarse buffer for level " + std::to_string(l));
					L.push_peer_cf[b][i] = (double *) m;
				}
			}
			L.push_blocks = true;
		}
	}
	return TE_OK;
	};
	const int         map_rc  = mapPeers();
	const std::string map_msg = map_rc ? std::string(te_last_error()) : std::string();
	// nobody pushes before everybody has finished mapping (and zeroing): one more reduction, which also carries "somebody failed"
	double failed = map_rc ? 1.0 : 0.0;
	HIPCHK(hipMemcpyAsync(g->result.p, &failed, sizeof failed, hipMemcpyHostToDevice, g->stream));
	if ((rc = finishReduce(g, 1, 1, true))) return rc;
	if (map_rc) return te::fail(map_rc, map_msg);
	if (g->result_host[0] != 0.0) return te::fail(TE_ESTATE, "te_gmg_use_push: another rank could not map its peers' buffers");
	P.ready = true;
	return TE_OK;
}
int te_gmg_use_push(te_gmg *g, int enable)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_use_push: null");
		// Every switch between the transports is a point where ALL ranks have finished what they had queued: the argument that
		// lets a rank run one exchange ahead of a peer (LevelHost::ghost_alt) needs every exchange to be a direct one -- a rank that
		// starts pushing while a peer still reads its buffers in a cycle of the other transport would overwrite them. Collective.
		auto meet = [&]() -> int {
			HIPCHK(hipStreamSynchronize(g->stream));
			HIPCHK(hipStreamSynchronize(g->comm_stream));
			if (g->nranks < 2 || (!g->rccl.comm && !g->allreduce)) return TE_OK;
			double one = 1.0;
			HIPCHK(hipMemcpyAsync(g->result.p, &one, sizeof one, hipMemcpyHostToDevice, g->stream));
			return finishReduce(g, 1, 0, true);
		};
		int rc;
		if (!enable) {
			if (!g->push.on) return TE_OK;
			if ((rc = meet())) return rc;
			// back to the other transport: the coarse vectors return to their own storage
			for (size_t l = 0; l + 1 < g->levels.size(); l++)
				if (g->levels[l]->cf_buf[0]) g->levels[l + 1]->f->d = g->levels[l]->cf_buf[0];
			for (auto &L : g->levels) L->ghost_par = 0;
			g->push.on = false;
			return TE_OK;
		}
		if (g->push.on) return TE_OK;
		if ((rc = pushSetup(g)) || (rc = meet())) return rc;
		g->push.on = true;
		return TE_OK;
	});
}
// 0: no direct-store exchange has given up waiting; 1: one has (its data never arrived within TE_PUSH_TIMEOUT seconds: the results
// since then are garbage, and every later exchange of this solver returns at once). Reads the pinned host copy: no device call.
int te_gmg_push_failed(te_gmg *g)
{
	return guarded([&]() -> int { return (g && g->push.err_host && *g->push.err_host) ? 1 : 0; });
}

// moves n doubles from a scratch send buffer to a scratch receive buffer of level 0 through the same
// code path as a real exchange, with this rank as its own peer; returns TE_OK iff the data arrived intact
int te_gmg_exchange_selftest(te_gmg *g, int n)
{
	return guarded([&]() -> int {
		if (!g || n < 1) return te::fail(TE_EINVAL, "te_gmg_exchange_selftest: bad argument");
		LevelHost &L = *g->levels[0];
		if ((size_t) 2 * n > L.r->n) return te::fail(TE_EINVAL, "te_gmg_exchange_selftest: n too large");
		std::vector<double> h(n), back(n);
		for (int i = 0; i < n; i++) h[i] = 1.0 + i * 0.5;
		double *send = L.r->d, *recv = L.r->d + n;
		HIPCHK(hipMemcpyAsync(send, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, g->stream));
		HIPCHK(hipMemsetAsync(recv, 0, sizeof(double) * n, g->stream));
		ExPlan pl;
		int    me = 0;
		pl.peers  = {me};
		pl.send_off = {0}, pl.send_cnt = {n}, pl.recv_off = {0}, pl.recv_cnt = {n};
		int rc = doExchange(g, 9, pl, send, recv);
		if (rc) return rc;
		HIPCHK(hipMemcpyAsync(back.data(), recv, sizeof(double) * n, hipMemcpyDeviceToHost, g->stream));
		HIPCHK(hipStreamSynchronize(g->stream));
		for (int i = 0; i < n; i++)
			if (back[i] != h[i]) return te::fail(TE_ESTATE, "te_gmg_exchange_selftest: data mismatch");
		if (g->rccl.comm) { // the scalar reduction of te_bicgstab / te_gmg_verify_schedule: ncclAllReduce on the solver stream
			const double v[4] = {1.5, -2.25, 3.0, 0.125};
			HIPCHK(hipMemcpyAsync(g->result.p, v, sizeof v, hipMemcpyHostToDevice, g->stream));
			int r2 = g->rccl.AllReduce(g->result.p, g->result.p, 4, ncclFloat64, ncclSum, g->rccl.comm, g->stream);
			if (r2) return te::fail(TE_ESTATE, std::string("ncclAllReduce failed: ") + g->rccl.GetErrorString(r2));
			HIPCHK(hipMemcpyAsync(g->result_host, g->result.p, sizeof v, hipMemcpyDeviceToHost, g->stream));
			HIPCHK(hipStreamSynchronize(g->stream));
			for (int i = 0; i < 4; i++)
				if (g->result_host[i] != v[i] * g->nranks) return te::fail(TE_ESTATE, "te_gmg_exchange_selftest: all-reduce mismatch");
		}
		return TE_OK;
	});
}

int te_vec_create(te_gmg *g, int level, te_vec **out)
{
	return guarded([&]() -> int {
		if (!g || !out || level < 0 || level >= (int) g->levels.size())
			return te::fail(TE_EINVAL, "te_vec_create: bad argument");
		HIPCHK(hipSetDevice(g->device));
		return newVec(g, level, out);
	});
}
void te_vec_destroy(te_vec *v)
{
	if (!v) return;
	(void) hipStreamSynchronize(v->g->stream);
	(void) hipFree(v->d);
	delete v;
}
size_t te_vec_size(const te_vec *v) { return v ? v->n : 0; }
int    te_vec_upload(te_vec *v, const double *host)
{
	return guarded([&]() -> int {
		if (!v || !host) return te::fail(TE_EINVAL, "te_vec_upload: null");
		HIPCHK(hipMemcpyAsync(v->d, host, sizeof(double) * v->n, hipMemcpyHostToDevice, v->g->stream));
		HIPCHK(hipStreamSynchronize(v->g->stream));
		return TE_OK;
	});
}
int te_vec_download(const te_vec *v, double *host)
{
	return guarded([&]() -> int {
		if (!v || !host) return te::fail(TE_EINVAL, "te_vec_download: null");
		HIPCHK(hipMemcpyAsync(host, v->d, sizeof(double) * v->n, hipMemcpyDeviceToHost, v->g->stream));
		HIPCHK(hipStreamSynchronize(v->g->stream));
		return TE_OK;
	});
}
void *te_vec_device_ptr(te_vec *v) { return v ? v->d : nullptr; }

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

static int checkLevelVec(te_gmg *g, int level, const te_vec *v, const char *who)
{
	if (!g || !v || level < 0 || level >= (int) g->levels.size() || v->g != g || v->level != level)
		return te::fail(TE_EINVAL, std::string(who) + ": vector does not belong to this level");
	return TE_OK;
}
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
		if (L.dim == 2) { // (2D: the residual kernel, then the reduction pass)
			if ((rc = launchStencil<MODE_RESID>(g, L, u->d, f->d, r->d, 0.0))) return rc;
			return reduce<RED_SUMSQ>(r, nullptr, norm_sq);
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
int te_gmg_verify_schedule(te_gmg *g, const te_cycle_opts *o)
{
	return guarded([&]() -> int {
		if (!g || !o) return te::fail(TE_EINVAL, "te_gmg_verify_schedule: null argument");
		int rc = verifySchedule(g, o);
		if (rc == TE_OK) g->verified_opts.insert(optsKey(o));
		return rc;
	});
}
static int vcycleWith(te_gmg *g, const te_cycle_opts *o, const te_vec *f, te_vec *u, const PendingRhs *pending);
int te_vcycle(te_gmg *g, const te_cycle_opts *o, const te_vec *f, te_vec *u)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_vcycle: null solver");
		WatchdogBatch batch(g);
		return vcycleWith(g, o, f, u, nullptr);
	});
}
// te_vcycle; `pending`: f is still to be formed (te_bicgstab; consumed by level 0's first reader, visit())
static int vcycleWith(te_gmg *g, const te_cycle_opts *o, const te_vec *f, te_vec *u, const PendingRhs *pending)
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
                g->profiling = prof;
                return r2;
			};
			g->push.fatal.store(false);
			if ((rc = te_gmg_use_push(g, 0)) || (rc = timeIt(&t_other)) || (rc = vcycleWith(g, o, f, ref, nullptr))) return fail(rc);
			if ((rc = te_vec_copy(f2, f)) || (rc = te_vec_scale(f2, -0.625))) return fail(rc);
			if ((rc = te_gmg_use_push(g, 1))) return fail(rc);
			for (int trial = 0; trial < 3 && bad == 0.0; trial++) { // (three times: a race does not show every time)
				if ((rc = vcycleWith(g, o, f2, u, nullptr)) || (rc = vcycleWith(g, o, f, u, nullptr))) return fail(rc);
				if ((rc = te_vec_add_scaled(u, -1.0, ref))) return fail(rc);
				double dmax = 0;
				if (u->n > 0 && (rc = reduce<RED_MAXABS>(u, nullptr, &dmax))) return fail(rc);
				HIPCHK(hipStreamSynchronize(g->stream));
				if (dmax != 0.0 || te_gmg_push_failed(g)) bad = 1.0;
			}
			HIPCHK(hipMemcpyAsync(g->result.p, &bad, sizeof bad, hipMemcpyHostToDevice, g->stream));
			if ((rc = finishReduce(g, 1, 1, true))) return fail(rc);
			bad = g->result_host[0];
			if (bad == 0.0 && (rc = timeIt(&t_push))) return fail(rc);
			char buf[160];
			if (bad != 0.0) {
				(void) te_gmg_use_push(g, 0);
				g->push.ready = false; // not usable on this machine: never again for this solver
				snprintf(buf, sizeof buf, "transport: direct-store REJECTED (result differs from the other transport's or a wait gave up) -> %s; ",
				         g->rccl.comm ? "rccl" : "callback");
			} else {
				const bool take = t_push < 0.98 * t_other;
				if (!take) (void) te_gmg_use_push(g, 0);
				snprintf(buf, sizeof buf, "transport: %s=%.1fus direct-store=%.1fus (results identical) -> %s; ", g->rccl.comm ? "rccl" : "callback",
				         t_other * 1e3, t_push * 1e3, take ? "direct-store" : (g->rccl.comm ? "rccl" : "callback"));
			}
			tnote = buf;
			g->push.fatal.store(true);
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
// The communicator the native back-end really runs on: ncclCommCount / ncclCommUserRank of te_gmg_use_rccl's communicator
// (0 / -1 without one) -- so that a bench line can show that RCCL saw N ranks rather than say so.
int te_gmg_comm_info(te_gmg *g, int *rccl_nranks, int *rccl_rank)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_comm_info: null");
		int n = 0, r = -1;
		if (g->rccl.comm && g->rccl.CommCount && g->rccl.CommUserRank) {
			int rc = g->rccl.CommCount(g->rccl.comm, &n);
			if (rc == 0) rc = g->rccl.CommUserRank(g->rccl.comm, &r);
			if (rc) return te::fail(TE_ESTATE, std::string("ncclCommCount failed: ") + g->rccl.GetErrorString(rc));
		}
		if (rccl_nranks) *rccl_nranks = n;
		if (rccl_rank) *rccl_rank = r;
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

// Init::initDirichlet / initNeumann for the drivers' canned problems, on the device (initkernels.hpp)
int te_init_problem(te_gmg *g, int level, int problem, int neumann, te_vec *f, te_vec *exact)
{
	return guarded([&]() -> int {
		int rc;
		if ((rc = checkLevelVec(g, level, f, "te_init_problem"))) return rc;
		if (exact && (rc = checkLevelVec(g, level, exact, "te_init_problem"))) return rc;
		if (exact == f) return te::fail(TE_EINVAL, "te_init_problem: f and exact must be different vectors");
		LevelHost &L = *g->levels[level];
		if (L.xf_valid_for == f->d || (exact && L.xf_valid_for == exact->d)) L.xf_valid_for = nullptr;
		if (L.P == 0) return TE_OK;
		InitGeom G;
		G.dim = L.dim, G.n = L.n, G.P = L.P;
		G.starts = L.geom_starts.p, G.h = L.geom_h.p, G.face_kind = L.face_kind.p, G.ids = L.node_ids.p;
		const dim3 grid(gridFor(f->n, 256, 1 << 20)), blk(256);
		double    *e = exact ? exact->d : nullptr;
		Timed      t(g, KC_VECOP, f->n);
#define TE_INIT(K, PROB)                                                                              \
		if (neumann)                                                                                      \
			hipLaunchKernelGGL((K<PROB, true>), grid, blk, 0, g->stream, G, f->d, e);                     \
		else                                                                                              \
			hipLaunchKernelGGL((K<PROB, false>), grid, blk, 0, g->stream, G, f->d, e);
		if (problem == PROBLEM_RANDOM) {
			hipLaunchKernelGGL(k_init_random, grid, blk, 0, g->stream, G, L.nc, (uint64_t) 0x5EED, f->d, e);
		} else if (problem == PROBLEM_TRIG) {
			if (L.dim == 3) {
				TE_INIT(k_init3d, PROBLEM_TRIG)
			} else {
				TE_INIT(k_init2d, PROBLEM_TRIG)
			}
		} else if (problem == PROBLEM_GAUSS) {
			if (L.dim == 3) {
				TE_INIT(k_init3d, PROBLEM_GAUSS)
			} else {
				TE_INIT(k_init2d, PROBLEM_GAUSS)
			}
		} else {
			return te::fail(TE_EINVAL, "te_init_problem: unknown problem");
		}
#undef TE_INIT
		HIPCHK(hipGetLastError());
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
// Vector<D>::getLocalData(i) for a run of patches (PetscVector.h:87-98): what Init::initDirichlet, the writers and
// the C++ adaptor's host mirror move -- never the whole vector for one patch.
int te_vec_upload_patches(te_vec *v, int first_patch, int npatches, const double *host)
{
	return guarded([&]() -> int {
		if (!v || !host) return te::fail(TE_EINVAL, "te_vec_upload_patches: null");
		const size_t nc = v->g->levels[v->level]->nc;
		if (first_patch < 0 || npatches < 0 || ((size_t) first_patch + npatches) * nc > v->n)
			return te::fail(TE_EINVAL, "te_vec_upload_patches: patch range outside the vector");
		if (npatches == 0) return TE_OK;
		LevelHost &L = *v->g->levels[v->level];
		if (L.xf_valid_for == v->d) L.xf_valid_for = nullptr;
		HIPCHK(hipMemcpyAsync(v->d + (size_t) first_patch * nc, host, sizeof(double) * nc * npatches, hipMemcpyHostToDevice, v->g->stream));
		HIPCHK(hipStreamSynchronize(v->g->stream));
		return TE_OK;
	});
}
int te_vec_download_patches(const te_vec *v, int first_patch, int npatches, double *host)
{
	return guarded([&]() -> int {
		if (!v || !host) return te::fail(TE_EINVAL, "te_vec_download_patches: null");
		const size_t nc = v->g->levels[v->level]->nc;
		if (first_patch < 0 || npatches < 0 || ((size_t) first_patch + npatches) * nc > v->n)
			return te::fail(TE_EINVAL, "te_vec_download_patches: patch range outside the vector");
		if (npatches == 0) return TE_OK;
		HIPCHK(hipMemcpyAsync(host, v->d + (size_t) first_patch * nc, sizeof(double) * nc * npatches, hipMemcpyDeviceToHost, v->g->stream));
		HIPCHK(hipStreamSynchronize(v->g->stream));
		return TE_OK;
	});
}

int te_gmg_profile(te_gmg *g, int enable)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_profile: null");
		drainEvents(g);
		g->profiling = enable != 0;
		return TE_OK;
	});
}
int te_integrate(te_gmg *g, int level, const te_vec *v, double *out)
{
	return guarded([&]() -> int {
		int rc;
		if (!out) return te::fail(TE_EINVAL, "te_integrate: null result");
		if ((rc = checkLevelVec(g, level, v, "te_integrate"))) return rc;
		LevelHost &L = *g->levels[level];
		*out         = 0.0;
		if (L.P == 0 || (L.replicated && g->rank != 0)) return TE_OK; // (a level on every rank counts once: rank 0's)
		DevBuf<double> part;
		if ((rc = part.alloc(L.P))) return rc;
		hipLaunchKernelGGL(k_patch_integrals, dim3(L.P), dim3(256), 0, g->stream, (int) L.nc, v->d, L.cellvol.p, part.p);
		HIPCHK(hipGetLastError());
		std::vector<double> h(L.P);
		HIPCHK(hipMemcpyAsync(h.data(), part.p, sizeof(double) * L.P, hipMemcpyDeviceToHost, g->stream));
		HIPCHK(hipStreamSynchronize(g->stream));
		double sum = 0.0;
		for (double x : h) sum += x; // patch order, as the reference's loop over its patch map
		*out = sum;
		return TE_OK;
	});
}
int te_volume(te_gmg *g, int level, double *out)
{
	return guarded([&]() -> int {
		if (!g || !out || level < 0 || level >= (int) g->levels.size()) return te::fail(TE_EINVAL, "te_volume: bad argument");
		double sum = 0.0;
		if (!(g->levels[level]->replicated && g->rank != 0)) // (a level on every rank counts once: rank 0's)
			for (double x : g->levels[level]->patch_vol) sum += x;
		*out = sum;
		return TE_OK;
	});
}
int te_gmg_profile_select(te_gmg *g, const char *name)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_profile_select: null");
		drainEvents(g);
		g->prof_only = -1;
		if (!name || !*name) return TE_OK;
		for (int k = 0; k < KC_COUNT; k++)
			if (!strcmp(name, kclassName[k])) {
				g->prof_only = k;
				return TE_OK;
			}
		return te::fail(TE_EINVAL, std::string("te_gmg_profile_select: unknown kernel class ") + name);
	});
}
int te_gmg_profile_reset(te_gmg *g)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_profile_reset: null");
		drainEvents(g);
		memset(g->calls, 0, sizeof(g->calls));
		memset(g->cells, 0, sizeof(g->cells));
		memset(g->total_ms, 0, sizeof(g->total_ms));
		return TE_OK;
	});
}
int te_gmg_profile_rows(te_gmg *g, int max_rows, char (*name)[64], int64_t *calls, double *total_ms,
                        int64_t *cells)
{
	return guarded([&]() -> int {
		if (!g || !name || !calls || !total_ms || !cells) return te::fail(TE_EINVAL, "te_gmg_profile_rows: null");
		drainEvents(g);
		int n = 0;
		for (int k = 0; k < KC_COUNT && n < max_rows; k++) {
			if (g->calls[k] == 0) continue;
			strncpy(name[n], kclassName[k], 63);
			name[n][63] = 0;
			calls[n]    = g->calls[k];
			total_ms[n] = g->total_ms[k];
			cells[n]    = g->cells[k];
			n++;
		}
		return n;
	});
}
// Diagnostic for the watchdog's bookkeeping: for `seconds` of wall time the host enqueues, WITHOUT ever synchronising, a
// level-0 vector kernel followed by an armed "exchange" (this rank as its own peer through the active back-end, or a
// device-to-device copy when none is set). The host runs ahead of the GPU, so the newest exchange is never complete when
// the watchdog polls; every exchange does complete within milliseconds, so a correct watchdog (deadline of the OLDEST
// outstanding exchange) stays quiet even when `seconds` exceeds TE_EXCHANGE_TIMEOUT. Returns the number of exchanges issued.
int te_gmg_watchdog_selftest(te_gmg *g, double seconds)
{
	return guarded([&]() -> int {
			if (!g || seconds <= 0) return te::fail(TE_EINVAL, "te_gmg_watchdog_selftest: bad argument");
			watchdogStart(g);
			LevelHost &L = *g->levels[0];
			const int  n = 256;
			if ((size_t) 2 * n > L.r->n) return te::fail(TE_EINVAL, "te_gmg_watchdog_selftest: level 0 too small");
			double *send = L.r->d, *recv = L.r->d + n;
			ExPlan  pl;
			pl.peers    = {g->rank};
			pl.send_off = {0}, pl.send_cnt = {n}, pl.recv_off = {0}, pl.recv_cnt = {n};
			const auto t0 = std::chrono::steady_clock::now();
			int        count = 0;
			while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
				int rc = vecop<VOP_SCALE>(L.t.get(), nullptr, nullptr, 1.0, 0, 0);
				if (rc) return rc;
				if (g->rccl.comm || g->exchange) {
					if ((rc = doExchange(g, 9, pl, send, recv))) return rc;
				} else {
					WatchdogArm arm(g, g->stream, 9);
					HIPCHK(hipMemcpyAsync(recv, send, sizeof(double) * n, hipMemcpyDeviceToDevice, g->stream));
				}
				count++;
			}
			HIPCHK(hipStreamSynchronize(g->stream));
			return count;
	});
}
// te_bicgstab keeps its eight level-0 work vectors between solves (8 GiB at 512^3); a caller that is done solving hands
// them back with this call (they are allocated again by the next te_bicgstab)
int te_gmg_release_workspace(te_gmg *g)
{
	return guarded([&]() -> int {
			if (!g) return te::fail(TE_EINVAL, "te_gmg_release_workspace: null");
			for (te_vec *&v : g->bicg_work) {
				if (v) te_vec_destroy(v);
				v = nullptr;
			}
			return TE_OK;
	});
}
// One TE_* switch of this solver: value == NULL clears it (back to the default). te_gmg_create reads all of
// them from the environment once; afterwards this is the only way to change one. Switches that shape the level tables
// (TE_2D_SIMPLE, TE_NO_CFP, TE_2D_NO_MR_FUSE, TE_NO_OVERLAP, TE_EXCHANGE_TIMEOUT) are fixed at creation: TE_ESTATE.
int te_gmg_set_option(te_gmg *g, const char *name, const char *value)
{
	return guarded([&]() -> int {
			if (!g || !name) return te::fail(TE_EINVAL, "te_gmg_set_option: null argument");
			for (int o = 0; o < O_COUNT; o++)
				if (!strcmp(name, optName[o])) {
					if (optStructural(o))
						return te::fail(TE_ESTATE, std::string("te_gmg_set_option: ") + name + " is read when the solver is created; set it in the environment before te_gmg_create");
					g->cfg.set(o, value);
					g->verified_opts.clear(); // (an option may change which exchanges a cycle issues)
					return TE_OK;
				}
			return te::fail(TE_EINVAL, std::string("te_gmg_set_option: unknown option ") + name);
	});
}
} // extern "C"
