// Hand-written HIP kernels (gfx950 / CDNA4, wave64) for the 3D GMG hot path.
//
// All kernels work on the reference's vector layout (patch-major, x-fastest, interior cells
// only: v[p*N^3 + x + N*y + N*N*z], src/Thunderegg/PetscVector.h:70-98). Coupling between
// patches uses ghost values instead of the reference's interface vector gamma; on a
// same-level face ghost = neighbour cell, which is algebraically the reference's
// (2*gamma - 3*m + u)/h^2 closure with gamma = (m + nbr)/2 (StarPatchOp.h:46-48,
// TriLinInterp.cpp:78-84). Physical faces: Dirichlet ghost = -m, Neumann ghost = +m
// (StarPatchOp.h:49-65). Coarse/fine faces read a ghost plane that k_cf_ghost3d fills
// with 2*gamma - m from TriLinInterp.cpp:85-170's weights.
//
// Bandwidth-bound stencils: no MFMA. One workgroup marches one patch (or a z-slab of it)
// plane by plane (march3d.hpp); z-neighbours live in registers, x/y-neighbours in a double-buffered
// LDS plane with a one-cell halo; HBM sees each interior cell once per operand.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pushkernels.hpp" // PushFlag / pushRaise: the pack kernels raise the peers' flags themselves (PackPush)

namespace te
{
enum FaceKind : int32_t { FACE_DIRICHLET = 0, FACE_NEUMANN = 1, FACE_LOCAL = 2, FACE_GHOST = 3 };
enum StencilMode : int { MODE_APPLY = 0, MODE_RESID = 1, MODE_JACOBI = 2, MODE_RESID_RESTRICT = 3 };

// where the fused residual+restriction kernel puts the coarse right-hand side of a patch:
// parent[p] >= 0: octant orth[p] of that coarse patch; parent[p] <= -2: block -(parent+2) of `remote`
// (the parent lives on another rank); orth[p] < 0: the patch does not coarsen, r is copied through.
struct RestrictDst {
	const int32_t *parent, *orth;
	double        *coarse, *remote;
	const int64_t *remote_off;
	double        *rs6;   // [P][6][(N/2)^2]: 2x2 sums of the patches' face layers (march3d.hpp, k_rbgs_zero_resid3d EXPORT), or null
};

// Diagnostic build only (-DTE_STAMPS=1, tools/tail_stamps.py; never in libte_hip.so): wave 0 of every workgroup of the small-level
// kernels keeps wall-clock stamps (s_memrealtime, 100 MHz) of the links of its dependent chain -- entry, tables there, first data
// there, march begun, last result formed, stores retired -- in scalar registers and writes them out once, at its very end.
#ifndef TE_STAMPS
#define TE_STAMPS 0
#endif
#if TE_STAMPS
constexpr int TE_NSTAMP = 8;
struct StampDst {
	unsigned long long *p = nullptr; // [workgroup][TE_NSTAMP], or null
};
struct Stamps {
	unsigned long long t[TE_NSTAMP] = {};
	// DRAIN: everything this wave has requested is there (the link's data has arrived) before the clock is read
	template <int K, bool DRAIN> __device__ __forceinline__ void at()
	{
		if (DRAIN) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
		t[K] = __builtin_amdgcn_s_memrealtime();
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
	}
	__device__ __forceinline__ void flush(const StampDst &d, int wg) const
	{
		if (d.p && threadIdx.x == 0)
#pragma unroll
			for (int k = 0; k < TE_NSTAMP; k++) d.p[(size_t) wg * TE_NSTAMP + k] = t[k];
	}
};
// (a value the stamp that follows must find in its register: without it the compiler sinks loads of read-only memory below the stamp)
#define TE_STAMP_PIN(v) asm volatile("" ::"v"(v))
#define TE_STAMP_DECL Stamps stamps_
#define TE_STAMP(K, DRAIN) stamps_.template at<K, DRAIN>()
#define TE_STAMP_FLUSH(dst, wg) stamps_.flush(dst, wg)
#define TE_STAMP_PARAM , StampDst stamp_dst
#else
#define TE_STAMP_PIN(v)
#define TE_STAMP_DECL
#define TE_STAMP(K, DRAIN)
#define TE_STAMP_FLUSH(dst, wg)
#define TE_STAMP_PARAM
#endif

struct LevelDev {
	int32_t        P;         // local patches
	const int32_t *face_kind; // [P*6]
	const int32_t *face_src;  // [P*6] FACE_LOCAL: neighbour patch; FACE_GHOST: ghost plane slot
	const double  *face_kadj; // [P*6] change of the diagonal's per-axis factor 2 at that face
	const double  *rh2;       // [P*3] 1/h^2
	const double  *ghost;     // [nslots*N*N]
	// a launch covers patches order[first .. first+count) (order == nullptr: patches first .. first+count):
	// interior patches can run while the ghost exchange of the boundary patches is still in flight
	const int32_t *order;
	int32_t        first, count;
	// compact copies of the patches' two x-face columns, [p][W|E][y + N z]: an x-halo gathered from the
	// neighbour patch itself uses 8 B of every 128-B line; from here it is a contiguous 256-B run per plane.
	// xf (may be null) belongs to the input iterate u; xf_out (may be null) is filled for the output.
	const double *xf;
	double       *xf_out;
	// all six face layers of an iterate that is never stored as a whole (opts.fuse = 3): [P][6][N*N], a face cell at
	// a + N b with (a, b) the two other axes in order (the ghost-slot layout). f6 belongs to the input, f6_out is filled.
	const double *f6;
	double       *f6_out;
	// where face layer (p, s) sits inside f6 / f6_out, in units of N*N doubles (null: at p * 6 + s). A level cut by rank
	// boundaries keeps the layers that travel to other ranks first, in send order: the exchange sends them from where
	// they are and no pack kernel runs (gmg_core.hip buildLevel)
	const int32_t *f6off;
	// ghost terms that still belong to this level's right-hand side (march3d.hpp FCorrSrc), or null
	const double *fcorr;
#if TE_STAMPS
	StampDst stamp_dst;
#endif
};

// Kernel arguments are fetched with scalar loads where the code first needs them: behind the early exit, behind the `order`
// branch, behind the position tests -- four or five dependent round trips of 0.3-0.4 us each before the first table load of a
// small-level kernel can be issued (profiles/r06_tail_stamps.txt: 1.7 us from a workgroup's first instruction to its tables).
// Naming them as inputs of an empty asm statement at the top of the kernel makes the compiler fetch them there, together.
__device__ __forceinline__ void argsUpFront(const LevelDev &L, const void *a = nullptr, const void *b = nullptr, const void *c = nullptr,
                                            const void *d = nullptr, const void *e = nullptr, const void *f = nullptr)
{
	asm volatile("" ::"s"(L.P), "s"(L.face_kind), "s"(L.face_src), "s"(L.face_kadj), "s"(L.rh2), "s"(L.ghost), "s"(L.order), "s"(L.first),
	             "s"(L.count), "s"(L.xf), "s"(L.xf_out), "s"(L.f6), "s"(L.f6_out), "s"(L.f6off), "s"(L.fcorr), "s"(a), "s"(b), "s"(c), "s"(d),
	             "s"(e), "s"(f));
}

template <int N> __device__ __forceinline__ size_t f6Face(const int32_t *f6off, int p, int s)
{
	return (size_t) (f6off ? f6off[(size_t) p * 6 + s] : p * 6 + s) * (N * N);
}

// Blocks b, b+8, b+16, ... share an XCD (observed round-robin dispatch); give every XCD one
// contiguous run of patches so that x/y/z-neighbour faces are mostly served from that XCD's
// L2. Speed only; correctness never depends on placement.
#ifndef TE_XCD_REMAP
#define TE_XCD_REMAP 1
#endif
__device__ __forceinline__ int xcdRemap(int b, int nblocks)
{
#if TE_XCD_REMAP
	int chunk = (nblocks + 7) >> 3;
	return (b & 7) * chunk + (b >> 3);
#else
	return b;
#endif
}

// Workgroup barrier that waits for this wave's LDS traffic only. __syncthreads() also drains
// vmcnt(0), i.e. every global prefetch issued just before it and every store of the previous
// plane; keeping those in flight across the barrier is what lets the plane pipeline stream.
__device__ __forceinline__ void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS plane of the marching kernels (march3d.hpp): [pad, west halo, N cells, east halo, pad] per row so that
// the interior is 16-B aligned for ds_read/write_b128, plus one halo row below and above.
// Row pairs (1,2), (3,4), ... (r = y + 1) alternate a skew of ONE double. A 32-lane group of the thread map (16 x-pairs x
// 2 row pairs, march3d.hpp) reads single doubles with ds_read_b64 (banks (a/4) mod 64 per 32 lanes): without the skew the
// rows of lanes (X, Yp) and (X, Yp+1) lie 2 LW doubles = 144 dwords = 16 (mod 64) apart, so lane (X, Yp+1) meets lane
// (X+4, Yp) on its two banks -- every such read was two-way conflicted (round 2 counters: 40-46 % of the LDS cycles of
// the sweep kernels). With it the two row pairs lie an odd number of doubles apart and occupy disjoint banks. The price:
// an interior row start is only 8-B aligned, so pairs are moved with ds_read2/write2_b64 (ldsLoad2 / ldsStore2), not b128.
#ifndef TE_LDS_SKEW
#define TE_LDS_SKEW 1 // (0: rows at plain multiples of LW -- tooling builds a second library with it for same-box comparisons)
#endif
template <int N> struct Tile2 {
	static constexpr int LW  = N + 4;
	static constexpr int LSZ = LW * (N + 2) + 2;
	__host__ __device__ static constexpr int row(int r) { return r * LW + (TE_LDS_SKEW ? (((r + 1) >> 1) & 1) : 0); }
};
struct __attribute__((packed, aligned(8))) double2a8 {
	double x, y;
};
__device__ __forceinline__ double2 ldsLoad2(const double *p)
{
	const double2a8 v = *reinterpret_cast<const double2a8 *>(p);
	return double2{v.x, v.y};
}
__device__ __forceinline__ void ldsStore2(double *p, double2 v) { *reinterpret_cast<double2a8 *>(p) = double2a8{v.x, v.y}; }

// A plane just outside the patch in z, as "sign * memory": FACE_LOCAL -> the neighbour's facing
// plane, FACE_GHOST -> the ghost slot, physical -> +-(own boundary plane) (or 0 when the caller
// folds the physical closure into the diagonal).
struct PlaneSrc {
	const double2 *p;
	double         s;
};
template <int N>
__device__ __forceinline__ PlaneSrc zPlaneSrc(int kind, int src, bool top, const double *u, const double *up,
                                              const double *ghost, double dir_sign, double neu_sign)
{
	constexpr int NN = N * N, NNN = N * N * N;
	PlaneSrc      r;
	const double *own = up + (top ? (N - 1) * NN : 0);
	const double *p   = own;
	r.s               = 1.0;
	if (kind == FACE_LOCAL) p = u + (size_t) src * NNN + (top ? 0 : (N - 1) * NN);
	if (kind == FACE_GHOST) p = ghost + (size_t) src * NN;
	if (kind == FACE_DIRICHLET) r.s = dir_sign;
	if (kind == FACE_NEUMANN) r.s = neu_sign;
	r.p = reinterpret_cast<const double2 *>(p);
	return r;
}
// One x/y halo entry per thread (threads >= 4N get a harmless dummy): value(z) = s * p[z*stride].
struct HaloSrc {
	const double *p;
	int           stride;
	double        s;
	int           lds; // slot in the LDS tile, -1 = none
};
// The six entries of one patch's row of a face table, read together and kept in registers: an index that differs from lane to lane
// (the halo threads' `side`) picks among them with selects. Taken from memory entry by entry where the code first asks -- under the
// face-kind tests, in one helper after the other -- the small-level kernels spent five to ten dependent round trips on them before
// their first plane was requested (profiles/r06_tail_stamps.txt).
struct Reg6 {
	int32_t v[6];
	__device__ __forceinline__ explicit Reg6(const int32_t *p)
	{
#pragma unroll
		for (int s = 0; s < 6; s++) v[s] = p[s];
	}
	__device__ __forceinline__ int32_t operator[](int i) const
	{
		// (each candidate passes through an empty asm first: over plain array elements the compiler turns the chain of selects back into
		// ONE load at a selected address -- from a copy of the array in scratch memory, 52 bytes per lane and a dependent round trip of
		// its own; tests/test_isa_waits.py checks that no slab kernel touches scratch)
		int32_t c[6];
#pragma unroll
		for (int s = 0; s < 6; s++) {
			c[s] = v[s];
			asm volatile("" : "+v"(c[s]));
		}
		int32_t r = c[0];
#pragma unroll
		for (int s = 1; s < 6; s++) r = (i == s) ? c[s] : r;
		return r;
	}
};
template <int N, class FK>
__device__ __forceinline__ HaloSrc haloSrc(int tid, const FK &fk, const FK &fs, const double *u,
                                           const double *up, const double *ghost, double dir_sign, double neu_sign,
                                           const double *xf = nullptr)
{
	constexpr int NN = N * N, NNN = N * N * N;
	using T2 = Tile2<N>;
	HaloSrc       h;
	h.p      = up;
	h.stride = 0;
	h.s      = 0.0;
	h.lds    = -1;
	if (tid < 4 * N) {
		const int side = tid / N, t = tid % N;
		const int kind = fk[side], src = fs[side];
		int       own, nbr;
		if (side == 0) { // west: own (0,t), neighbour's (N-1,t)
			own   = t * N;
			nbr   = t * N + (N - 1);
			h.lds = T2::row(t + 1) + 1;
		} else if (side == 1) {
			own   = t * N + (N - 1);
			nbr   = t * N;
			h.lds = T2::row(t + 1) + N + 2;
		} else if (side == 2) { // south: own (t,0), neighbour's (t,N-1)
			own   = t;
			nbr   = (N - 1) * N + t;
			h.lds = T2::row(0) + t + 2;
		} else {
			own   = (N - 1) * N + t;
			nbr   = t;
			h.lds = T2::row(N + 1) + t + 2;
		}
		h.stride = NN;
		h.s      = 1.0;
		h.p      = up + own;
		if (kind == FACE_LOCAL) h.p = u + (size_t) src * NNN + nbr;
		if (kind == FACE_LOCAL && side < 2 && xf) { // the neighbour's opposite x-face column, contiguous in (y, z)
			h.p      = xf + ((size_t) src * 2 + (side ^ 1)) * NN + t;
			h.stride = N;
		}
		if (kind == FACE_GHOST) { // ghost plane cell (a,b) = (t, z)
			h.p      = ghost + (size_t) src * NN + t;
			h.stride = N;
		}
		if (kind == FACE_DIRICHLET) h.s = dir_sign;
		if (kind == FACE_NEUMANN) h.s = neu_sign;
	}
	return h;
}

// The same two helpers for an iterate that exists only as its six face layers (LevelDev.f6) -- the relaxation kernels
// fold physical faces into the diagonal, so those contribute 0.
template <int N>
__device__ __forceinline__ PlaneSrc zPlaneSrc6(int kind, int src, bool top, const double *f6, const double *ghost, const int32_t *f6off)
{
	constexpr int NN = N * N;
	PlaneSrc      r;
	const double *p = f6; // harmless valid address where nothing applies (scale 0)
	r.s             = 0.0;
	if (kind == FACE_LOCAL) p = f6 + f6Face<N>(f6off, src, top ? 4 : 5), r.s = 1.0;
	if (kind == FACE_GHOST) p = ghost + (size_t) src * NN, r.s = 1.0;
	r.p = reinterpret_cast<const double2 *>(p);
	return r;
}
template <int N>
__device__ __forceinline__ HaloSrc haloSrc6(int tid, const int32_t *fk, const int32_t *fs, const double *f6, const double *ghost,
                                            const int32_t *f6off)
{
	constexpr int NN = N * N;
	using T2 = Tile2<N>;
	HaloSrc       h;
	h.p      = f6;
	h.stride = 0;
	h.s      = 0.0;
	h.lds    = -1;
	if (tid < 4 * N) {
		const int side = tid / N, t = tid % N;
		const int kind = fk[side], src = fs[side];
		h.lds = (side == 0) ? T2::row(t + 1) + 1 : (side == 1) ? T2::row(t + 1) + N + 2 : (side == 2) ? T2::row(0) + t + 2 : T2::row(N + 1) + t + 2;
		if (kind == FACE_LOCAL) h.p = f6 + f6Face<N>(f6off, src, side ^ 1) + t, h.stride = N, h.s = 1.0;
		if (kind == FACE_GHOST) h.p = ghost + (size_t) src * NN + t, h.stride = N, h.s = 1.0;
	}
	return h;
}

// Ghost planes for coarse/fine faces: ghost = 2*gamma - m with gamma assembled from the
// weights of TriLinInterp.cpp:85-170. One workgroup per coarse/fine face.
// desc[8] = {patch, side, kind (2 = my neighbour is coarser, 3 = my neighbours are finer),
//            quadrant on the coarse face, nbr0, nbr1, nbr2, nbr3}; slot = face index in list.
template <int N>
__global__ void k_cf_ghost3d(const int32_t *__restrict__ desc, const int32_t *__restrict__ slots,
                             const double *__restrict__ u, double *__restrict__ ghost)
{
	constexpr int NN = N * N, NNN = N * N * N;
	const int32_t *d    = desc + (size_t) blockIdx.x * 8;
	const int      p = d[0], s = d[1], kind = d[2], q = d[3];
	const int      ax   = s >> 1;
	const int      sa   = (ax == 0) ? N : 1;        // stride of face coord a
	const int      sb   = (ax == 2) ? N : NN;       // stride of face coord b
	const int      sn   = (ax == 0) ? 1 : (ax == 1 ? N : NN);
	const int      mine = (s & 1) ? (N - 1) * sn : 0; // my face layer
	const int      oth  = (s & 1) ? 0 : (N - 1) * sn; // neighbour's facing layer
	double        *g    = ghost + (size_t) slots[blockIdx.x] * NN;
	const double  *up   = u + (size_t) p * NNN;
	for (int i = threadIdx.x; i < NN; i += blockDim.x) {
		const int a = i % N, b = i / N;
		double    m = up[mine + a * sa + b * sb];
		double    gamma;
		if (kind == 2) {
			const int a0 = a & ~1, b0 = b & ~1;
			double    sum = 0; // the other three cells of the 2x2 block
			for (int bb = 0; bb < 2; bb++)
				for (int aa = 0; aa < 2; aa++)
					if (a0 + aa != a || b0 + bb != b) sum += up[mine + (a0 + aa) * sa + (b0 + bb) * sb];
			const int ca = (a + ((q & 1) ? N : 0)) / 2, cb = (b + ((q & 2) ? N : 0)) / 2;
			// d[4] >= 0: the coarse neighbour is local; else its facing layer was received in a ghost slot
			double    C  = d[4] >= 0 ? u[(size_t) d[4] * NNN + oth + ca * sa + cb * sb]
			                         : ghost[(size_t) (-(d[4] + 2)) * NN + ca + N * cb];
			gamma        = (11 * m - sum) / 12.0 + 4.0 * C / 12.0;
		} else {
			const int qa = (a >= N / 2), qb = (b >= N / 2);
			const int nbq = d[4 + qa + 2 * qb];
			const int fa = 2 * (a - qa * (N / 2)), fb = 2 * (b - qb * (N / 2));
			double    sum = 0;
			for (int bb = 0; bb < 2; bb++)
				for (int aa = 0; aa < 2; aa++)
					sum += 1.0 / 6.0 * (nbq >= 0 ? u[(size_t) nbq * NNN + oth + (fa + aa) * sa + (fb + bb) * sb]
					                             : ghost[(size_t) (-(nbq + 2)) * NN + (fa + aa) + N * (fb + bb)]);
			gamma = 2.0 / 6.0 * m + sum;
		}
		g[i] = 2 * gamma - m;
	}
}

// Multi-rank: copy the face layers other ranks need into one contiguous send buffer.
// Direct-store transport (pushkernels.hpp): a pack kernel can store each face layer straight into the receiving rank's ghost
// slot (dst[i]: where face i of the send order goes) and raise the peers' flags when its last workgroup is done -- pack and
// push in one launch. dst == null: the ordinary pack into the send buffer.
struct PackPush {
	double *const  *dst    = nullptr; // [faces]
	const PushFlag *flags  = nullptr; // [nflags] the peers' flags to raise (pushkernels.hpp: with the checks that go with raising one)
	int             nflags = 0;
	unsigned long long epoch = 0, raise = 0; // this exchange's epoch; the value stored in the flags (the same, but for TE_PUSH_FAULT)
	unsigned       *done     = nullptr; // arrival counter of the workgroups (zero between launches)
	int            *err      = nullptr; // the solver's error word: nothing is stored once a wait has given up
	int            *err_host = nullptr;
};
__device__ __forceinline__ void packPushTail(const PackPush &pp)
{
	if (!pp.dst) return;
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence_system();
		if (__hip_atomic_fetch_add(pp.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
			__hip_atomic_store(pp.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__threadfence_system();
			for (int k = 0; k < pp.nflags; k++) pushRaise(pp.flags[k], pp.epoch, pp.raise, pp.err, pp.err_host);
		}
	}
}
// faces[i] = (patch, side); one workgroup per face; layout (a, b) = remaining axes in order.
template <int N>
__global__ void k_pack_faces3d(const int32_t *__restrict__ faces, const double *__restrict__ u,
                               double *__restrict__ sendbuf, PackPush pp = PackPush())
{
	constexpr int NN = N * N, NNN = N * N * N;
	const int     p = faces[2 * blockIdx.x], s = faces[2 * blockIdx.x + 1];
	const int     ax = s >> 1;
	const int     sa = (ax == 0) ? N : 1, sb = (ax == 2) ? N : NN, sn = (ax == 0) ? 1 : (ax == 1 ? N : NN);
	const double *up = u + (size_t) p * NNN + ((s & 1) ? (N - 1) * sn : 0);
	double       *o  = pp.dst ? pp.dst[blockIdx.x] : sendbuf + (size_t) blockIdx.x * NN;
	if (!(pp.dst && *pp.err))
		for (int i = threadIdx.x; i < NN; i += blockDim.x) o[i] = up[(i % N) * sa + (i / N) * sb];
	packPushTail(pp);
}

// restricted value of coarse cell (hx,hy,hz) of the octant a fine patch covers: the eight fine
// cells summed in the order the reference's scatter loop visits them (AvgRstr.h:95-102: x, then
// y, then z), each divided by 2^D first -> bit-identical to the reference.
template <int N> __device__ __forceinline__ double restrictCell(const double *fp, int hx, int hy, int hz)
{
	constexpr int NN  = N * N;
	double        acc = 0.0;
#pragma unroll
	for (int dz = 0; dz < 2; dz++)
#pragma unroll
		for (int dy = 0; dy < 2; dy++) {
			const double2 v = *reinterpret_cast<const double2 *>(fp + 2 * hx + N * (2 * hy + dy) + NN * (2 * hz + dz));
			acc += v.x / 8;
			acc += v.y / 8;
		}
	return acc;
}

// AvgRstr.h:78-113, gathered per coarse cell (no atomics, no zero-fill pass).
// child[pc*8 + o] = fine patch holding orthant o (or child[pc*8] = source when copy[pc]);
// an entry <= -2 means the child lives on another rank and its already-restricted block
// number -(entry+2) sits in `remote` at remote_off[block].
template <int N>
__global__ __launch_bounds__(256) void k_restrict3d(int Pc, const int32_t *__restrict__ child,
                                                    const int32_t *__restrict__ copy,
                                                    const double *__restrict__ fine,
                                                    const double *__restrict__ remote,
                                                    const int64_t *__restrict__ remote_off,
                                                    double *__restrict__ coarse)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	const size_t  total = (size_t) Pc * NNN;
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total;
	     idx += (size_t) gridDim.x * blockDim.x) {
		const int pc = (int) (idx / NNN), c = (int) (idx % NNN);
		const int x = c % N, y = (c / N) % N, z = c / NN;
		if (copy[pc]) {
			const int src = child[(size_t) pc * 8];
			coarse[idx]   = 0.0 + (src >= 0 ? fine[(size_t) src * NNN + c] : remote[remote_off[-(src + 2)] + c]);
			continue;
		}
		const int ox = x >= H, oy = y >= H, oz = z >= H;
		const int hx = x - ox * H, hy = y - oy * H, hz = z - oz * H;
		const int src = child[(size_t) pc * 8 + ox + 2 * oy + 4 * oz];
		if (src >= 0)
			coarse[idx] = restrictCell<N>(fine + (size_t) src * NNN, hx, hy, hz);
		else
			coarse[idx] = remote[remote_off[-(src + 2)] + hx + H * hy + H * H * hz];
	}
}
// child side of a cross-rank restriction: desc[i] = (fine patch, orthant or -1), block i at off[i]
template <int N>
__global__ __launch_bounds__(256) void k_restrict_pack3d(const int32_t *__restrict__ desc,
                                                         const int64_t *__restrict__ off,
                                                         const double *__restrict__ fine, double *__restrict__ buf)
{
	constexpr int NNN = N * N * N, H = N / 2;
	const int     p = desc[2 * blockIdx.x], o = desc[2 * blockIdx.x + 1];
	const double *fp = fine + (size_t) p * NNN;
	double       *b  = buf + off[blockIdx.x];
	if (o < 0) {
		for (int i = threadIdx.x; i < NNN; i += blockDim.x) b[i] = fp[i];
	} else {
		for (int i = threadIdx.x; i < H * H * H; i += blockDim.x)
			b[i] = restrictCell<N>(fp, i % H, (i / H) % H, i / (H * H));
	}
}

// parent side of a cross-rank restriction when the local children were written by the fused
// residual+restrict kernel: place received block i (desc[i] = coarse patch, orthant or -1)
template <int N>
__global__ __launch_bounds__(256) void k_restrict_unpack3d(const int32_t *__restrict__ desc,
                                                           const int64_t *__restrict__ off,
                                                           const double *__restrict__ buf, double *__restrict__ coarse)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	const int     pc = desc[2 * blockIdx.x], o = desc[2 * blockIdx.x + 1];
	double       *cp = coarse + (size_t) pc * NNN;
	const double *b  = buf + off[blockIdx.x];
	if (o < 0) {
		for (int i = threadIdx.x; i < NNN; i += blockDim.x) cp[i] = b[i];
	} else {
		const int bx = (o & 1) ? H : 0, by = (o & 2) ? H : 0, bz = (o & 4) ? H : 0;
		for (int i = threadIdx.x; i < H * H * H; i += blockDim.x)
			cp[bx + i % H + N * (by + (i / H) % H) + NN * (bz + i / (H * H))] = b[i];
	}
}

// DrctIntp.h:80-113: fine += coarse[parent][(c + orthant offset)/2].
// parent[pf] <= -2: the parent lives on another rank; its octant (or whole patch, orthant -1)
// was received as block -(parent+2) of `remote`.
template <int N>
__global__ __launch_bounds__(256) void k_prolong3d(int Pf, const int32_t *__restrict__ parent,
                                                   const int32_t *__restrict__ orth,
                                                   const double *__restrict__ coarse,
                                                   const double *__restrict__ remote,
                                                   const int64_t *__restrict__ remote_off,
                                                   double *__restrict__ fine)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	const size_t  total = (size_t) Pf * (NNN / 2);
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total;
	     idx += (size_t) gridDim.x * blockDim.x) {
		const int pf = (int) (idx / (NNN / 2)), c = (int) (idx % (NNN / 2)) * 2;
		const int x = c % N, y = (c / N) % N, z = c / NN;
		const int o  = orth[pf];
		const int pa = parent[pf];
		double2  *fp = reinterpret_cast<double2 *>(fine + (size_t) pf * NNN + c);
		double2   v  = *fp;
		if (o >= 0) {
			double cv;
			if (pa >= 0) {
				const int cx = (x + ((o & 1) ? N : 0)) / 2, cy = (y + ((o & 2) ? N : 0)) / 2,
				          cz = (z + ((o & 4) ? N : 0)) / 2;
				cv = coarse[(size_t) pa * NNN + cx + N * cy + NN * cz];
			} else {
				cv = remote[remote_off[-(pa + 2)] + x / 2 + H * (y / 2) + H * H * (z / 2)];
			}
			v.x += cv;
			v.y += cv;
		} else {
			const double *cp = (pa >= 0) ? coarse + (size_t) pa * NNN : remote + remote_off[-(pa + 2)];
			v.x += cp[c];
			v.y += cp[c + 1];
		}
		*fp = v;
	}
}
// parent side of a cross-rank prolongation: desc[i] = (coarse patch, orthant or -1)
template <int N>
__global__ __launch_bounds__(256) void k_prolong_pack3d(const int32_t *__restrict__ desc,
                                                        const int64_t *__restrict__ off,
                                                        const double *__restrict__ coarse, double *__restrict__ buf)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	const int     pc = desc[2 * blockIdx.x], o = desc[2 * blockIdx.x + 1];
	const double *cp = coarse + (size_t) pc * NNN;
	double       *b  = buf + off[blockIdx.x];
	if (o < 0) {
		for (int i = threadIdx.x; i < NNN; i += blockDim.x) b[i] = cp[i];
	} else {
		const int bx = (o & 1) ? H : 0, by = (o & 2) ? H : 0, bz = (o & 4) ? H : 0;
		for (int i = threadIdx.x; i < H * H * H; i += blockDim.x)
			b[i] = cp[bx + i % H + N * (by + (i / H) % H) + NN * (bz + i / (H * H))];
	}
}

// ---- reference block-Jacobi smoother: exact patch solves (FftwPatchSolver.h:173-206) -------
// rhs[p] = f[p] - (2/h^2) * gamma on every face that has a neighbour, gamma = (m + ghost)/2
// (StarPatchOp.h:185-203 with the interface value rebuilt from the ghost).
template <int N>
__global__ __launch_bounds__(256) void k_patch_rhs3d(LevelDev L, const double *__restrict__ u,
                                                     const double *__restrict__ f, double *__restrict__ rhs)
{
	constexpr int NN = N * N, NNN = N * N * N;
	const size_t  total = (size_t) L.P * NNN;
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total;
	     idx += (size_t) gridDim.x * blockDim.x) {
		const int p = (int) (idx / NNN), c = (int) (idx % NNN);
		const int xyz[3] = {c % N, (c / N) % N, c / NN};
		double    v      = f[idx];
		const int st[3]  = {1, N, NN};
#pragma unroll
		for (int ax = 0; ax < 3; ax++) {
			const int a = xyz[ax == 0 ? 1 : 0], b = xyz[ax == 2 ? 1 : 2];
#pragma unroll
			for (int side = 0; side < 2; side++) {
				if (xyz[ax] != (side ? N - 1 : 0)) continue;
				const int s    = 2 * ax + side;
				const int kind = L.face_kind[(size_t) p * 6 + s];
				if (kind < FACE_LOCAL) continue;
				const int    src = L.face_src[(size_t) p * 6 + s];
				const double m   = u[idx];
				double       gh;
				if (kind == FACE_LOCAL)
					gh = u[(size_t) src * NNN + c + (side ? -(N - 1) : (N - 1)) * st[ax]];
				else
					gh = L.ghost[(size_t) src * NN + a + N * b];
				const double gamma = 0.5 * m + 0.5 * gh;
				v -= 2.0 * L.rh2[(size_t) p * 3 + ax] * gamma;
			}
		}
		rhs[idx] = v;
	}
}

// One axis of the dense DST/DCT (DftPatchSolver.h:295-347): out[..i..] = sum_j M[i][j] in[..j..].
// mats: [nplans][6][N*N] row-major (fwd x,y,z then inv x,y,z); lam: [nplans][3][N] = 4 sin^2(.),
// multiplied by the patch's 1/h^2 here;
// plan[p] selects the patch's set. STAGE 0..2 forward, 3..5 inverse. The eigenvalue divide
// (FftwPatchSolver.h:195) rides on stage 2's store, the (2/N)^3 scale on stage 5's.
template <int N, int STAGE>
__global__ __launch_bounds__(256) void k_dst_axis3d(int P, const int32_t *__restrict__ plan,
                                                    const double *__restrict__ mats,
                                                    const double *__restrict__ lam,
                                                    const int32_t *__restrict__ zero_mode,
                                                    const double *__restrict__ rh2,
                                                    const double *__restrict__ in, double *__restrict__ out)
{
	constexpr int NN = N * N, NNN = N * N * N;
	constexpr int AX = STAGE % 3;
	constexpr int ST = (AX == 0) ? 1 : (AX == 1 ? N : NN);
	__shared__ double Ms[NN]; // AX == 0: transposed so lanes (i) hit consecutive banks
	constexpr int BPP = (NNN + 255) / 256; // blocks per patch
	const int     pid = blockIdx.x / BPP;
	const int     blk = blockIdx.x % BPP;
	if (pid >= P) return;
	const int     pl = plan[pid];
	const double *M  = mats + ((size_t) pl * 6 + STAGE) * NN;
	for (int i = threadIdx.x; i < NN; i += blockDim.x) {
		if (AX == 0)
			Ms[(i % N) * N + i / N] = M[i];
		else
			Ms[i] = M[i];
	}
	__syncthreads();
	const double *ip = in + (size_t) pid * NNN;
	double       *op = out + (size_t) pid * NNN;
	const int c = blk * 256 + threadIdx.x;
	if (c < NNN) {
		const int i    = (c / ST) % N;
		const int base = c - i * ST;
		double    acc  = 0.0;
#pragma unroll 8
		for (int j = 0; j < N; j++) {
			const double w = (AX == 0) ? Ms[j * N + i] : Ms[i * N + j];
			acc += w * ip[base + j * ST];
		}
		if (STAGE == 2) {
			const double *lm = lam + (size_t) pl * 3 * N;
			const int     x = c % N, y = (c / N) % N, z = c / NN;
			const double *rh = rh2 + (size_t) pid * 3;
			acc /= -(lm[x] * rh[0] + lm[N + y] * rh[1] + lm[2 * N + z] * rh[2]);
			if (zero_mode[pl] && c == 0) acc = 0.0;
		}
		if (STAGE == 5) acc *= (8.0 / ((double) N * N * N));
		op[c] = acc;
	}
}

// ---- BLAS-1 (Vector.h:190-321) ------------------------------------------------------------------
enum VecOp : int {
	VOP_SET, VOP_SCALE, VOP_SHIFT, VOP_COPY, VOP_ADD, VOP_ADD_SCALED, VOP_ADD_SCALED2,
	VOP_SCALE_THEN_ADD, VOP_SCALE_THEN_ADD_SCALED, VOP_SCALE_THEN_ADD_SCALED2
};
template <int OP>
__global__ __launch_bounds__(256) void k_vecop(size_t n2, double2 *__restrict__ v, const double2 *__restrict__ a,
                                               const double2 *__restrict__ b, double alpha, double beta,
                                               double gamma)
{
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t) gridDim.x * blockDim.x) {
		double2 r;
		if (OP == VOP_SET) {
			r.x = r.y = alpha;
		} else {
			r = v[i];
			double2 av, bv;
			if (OP >= VOP_COPY) av = a[i];
			if (OP == VOP_ADD_SCALED2 || OP == VOP_SCALE_THEN_ADD_SCALED2) bv = b[i];
			switch (OP) {
				case VOP_SCALE: r.x *= alpha; r.y *= alpha; break;
				case VOP_SHIFT: r.x += alpha; r.y += alpha; break;
				case VOP_COPY: r = av; break;
				case VOP_ADD: r.x += av.x; r.y += av.y; break;
				case VOP_ADD_SCALED: r.x += av.x * alpha; r.y += av.y * alpha; break;
				case VOP_ADD_SCALED2:
					r.x += av.x * alpha + bv.x * beta;
					r.y += av.y * alpha + bv.y * beta;
					break;
				case VOP_SCALE_THEN_ADD: r.x = alpha * r.x + av.x; r.y = alpha * r.y + av.y; break;
				case VOP_SCALE_THEN_ADD_SCALED:
					r.x = alpha * r.x + beta * av.x;
					r.y = alpha * r.y + beta * av.y;
					break;
				case VOP_SCALE_THEN_ADD_SCALED2:
					r.x = alpha * r.x + beta * av.x + gamma * bv.x;
					r.y = alpha * r.y + beta * av.y + gamma * bv.y;
					break;
				default: break;
			}
		}
		v[i] = r;
	}
}

// ---- reductions (Vector.h:284-321): wave shuffles -> LDS -> one partial per block -> fixed-order
// final pass (deterministic; no float atomics) ------------------------------------------------------
enum RedOp : int { RED_DOT, RED_SUMSQ, RED_MAXABS };
template <int OP> __device__ __forceinline__ double redCombine(double a, double b)
{
	return (OP == RED_MAXABS) ? fmax(a, b) : a + b;
}
template <int OP> __device__ __forceinline__ double blockReduce(double v)
{
	__shared__ double wsum[16];
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = redCombine<OP>(v, __shfl_down(v, off, 64));
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (lane == 0) wsum[w] = v;
	__syncthreads();
	double r = 0.0;
	if (threadIdx.x == 0) {
		r = wsum[0];
		for (int i = 1; i < (int) (blockDim.x >> 6); i++) r = redCombine<OP>(r, wsum[i]);
	}
	return r;
}
template <int OP>
__global__ __launch_bounds__(256) void k_reduce(size_t n2, const double2 *__restrict__ a,
                                                const double2 *__restrict__ b, double *__restrict__ partial)
{
	double acc = 0.0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t) gridDim.x * blockDim.x) {
		const double2 av = a[i];
		if (OP == RED_DOT) {
			const double2 bv = b[i];
			acc += av.x * bv.x;
			acc += av.y * bv.y;
		} else if (OP == RED_SUMSQ) {
			acc += av.x * av.x;
			acc += av.y * av.y;
		} else {
			acc = fmax(acc, fmax(fabs(av.x), fabs(av.y)));
		}
	}
	acc = blockReduce<OP>(acc);
	if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
template <int OP>
__global__ __launch_bounds__(256) void k_reduce_final(int nparts, const double *__restrict__ partial,
                                                      double *__restrict__ result)
{
	double acc = 0.0;
	for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc = redCombine<OP>(acc, partial[i]);
	acc = blockReduce<OP>(acc);
	if (threadIdx.x == 0) result[0] = acc;
}

// te_vec_checksum: the wrap-around sum of the 64-bit patterns of the values. Integer addition commutes, so the result does
// not depend on the order of the terms -- nor on how a vector is cut over ranks: two vectors with equal checksums on every
// partition hold, with overwhelming probability, the same multiset of bits. One atomic add per wave into *out (zeroed before).
static __global__ __launch_bounds__(256) void k_checksum(size_t n2, const double2 *__restrict__ a, unsigned long long *__restrict__ out)
{
	unsigned long long acc = 0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t) gridDim.x * blockDim.x) {
		const double2 av = a[i];
		acc += (unsigned long long) __double_as_longlong(av.x);
		acc += (unsigned long long) __double_as_longlong(av.y);
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
	if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

// ---- fused BLAS-1 of one BiCGStab iteration (BiCGStab.h:71-104): same expressions, fewer HBM passes ----
// two sums per block, combined by k_reduce_final2 in fixed order
__device__ __forceinline__ void blockReduce2(double &a, double &b)
{
	__shared__ double w0[16], w1[16];
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		a += __shfl_down(a, off, 64);
		b += __shfl_down(b, off, 64);
	}
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (lane == 0) {
		w0[w] = a;
		w1[w] = b;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		a = w0[0];
		b = w1[0];
		for (int i = 1; i < (int) (blockDim.x >> 6); i++) {
			a += w0[i];
			b += w1[i];
		}
	}
}
static __global__ __launch_bounds__(256) void k_reduce_final2(int nparts, const double *__restrict__ partial, double *__restrict__ result)
{
	double a = 0.0, b = 0.0;
	for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
		a += partial[2 * i];
		b += partial[2 * i + 1];
	}
	blockReduce2(a, b);
	if (threadIdx.x == 0) {
		result[0] = a;
		result[1] = b;
	}
}
// Domain::integrate (Domain.h:258-278): per patch, the sum of its cells times the cell volume; one workgroup per
// patch (cells summed in a fixed order), the per-patch values are added on the host in patch order.
// vol[p] = product of the patch's spacings.
static __global__ __launch_bounds__(256) void k_patch_integrals(int nc, const double *__restrict__ v, const double *__restrict__ vol,
                                                         double *__restrict__ out)
{
	const double *p = v + (size_t) blockIdx.x * nc;
	double        acc = 0.0;
	for (int i = threadIdx.x; i < nc; i += blockDim.x) acc += p[i];
	double dummy = 0.0;
	blockReduce2(acc, dummy);
	if (threadIdx.x == 0) out[blockIdx.x] = acc * vol[blockIdx.x];
}

// s = resid; s += ap * (-alpha)          (BiCGStab.h:79-80)
static __global__ __launch_bounds__(256) void k_bicg_s(size_t n2, double2 *__restrict__ s, const double2 *__restrict__ resid,
                                                const double2 *__restrict__ ap, double malpha)
{
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (i >= n2) return;
	const double2 r = resid[i], a = ap[i];
	s[i]            = double2{__builtin_fma(a.x, malpha, r.x), __builtin_fma(a.y, malpha, r.y)}; // (as march3d.hpp fsrcCombine<1>)
}
// (as . s, as . as)                      (BiCGStab.h:87)
static __global__ __launch_bounds__(256) void k_bicg_omega(size_t n2, const double2 *__restrict__ as, const double2 *__restrict__ s,
                                                    double *__restrict__ partial)
{
	double d0 = 0.0, d1 = 0.0;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t) gridDim.x * 256) {
		const double2 a = as[i], b = s[i];
		d0 += a.x * b.x;
		d0 += a.y * b.y;
		d1 += a.x * a.x;
		d1 += a.y * a.y;
	}
	blockReduce2(d0, d1);
	if (threadIdx.x == 0) {
		partial[2 * blockIdx.x]     = d0;
		partial[2 * blockIdx.x + 1] = d1;
	}
}
// x += mp*alpha + ms*omega; resid += ap*(-alpha) + as*(-omega); (resid . rhat, resid . resid)   (BiCGStab.h:90-97,71)
static __global__ __launch_bounds__(256) void k_bicg_update(size_t n2, double2 *__restrict__ x, double2 *__restrict__ resid,
                                                     const double2 *__restrict__ mp, const double2 *__restrict__ ms,
                                                     const double2 *__restrict__ ap, const double2 *__restrict__ as,
                                                     const double2 *__restrict__ rhat, double alpha, double omega,
                                                     double *__restrict__ partial)
{
	double d0 = 0.0, d1 = 0.0;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t) gridDim.x * 256) {
		double2       xv = x[i], rv = resid[i];
		const double2 p = mp[i], q = ms[i], a = ap[i], b = as[i], h = rhat[i];
		xv.x += p.x * alpha + q.x * omega;
		xv.y += p.y * alpha + q.y * omega;
		rv.x += a.x * -alpha + b.x * -omega;
		rv.y += a.y * -alpha + b.y * -omega;
		x[i]     = xv;
		resid[i] = rv;
		d0 += rv.x * h.x;
		d0 += rv.y * h.y;
		d1 += rv.x * rv.x;
		d1 += rv.y * rv.y;
	}
	blockReduce2(d0, d1);
	if (threadIdx.x == 0) {
		partial[2 * blockIdx.x]     = d0;
		partial[2 * blockIdx.x + 1] = d1;
	}
}
// p += ap*(-omega); p = beta*p + resid      (BiCGStab.h:99-100)
static __global__ __launch_bounds__(256) void k_bicg_p(size_t n2, double2 *__restrict__ p, const double2 *__restrict__ ap,
                                                const double2 *__restrict__ resid, double momega, double beta)
{
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (i >= n2) return;
	double2       pv = p[i];
	const double2 a = ap[i], r = resid[i];
	pv.x = __builtin_fma(a.x, momega, pv.x); // (as march3d.hpp fsrcCombine<2>)
	pv.y = __builtin_fma(a.y, momega, pv.y);
	p[i] = double2{__builtin_fma(beta, pv.x, r.x), __builtin_fma(beta, pv.y, r.y)};
}
} // namespace te
