// HIP kernels for the 2D twin of the path (configs C1: one 256^2 patch, C5: 4096^2 in 64^2 patches):
// 5-point cell-centred Laplacian (StarPatchOp<2> == FivePtPatchOperator.h:28-261), interface weights of
// BilinearInterpolator.cpp:61-117, AvgRstr / DrctIntp in 2D. Layout v[p*n*n + x + n*y]; faces W,E,S,N;
// a face cell is addressed by the coordinate along the other axis. n is a run-time value (any even n).
//
// 2D problems are small next to 512^3 (16.8 M sites at most), so these kernels are the simple form:
// one thread per pair of x-adjacent cells, neighbours read straight from global memory (the 4 re-reads
// of a cell are L1/L2 hits); no LDS tiling. Same ghost formulation as kernels3d.hpp.
#pragma once
#include "kernels3d.hpp"

namespace te
{
struct Level2D {
	int32_t        P, n;
	const int32_t *face_kind; // [P*4]
	const int32_t *face_src;  // [P*4]
	const double  *face_kadj; // [P*4]
	const double  *rh2;       // [P*3] (x, y, unused)
	const double  *ghost;     // [nslots*n]
};

// ghost value outside side s of patch p at face coordinate t; `own` = the cell just inside.
// fold = true: physical faces return 0 (their closure is folded into the diagonal by the caller).
__device__ __forceinline__ double ghost2d(const Level2D &L, const double *u, int p, int s, int t, double own, bool fold)
{
	const int n = L.n, kind = L.face_kind[p * 4 + s], src = L.face_src[p * 4 + s];
	if (kind == FACE_DIRICHLET) return fold ? 0.0 : -own;
	if (kind == FACE_NEUMANN) return fold ? 0.0 : own;
	if (kind == FACE_GHOST) return L.ghost[(size_t) src * n + t];
	// neighbour's facing cell
	int cell;
	switch (s) {
		case 0: cell = (n - 1) + n * t; break;
		case 1: cell = n * t; break;
		case 2: cell = t + n * (n - 1); break;
		default: cell = t; break;
	}
	return u[(size_t) src * n * n + cell];
}
__device__ __forceinline__ double kfold2d(const Level2D &L, int p, int s)
{
	const int kind = L.face_kind[p * 4 + s];
	return kind == FACE_DIRICHLET ? 1.0 : (kind == FACE_NEUMANN ? -1.0 : 0.0);
}

// (l - 2c + r)/hx^2 + (ym - 2c + yp)/hy^2 with one fixed rounding sequence: shared by the plain and the fused
// residual kernels, which must agree bit for bit (implicit FMA contraction may differ between kernels)
__device__ __forceinline__ double lap2d(double l, double c, double r, double ym, double yp, double rhx, double rhy)
{
#pragma clang fp contract(off)
	return __builtin_fma(ym - 2 * c + yp, rhy, (l - 2 * c + r) * rhx);
}

// off-diagonal sum of the relaxation, (xl + xr)/hx^2 + (yl + yr)/hy^2, pinned the same way (plain and LDS sweeps)
__device__ __forceinline__ double offdiag2d(double xl, double xr, double yl, double yr, double rhx, double rhy)
{
#pragma clang fp contract(off)
	return __builtin_fma(yl + yr, rhy, (xl + xr) * rhx);
}

// MODE_APPLY / MODE_RESID / MODE_JACOBI as in k_stencil3d
// RED (march3d.hpp StencilRed; te_residual_norm_sq and te_bicgstab, Vector.h:294,319 before their MPI_Allreduce): sums over the result
// formed while it is in registers -- RED_OUT_OUT: sum out^2; RED_OUT_A: sum out * a; RED_OUT_A_OUT: sum out * a and sum out^2 --
// per thread in the order of its grid-stride loop, then wave shuffles -> LDS -> one PAIR per workgroup in partial[2 blockIdx.x],
// [2 blockIdx.x + 1] (RED_OUT_OUT: (sum, 0)); the caller's k_reduce_final2 adds the pairs in a fixed order (a fixed grid:
// deterministic). No second pass over the result: 16 B per site and dot product that a k_reduce launch would read.
template <int MODE, int RED = RED_NONE>
__global__ __launch_bounds__(256) void k_stencil2d(Level2D L, const double *__restrict__ u, const double *__restrict__ f,
                                                   double *__restrict__ out, double omega, double *__restrict__ partial = nullptr,
                                                   const double *__restrict__ a = nullptr)
{
	const int    n = L.n, h = n / 2;
	const size_t total = (size_t) L.P * n * h;
	double       s0 = 0.0, s1 = 0.0;
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t) gridDim.x * blockDim.x) {
		const int     p = (int) (idx / ((size_t) n * h)), q = (int) (idx % ((size_t) n * h));
		const int     y = q / h, x = 2 * (q % h);
		const double *up = u + (size_t) p * n * n;
		const double2 c  = *reinterpret_cast<const double2 *>(up + x + n * y);
		const double  rhx = L.rh2[p * 3], rhy = L.rh2[p * 3 + 1];
		const double  xl = (x > 0) ? up[x - 1 + n * y] : ghost2d(L, u, p, 0, y, c.x, false);
		const double  xr = (x + 2 < n) ? up[x + 2 + n * y] : ghost2d(L, u, p, 1, y, c.y, false);
		double2       ym, yp;
		if (y > 0)
			ym = *reinterpret_cast<const double2 *>(up + x + n * (y - 1));
		else
			ym = double2{ghost2d(L, u, p, 2, x, c.x, false), ghost2d(L, u, p, 2, x + 1, c.y, false)};
		if (y + 1 < n)
			yp = *reinterpret_cast<const double2 *>(up + x + n * (y + 1));
		else
			yp = double2{ghost2d(L, u, p, 3, x, c.x, false), ghost2d(L, u, p, 3, x + 1, c.y, false)};
		double2 lap;
		lap.x = lap2d(xl, c.x, c.y, ym.x, yp.x, rhx, rhy);
		lap.y = lap2d(c.x, c.y, xr, ym.y, yp.y, rhx, rhy);
		double2 r;
		if (MODE == MODE_APPLY) {
			r = lap;
		} else {
			const double2 fv = *reinterpret_cast<const double2 *>(f + (size_t) p * n * n + x + n * y);
			if (MODE == MODE_RESID) {
				r.x = fv.x - lap.x;
				r.y = fv.y - lap.y;
			} else {
				const double *ka = L.face_kadj + p * 4;
				const double  ky = 2.0 + (y == 0 ? ka[2] : 0.0) + (y == n - 1 ? ka[3] : 0.0);
				const double  k0 = 2.0 + (x == 0 ? ka[0] : 0.0), k1 = 2.0 + (x + 2 == n ? ka[1] : 0.0);
				r.x = c.x + omega * (fv.x - lap.x) / -(k0 * rhx + ky * rhy);
				r.y = c.y + omega * (fv.y - lap.y) / -(k1 * rhx + ky * rhy);
			}
		}
		*reinterpret_cast<double2 *>(out + (size_t) p * n * n + x + n * y) = r;
		if (RED == RED_OUT_OUT) {
			s0 += r.x * r.x;
			s0 += r.y * r.y;
		}
		if (RED == RED_OUT_A || RED == RED_OUT_A_OUT) {
			const double2 av = *reinterpret_cast<const double2 *>(a + (size_t) p * n * n + x + n * y);
			s0 += r.x * av.x;
			s0 += r.y * av.y;
			if (RED == RED_OUT_A_OUT) {
				s1 += r.x * r.x;
				s1 += r.y * r.y;
			}
		}
	}
	if (RED != RED_NONE) {
		blockReduce2(s0, s1);
		if (threadIdx.x == 0) partial[2 * blockIdx.x] = s0, partial[2 * blockIdx.x + 1] = s1;
	}
}

// patch-local red-black GS, ghosts frozen at the old iterate u. PHASE 0: red cells ((x+y) even) are
// relaxed from u, black cells copied; PHASE 1: black cells relaxed in `out` from the new red values
// in `out` (neighbour patches / ghost slots still come from the OLD iterate u).
template <int PHASE>
__global__ __launch_bounds__(256) void k_rbgs2d(Level2D L, const double *__restrict__ u, const double *__restrict__ f,
                                                double *__restrict__ out)
{
	const int    n = L.n;
	const size_t total = (size_t) L.P * n * n;
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t) gridDim.x * blockDim.x) {
		const int p = (int) (idx / ((size_t) n * n)), c = (int) (idx % ((size_t) n * n));
		const int x = c % n, y = c / n;
		if (((x + y) & 1) != PHASE) {
			if (PHASE == 0) out[idx] = u[idx];
			continue;
		}
		const double *in  = (PHASE == 0 ? u : out) + (size_t) p * n * n;
		const double  rhx = L.rh2[p * 3], rhy = L.rh2[p * 3 + 1];
		const double  own = u[idx];
		const double  xl = (x > 0) ? in[c - 1] : ghost2d(L, u, p, 0, y, own, true);
		const double  xr = (x + 1 < n) ? in[c + 1] : ghost2d(L, u, p, 1, y, own, true);
		const double  yl = (y > 0) ? in[c - n] : ghost2d(L, u, p, 2, x, own, true);
		const double  yr = (y + 1 < n) ? in[c + n] : ghost2d(L, u, p, 3, x, own, true);
		const double  kx = 2.0 + (x == 0 ? kfold2d(L, p, 0) : 0.0) + (x == n - 1 ? kfold2d(L, p, 1) : 0.0);
		const double  ky = 2.0 + (y == 0 ? kfold2d(L, p, 2) : 0.0) + (y == n - 1 ? kfold2d(L, p, 3) : 0.0);
		const double  o  = offdiag2d(xl, xr, yl, yr, rhx, rhy);
		out[idx]         = (o - f[idx]) * (1.0 / (kx * rhx + ky * rhy)); // (reciprocal of the diagonal, as the 3D kernels and the LDS forms below)
	}
}

// Coarse correction of the post-smoothing sweep (DrctIntp.h:99-106) that is never stored: the sweep reads
// u + coarse[parent][(c + quadrant offset)/2]. coarse == nullptr: none. (Levels without ghost slots only.)
struct Prolong2D {
	const int32_t *parent, *orth;
	const double  *coarse;
};
__device__ __forceinline__ double coarseAt2d(const Prolong2D &ps, int n, int p, int x, int y)
{
	const int o = ps.orth[p];
	return ps.coarse[(size_t) ps.parent[p] * n * n + (x + ((o & 1) ? n : 0)) / 2 + n * ((y + ((o & 2) ? n : 0)) / 2)];
}

// Fused residual + restriction for patches that fit in LDS (Cycle.h:59-65 in one pass): one workgroup per
// fine patch loads u and its ghost ring, each thread forms the four residuals of a coarse cell and adds
// them in AvgRstr's order (restrictCell2d): bit-identical to k_stencil2d<MODE_RESID> + k_restrict2d,
// 18 B per site (read u, f; write 1/4) instead of 34. Parents are local (checked by the host).
static __global__ __launch_bounds__(256) void k_resid_restrict2d_lds(Level2D L, const double *__restrict__ u,
                                                              const double *__restrict__ f, Prolong2D dst /* coarse = coarse f (written) */,
                                                              double *__restrict__ coarse)
{
	extern __shared__ __attribute__((aligned(16))) double tile2d[]; // (n+2) x (n+2)
	const int     n = L.n, lw = n + 2, nn = n * n, h = n / 2;
	const int     p = blockIdx.x, tid = threadIdx.x;
	const double *up = u + (size_t) p * nn;
	const double *fp = f + (size_t) p * nn;
	const double  rhx = L.rh2[p * 3], rhy = L.rh2[p * 3 + 1];
	for (int i = tid; i < nn / 2; i += blockDim.x) {
		const int     y = (2 * i) / n, x = (2 * i) % n;
		const double2 v = reinterpret_cast<const double2 *>(up)[i];
		tile2d[(y + 1) * lw + x + 1] = v.x;
		tile2d[(y + 1) * lw + x + 2] = v.y;
	}
	for (int i = tid; i < 4 * n; i += blockDim.x) { // ghost ring (physical faces: -own / +own)
		const int    s = i / n, t = i % n;
		const double own = up[s == 0 ? n * t : (s == 1 ? n - 1 + n * t : (s == 2 ? t : t + n * (n - 1)))];
		const int    idx = (s == 0) ? (t + 1) * lw : (s == 1) ? (t + 1) * lw + n + 1 : (s == 2) ? t + 1 : (n + 1) * lw + t + 1;
		tile2d[idx]      = ghost2d(L, u, p, s, t, own, false);
	}
	__syncthreads();
	const int pa = dst.parent[p], o = dst.orth[p];
	double   *cp = coarse + (size_t) pa * nn;
	if (o < 0) { // copy-through: the residual lands unchanged in the coarse patch
		for (int i = tid; i < nn; i += blockDim.x) {
			const int     x = i % n, y = i / n;
			const double *t0 = tile2d + (y + 1) * lw + x + 1;
			cp[i]            = 0.0 + (fp[i] - lap2d(t0[-1], t0[0], t0[1], t0[-lw], t0[lw], rhx, rhy));
		}
		return;
	}
	cp += ((o & 1) ? h : 0) + n * ((o & 2) ? h : 0);
	for (int i = tid; i < h * h; i += blockDim.x) {
		const int hx = i % h, hy = i / h;
		double    acc = 0.0; // AvgRstr.h:95-102 order: x then y, each /(1 << D)
#pragma unroll
		for (int dy = 0; dy < 2; dy++) {
			const int     y = 2 * hy + dy, x = 2 * hx;
			const double *t0 = tile2d + (y + 1) * lw + x + 1;
			const double2 fv = *reinterpret_cast<const double2 *>(fp + x + n * y);
			acc += (fv.x - lap2d(t0[-1], t0[0], t0[1], t0[-lw], t0[lw], rhx, rhy)) / 4;
			acc += (fv.y - lap2d(t0[0], t0[1], t0[2], t0[1 - lw], t0[1 + lw], rhx, rhy)) / 4;
		}
		cp[hx + n * hy] = acc;
	}
}

// ---- the 3D fusions in 2D (opts.fuse = 2 and 3; levels whose patches fit in LDS, all parents and neighbours local) ----
// v = S(0, f): the zero-guess sweep of one patch entirely in LDS -- the arithmetic of k_rbgs2d_lds<ZERO> (the halo ring of
// a zero iterate is zero). The tile must hold zeros on entry; on exit it holds v, ring still zero. The right-hand side
// sits in registers (fr[k] = the x-pair number tid + 256 k of the patch, loaded once, 16 B per lane): the kernels that
// sweep twice read f once, and no global load sits between two barriers.
constexpr int F2D_MAX = 8; // pairs per thread: n <= 64 with 256 threads
struct F2D { // (named scalars: an array of double2 indexed in an unrolled loop ended up in scratch memory)
	double x0, y0, x1, y1, x2, y2, x3, y3, x4, y4, x5, y5, x6, y6, x7, y7;
};
#define TE_F2D_EACH(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
template <int TPB> __device__ __forceinline__ void loadF2d(const double *fp, int nn, F2D &fr)
{
#define TE_LD(K)                                                                                          \
	{                                                                                                     \
		const int     i = threadIdx.x + K * TPB;                                                          \
		const double2 v = (i < nn / 2) ? reinterpret_cast<const double2 *>(fp)[i] : double2{0.0, 0.0};    \
		fr.x##K = v.x, fr.y##K = v.y;                                                                     \
	}
	TE_F2D_EACH(TE_LD)
#undef TE_LD
}
__device__ __forceinline__ void idiag2d(const Level2D &L, int p, double rhx, double rhy, double *idg)
{
	if (threadIdx.x < 9) {
		const int    cx = threadIdx.x % 3, cy = threadIdx.x / 3;
		const double kx = 2.0 + (cx == 0 ? kfold2d(L, p, 0) : 0.0) + (cx == 2 ? kfold2d(L, p, 1) : 0.0);
		const double ky = 2.0 + (cy == 0 ? kfold2d(L, p, 2) : 0.0) + (cy == 2 ? kfold2d(L, p, 3) : 0.0);
		idg[threadIdx.x] = 1.0 / (kx * rhx + ky * rhy);
	}
}
// The tile of the fused kernels keeps the two colours apart: cell (x, y), x and y in -1 .. n (halo ring included), sits in
// plane (x + y) & 1 at [(y + 1) * lwh + ((x + 1) >> 1)], lwh = n/2 + 1. A half sweep reads one plane and writes the other, and
// the lanes of a row touch consecutive doubles (in the natural layout every access of a half sweep has stride 2: two-way
// bank conflicts on each ds_read_b64). Same size as the natural tile: 2 (n + 2)(n/2 + 1) = (n + 2)^2.
struct Tile2D {
	double *t;
	int     lwh, cs;
	__device__ Tile2D(double *base, int n) : t(base), lwh(n / 2 + 1), cs((n + 2) * (n / 2 + 1)) {}
	__device__ __forceinline__ double &at(int x, int y) const { return t[((x + y) & 1) * cs + (y + 1) * lwh + ((x + 1) >> 1)]; }
};
// Pair K of a thread (the x-pair number i = tid + 256 K of the patch): row y, pair column q (cells x = 2q, 2q + 1),
// o = y & 1. Its red cell ((x + y) even) is x = 2q + o and sits at [y + 1][q + o] of plane 0, its black cell x = 2q + 1 - o
// at [y + 1][q + 1 - o] of plane 1; a red cell's neighbours are [y + 1][q], [y + 1][q + 1], [y][q + o], [y + 2][q + o] of the black
// plane, a black cell's the same with o -> 1 - o in the red plane.
struct Pair2D {
	int  y, q, o;
	bool live;
};
__device__ __forceinline__ double fr_x(const F2D &fr, int K)
{
	switch (K) { // (K is a constant after unrolling)
#define TE_FX(J) \
	case J: return fr.x##J;
		TE_F2D_EACH(TE_FX)
#undef TE_FX
	}
	return 0.0;
}
__device__ __forceinline__ double fr_y(const F2D &fr, int K)
{
	switch (K) {
#define TE_FY(J) \
	case J: return fr.y##J;
		TE_F2D_EACH(TE_FY)
#undef TE_FY
	}
	return 0.0;
}
template <int NC, int TPB> __device__ __forceinline__ Pair2D pairOf(int K, int n)
{
	const int i = threadIdx.x + K * TPB, m = NC ? NC : n;
	return Pair2D{(2 * i) / m, ((2 * i) % m) >> 1, ((2 * i) / m) & 1, i < m * m / 2};
}
// reciprocal diagonal of a cell: from the table in LDS, or -- NC = 64, where a thread's cells of one colour all have the same x
// class and only its first and last pair can lie on the bottom / top row -- from two registers (mid[colour])
template <int NC, int TPB> __device__ __forceinline__ double idgOf(const double *idg, const double *mid, int K, int colour, int x, int y, int n)
{
	if (NC == 64 && K >= 1 && K <= NC * NC / 2 / TPB - 2) return mid[colour];
	const int cx = (x == 0) ? 0 : (x == n - 1 ? 2 : 1), cy = (y == 0) ? 0 : (y == n - 1 ? 2 : 1);
	return idg[cx + 3 * cy];
}
__device__ __forceinline__ double addRounded2d(double a, double b)
{
#pragma clang fp contract(off)
	return a + b; // (never folded into the multiplication that produced a)
}
// v = S(0, f), the zero-guess sweep of one patch: red cells see zero neighbours only, (0 - f) / diag, and touch no LDS but to
// store; black cells read the new red values, their ghosts (zero: the halo of a zero iterate, physical faces folded into the
// diagonal) masked rather than read, so that the ring may already hold the next sweep's halo. ADDC: the black plane receives
// v + c (the red cells of an iterate are never read by the sweep that follows it). rv / bv: the values of the thread's pairs.
// The arithmetic of k_rbgs2d_lds<ZERO>: bit-identical.
template <int NC, int TPB, bool ADDC>
__device__ __forceinline__ void zeroSweep2d(const Tile2D &T, const double *idg, const double *mid, const F2D &fr, const double *cr, int n,
                                            double rhx, double rhy, double *rv, double *bv)
{
	double *R = T.t, *B = T.t + T.cs;
#define TE_R1(K)                                                                                     \
	{                                                                                                \
		const Pair2D pr = pairOf<NC, TPB>(K, n);                                                          \
		if (pr.live) {                                                                               \
			double fa = fr.x##K, fb = fr.y##K;                                                       \
			asm volatile("" : "+v"(fa), "+v"(fb));                                                   \
			rv[K] = (0.0 - (pr.o ? fb : fa)) * idgOf<NC, TPB>(idg, mid, K, 0, 2 * pr.q + pr.o, pr.y, n);  \
			R[(pr.y + 1) * T.lwh + pr.q + pr.o] = rv[K];                                             \
		}                                                                                            \
	}
	TE_F2D_EACH(TE_R1)
#undef TE_R1
	ldsBarrier();
#define TE_B1(K)                                                                                     \
	{                                                                                                \
		const Pair2D pr = pairOf<NC, TPB>(K, n);                                                          \
		if (pr.live) {                                                                               \
			const int     x = 2 * pr.q + 1 - pr.o, c = pr.q + 1 - pr.o;                              \
			const double *r0 = R + (pr.y + 1) * T.lwh;                                               \
			double        xl = r0[pr.q], xr = r0[pr.q + 1], yl = r0[c - T.lwh], yr = r0[c + T.lwh];  \
			xl = (x == 0) ? 0.0 : xl, xr = (x == n - 1) ? 0.0 : xr;                                  \
			yl = (pr.y == 0) ? 0.0 : yl, yr = (pr.y == n - 1) ? 0.0 : yr;                            \
			double fa = fr.x##K, fb = fr.y##K;                                                       \
			asm volatile("" : "+v"(fa), "+v"(fb));                                                   \
			bv[K] = (offdiag2d(xl, xr, yl, yr, rhx, rhy) - (pr.o ? fa : fb)) * idgOf<NC, TPB>(idg, mid, K, 1, x, pr.y, n); \
			B[(pr.y + 1) * T.lwh + c] = ADDC ? addRounded2d(bv[K], cr[K]) : bv[K];                   \
		}                                                                                            \
	}
	TE_F2D_EACH(TE_B1)
#undef TE_B1
	ldsBarrier();
}
// one sweep over the tile (ring = the frozen halo), the new values of the thread's pairs to rv / bv; the black plane of the
// result is not stored (nobody reads it: the caller writes the pairs to memory)
template <int NC, int TPB>
__device__ __forceinline__ void lastSweep2d(const Tile2D &T, const double *idg, const double *mid, const F2D &fr, int n, double rhx, double rhy,
                                            double *rv, double *bv)
{
	double *R = T.t, *B = T.t + T.cs;
#define TE_R2(K)                                                                                     \
	{                                                                                                \
		const Pair2D pr = pairOf<NC, TPB>(K, n);                                                          \
		if (pr.live) {                                                                               \
			const int     c  = pr.q + pr.o;                                                          \
			const double *b0 = B + (pr.y + 1) * T.lwh;                                               \
			const double  o  = offdiag2d(b0[pr.q], b0[pr.q + 1], b0[c - T.lwh], b0[c + T.lwh], rhx, rhy); \
			double        fa = fr.x##K, fb = fr.y##K;                                                \
			asm volatile("" : "+v"(fa), "+v"(fb));                                                   \
			rv[K] = (o - (pr.o ? fb : fa)) * idgOf<NC, TPB>(idg, mid, K, 0, 2 * pr.q + pr.o, pr.y, n);    \
			R[(pr.y + 1) * T.lwh + c] = rv[K];                                                       \
		}                                                                                            \
	}
	TE_F2D_EACH(TE_R2)
#undef TE_R2
	ldsBarrier();
#define TE_B2(K)                                                                                     \
	{                                                                                                \
		const Pair2D pr = pairOf<NC, TPB>(K, n);                                                          \
		if (pr.live) {                                                                               \
			const int     c  = pr.q + 1 - pr.o;                                                      \
			const double *r0 = R + (pr.y + 1) * T.lwh;                                               \
			const double  o  = offdiag2d(r0[pr.q], r0[pr.q + 1], r0[c - T.lwh], r0[c + T.lwh], rhx, rhy); \
			double        fa = fr.x##K, fb = fr.y##K;                                                \
			asm volatile("" : "+v"(fa), "+v"(fb));                                                   \
			bv[K] = (o - (pr.o ? fa : fb)) * idgOf<NC, TPB>(idg, mid, K, 1, 2 * pr.q + 1 - pr.o, pr.y, n); \
		}                                                                                            \
	}
	TE_F2D_EACH(TE_B2)
#undef TE_B2
}
// the pairs of a thread to memory, 16 B per lane
template <int NC, int TPB> __device__ __forceinline__ void storePairs2d(double *op, const double *rv, const double *bv, int n)
{
#pragma unroll
	for (int K = 0; K < F2D_MAX; K++) {
		const Pair2D pr = pairOf<NC, TPB>(K, n);
		if (pr.live) reinterpret_cast<double2 *>(op)[threadIdx.x + K * TPB] = pr.o ? double2{bv[K], rv[K]} : double2{rv[K], bv[K]};
	}
}
// table of reciprocal diagonals to LDS, the thread's two mid-row values to registers (see idgOf); ends with a barrier
template <int NC, int TPB> __device__ __forceinline__ void idiagSetup2d(const Level2D &L, int p, double rhx, double rhy, double *idg, double *mid, int n)
{
	idiag2d(L, p, rhx, rhy, idg);
	ldsBarrier();
	if (NC == 64) {
		const Pair2D pr = pairOf<NC, TPB>(1, n);
		const int    xr = 2 * pr.q + pr.o, xb = 2 * pr.q + 1 - pr.o;
		mid[0] = idg[((xr == 0) ? 0 : (xr == n - 1 ? 2 : 1)) + 3];
		mid[1] = idg[((xb == 0) ? 0 : (xb == n - 1 ? 2 : 1)) + 3];
	}
}
// The sweep k_rbgs2d<0> + k_rbgs2d<1> in ONE pass for patches that fit in LDS (n <= 64: (n+2)^2 doubles = 34 KiB): one
// workgroup per patch loads u and its frozen halo ring once, relaxes red then black on the colour-split tile and stores the
// result from registers: 24 B per site instead of two passes over u, f and out. Bit-identical to the two-pass form.
// Only the black cells of the old iterate go to LDS (the red half sweep reads black neighbours and overwrites red cells).
// ZERO: the iterate is zero (first sweep of a cycle): u and its ghosts are never read (16 B per site).
// PROLONG: the iterate is u + P(coarse) (see Prolong2D), formed while loading (26 B per site).
template <bool ZERO, bool PROLONG, int NC, int TPB = 256>
__global__ __launch_bounds__(TPB) void k_rbgs2d_lds(Level2D L, const double *__restrict__ u, const double *__restrict__ f,
                                                    double *__restrict__ out, Prolong2D ps)
{
	extern __shared__ __attribute__((aligned(16))) double tile2d[]; // Tile2D, then the 9 reciprocals of the diagonal
	const int     n = NC ? NC : L.n, nn = n * n;
	const int     p = blockIdx.x, tid = threadIdx.x;
	const double *up = u + (size_t) p * nn;
	const Tile2D  T(tile2d, n);
	double       *idg = tile2d + 2 * T.cs;
	const double  rhx = L.rh2[p * 3], rhy = L.rh2[p * 3 + 1];
	F2D           fr;
	double        rv[F2D_MAX], bv[F2D_MAX], mid[2];
	loadF2d<TPB>(f + (size_t) p * nn, nn, fr);
	if (!ZERO) {
#pragma unroll
		for (int K = 0; K < F2D_MAX; K++) {
			const Pair2D pr = pairOf<NC, TPB>(K, n);
			if (!pr.live) continue;
			double2 v = reinterpret_cast<const double2 *>(up)[tid + K * TPB];
			if (PROLONG) {
				const double c = coarseAt2d(ps, n, p, 2 * pr.q, pr.y); // both cells of the pair share the coarse cell
				v.x += c, v.y += c;
			}
			T.t[T.cs + (pr.y + 1) * T.lwh + pr.q + 1 - pr.o] = pr.o ? v.x : v.y; // the black cell: x = 2q + 1 - o
		}
		for (int i = tid; i < 4 * n; i += blockDim.x) { // halo ring: frozen ghosts (physical faces folded -> 0)
			const int s = i / n, t = i % n;
			double    g = ghost2d(L, u, p, s, t, 0.0, true);
			if (PROLONG && L.face_kind[p * 4 + s] == FACE_LOCAL) { // the neighbour's facing cell carries its own correction
				const int src = L.face_src[p * 4 + s];
				g += coarseAt2d(ps, n, src, s == 0 ? n - 1 : (s == 1 ? 0 : t), s == 2 ? n - 1 : (s == 3 ? 0 : t));
			}
			T.at(s == 0 ? -1 : (s == 1 ? n : t), s == 2 ? -1 : (s == 3 ? n : t)) = g;
		}
	}
	idiagSetup2d<NC, TPB>(L, p, rhx, rhy, idg, mid, n);
	if (ZERO)
		zeroSweep2d<NC, TPB, false>(T, idg, mid, fr, nullptr, n, rhx, rhy, rv, bv);
	else
		lastSweep2d<NC, TPB>(T, idg, mid, fr, n, rhx, rhy, rv, bv);
	storePairs2d<NC, TPB>(out + (size_t) p * nn, rv, bv, n);
}

// Cycle.h:57-65 for the first sweep of a cycle in one pass over f (the 2D twin of k_rbgs_zero_resid3d): u = S(0, f),
// coarse f = AvgRstr(f - A u) with a zero ghost on faces that have a neighbour (k_restrict_fixup2d adds that term from the
// neighbours' edge layers afterwards). STORE_U: u is written (16 + 2 B per site); otherwise only its four edge layers
// e4 [P][4][n] (opts.fuse = 3: k_rbgs_resweep_prolong2d_lds recomputes u from f): 8 + 2 B per site.
// NC: the patch size as a compile-time constant (0: run-time L.n) -- every cell's (x, y) comes from an integer division by n,
// dozens of instructions each at run time, shifts for NC = 64 (config C5)
// A parent on another rank (dst.parent < -1): the restricted block goes to `remote` (h x h, shipped afterwards), as in 3D.
// The ghost terms the kernel below leaves out of the restricted residual, for ONE fine patch p whose restricted block starts at
// cb (row stride cs): for every face with a neighbour, -(1/h^2)/4 * (the two neighbour values behind a pair of face cells) belongs
// to the coarse cell behind the pair. edges = e4 of the new iterate, or null: read them from u. A neighbour on another rank: its
// edge arrived in a ghost slot. Shared by k_restrict_fixup2d (a pass of its own) and the FOLD prologue of the coarse level's
// pre-sweep (the reader of the coarse right-hand side completes it itself): the same expressions, bit-identical.
struct Fixup2D {
	const Level2D &L;
	const double  *u, *e4;
	int            p;
	double        *cb;
	int            cs;
	// own: the residual that is being completed was formed with the PATCH operator (exact patch solves: faces with a neighbour
	// closed as homogeneous Dirichlet, ghost = -m, StarPatchOp.h:204-319) instead of a zero ghost: the missing term is
	// -(g + m)/h^2 = -2 gamma/h^2 with m = this patch's own value behind the face (u: the stored iterate). As k_restrict_fixup3d<N, OWN>.
	bool own = false;
	// the sum of face s for the coarse cell number i along it
	__device__ __forceinline__ double term(int s, int kind, int i) const
	{
		const int    n = L.n, nn = n * n;
		const int    src = L.face_src[p * 4 + s], ax = s >> 1;
		const double w   = -L.rh2[p * 3 + ax];
		double       acc = 0.0;
#pragma unroll
		for (int d = 0; d < 2; d++) {
			const int t = 2 * i + d;
			double    g;
			if (kind == FACE_GHOST)
				g = L.ghost[(size_t) src * n + t];
			else if (e4)
				g = e4[((size_t) src * 4 + (s ^ 1)) * n + t];
			else
				g = u[(size_t) src * nn + (s == 0 ? n - 1 + n * t : (s == 1 ? n * t : (s == 2 ? t + n * (n - 1) : t)))];
			if (own) g = g + u[(size_t) p * nn + (s == 0 ? n * t : (s == 1 ? n - 1 + n * t : (s == 2 ? t : t + n * (n - 1))))];
			acc += (w * g) / 4;
		}
		return acc;
	}
	__device__ __forceinline__ int cell(int s, int i) const
	{
		const int h = L.n / 2;
		return (s >> 1) == 0 ? ((s & 1) ? h - 1 : 0) + cs * i : i + cs * ((s & 1) ? h - 1 : 0);
	}
};
// FOLD: where the finer level's pre-sweep left its neighbours' edge layers (and which fine patch sits in which quadrant of a patch
// of this level): see k_rbgs_zero_resid2d_lds
struct Fold2D {
	Level2D        fine;
	const double  *u, *e4;  // the finer level's new iterate (stored), or its four edge layers
	const int32_t *child;   // [P][4]: the fine patch in quadrant ox + 2 oy
};

// FOLD: this level's right-hand side was produced by the same kernel one level up WITHOUT its ghost terms, and no
// k_restrict_fixup2d pass has run: the workgroup adds them to its own patch of f first (the fix-up's expressions in the fix-up's
// order -- W/E terms, barrier, S/N terms -- on global memory, which the waves of one workgroup see coherently after a barrier:
// they share the CU's L1), then proceeds as usual. A launch of 3-12 us per level becomes a prologue of three dependent loads; the
// corrected f is what the post-sweep reads later. 4 <= n <= 64, every child local (the host decides).
template <bool STORE_U, int NC, int TPB = 256, bool FOLD = false>
__global__ __launch_bounds__(TPB) void k_rbgs_zero_resid2d_lds(Level2D L, std::conditional_t<FOLD, double *, const double *__restrict__> f,
                                                               double *__restrict__ out, double *__restrict__ e4, Prolong2D dst,
                                                               double *__restrict__ coarse, double *__restrict__ remote,
                                                               const int64_t *__restrict__ remote_off, Fold2D fold = Fold2D())
{
	extern __shared__ __attribute__((aligned(16))) double tile2d[]; // the colour-split tile (Tile2D), then 9 diagonals
	const int     n = NC ? NC : L.n, nn = n * n, h = n / 2;
	const int     p = blockIdx.x, tid = threadIdx.x;
	if constexpr (FOLD) {
		static_assert(TPB % 128 == 0, "a quadrant's four faces take 128 threads");
#pragma unroll 1
		for (int base = 0; base < 4; base += TPB / 128) {
			const int     qd = base + (tid >> 7), t = tid & 127, s = t >> 5, i = t & 31;
			const int     c  = fold.child[(size_t) p * 4 + qd];
			double       *cb = f + (size_t) p * nn + ((qd & 1) ? h : 0) + n * ((qd & 2) ? h : 0);
			const Fixup2D fx{fold.fine, fold.u, fold.e4, c, cb, n};
			const int     kind = fold.fine.face_kind[c * 4 + s];
			const bool    actf = i < h && kind >= FACE_LOCAL;
			const double  acc  = actf ? fx.term(s, kind, i) : 0.0;
			if (actf && s < 2) cb[fx.cell(s, i)] += acc;
			__syncthreads();
			if (actf && s >= 2) cb[fx.cell(s, i)] += acc;
			__syncthreads();
		}
	}
	const double *fp = f + (size_t) p * nn;
	const Tile2D  T(tile2d, n);
	double       *idg = tile2d + 2 * T.cs;
	const double  rhx = L.rh2[p * 3], rhy = L.rh2[p * 3 + 1];
	F2D           fr;
	double        rv[F2D_MAX], bv[F2D_MAX], mid[2];
	loadF2d<TPB>(fp, nn, fr);
	idiagSetup2d<NC, TPB>(L, p, rhx, rhy, idg, mid, n);
	zeroSweep2d<NC, TPB, false>(T, idg, mid, fr, nullptr, n, rhx, rhy, rv, bv);
	// the new iterate, or only its four edge layers: from the registers that hold the pairs
	if (STORE_U) {
		storePairs2d<NC, TPB>(out + (size_t) p * nn, rv, bv, n);
	} else {
		// (from the tile, which is complete and no longer written: one thread per edge value, 512 contiguous bytes per side --
		// from the registers that hold the pairs the W and E columns were 64 single 8-byte stores each)
		double *e = e4 + (size_t) p * 4 * n;
		for (int i = tid; i < 4 * n; i += TPB) {
			const int s = i / n, t = i % n;
			e[i]        = T.at(s == 0 ? 0 : (s == 1 ? n - 1 : t), s == 2 ? 0 : (s == 3 ? n - 1 : t));
		}
	}
	// residual and restriction; ghosts: physical faces -own / +own (StarPatchOp.h:39-65), faces with a neighbour 0
	const int pa = dst.parent[p], o = dst.orth[p];
	double   *cp = pa >= 0 ? coarse + (size_t) pa * nn + ((o & 1) ? h : 0) + n * ((o & 2) ? h : 0) : remote + remote_off[-(pa + 2)];
	const int cs = pa >= 0 ? n : h; // row stride of the destination
	const int k0 = L.face_kind[p * 4], k1 = L.face_kind[p * 4 + 1], k2 = L.face_kind[p * 4 + 2], k3 = L.face_kind[p * 4 + 3];
	auto      gh = [](int kind, double own) { return kind == FACE_DIRICHLET ? -own : (kind == FACE_NEUMANN ? own : 0.0); };
	if (NC == 64) {
		// a thread forms the residual of its own pairs (values and f in registers); rows y and y ^ 1 sit in the two halves of
		// a wave: the even row's lanes fetch the odd row's two quarters and add them in AvgRstr's order
		const double *R = T.t, *B = T.t + T.cs;
#pragma unroll
		for (int K = 0; K < F2D_MAX; K++) {
			const Pair2D  pr = pairOf<NC, TPB>(K, n);
			if (!pr.live) continue; // (the same for the whole workgroup)
			const int     y = pr.y, q = pr.q, ro = (y + 1) * T.lwh;
			// cell x = 2q has colour o (its plane A), cell 2q + 1 the other (plane Bp)
			const double *A = pr.o ? B : R, *Bp = pr.o ? R : B;
			const double  c0 = pr.o ? bv[K] : rv[K], c1 = pr.o ? rv[K] : bv[K];
			// (x - 1, y) has colour of cell x + 1: plane Bp at column (2q) >> 1 = q; (x + 2, y) has cell x's colour: plane A at q + 1
			double xl = Bp[ro + q], xr = A[ro + q + 1];
			// (x, y -+ 1): the other colour than cell x: plane Bp at (2q + 1) >> 1 = q; (x + 1, y -+ 1): plane A at (2q + 2) >> 1 = q + 1
			double d0 = Bp[ro - T.lwh + q], u0 = Bp[ro + T.lwh + q], d1 = A[ro - T.lwh + q + 1], u1 = A[ro + T.lwh + q + 1];
			xl = (q == 0) ? gh(k0, c0) : xl, xr = (q == h - 1) ? gh(k1, c1) : xr;
			if (y == 0) d0 = gh(k2, c0), d1 = gh(k2, c1);
			if (y == n - 1) u0 = gh(k3, c0), u1 = gh(k3, c1);
			const double ra = (fr_x(fr, K) - lap2d(xl, c0, c1, d0, u0, rhx, rhy)) / 4;
			const double rb = (fr_y(fr, K) - lap2d(c0, c1, xr, d1, u1, rhx, rhy)) / 4;
			const double pa_ = __shfl_xor(ra, 32), pb_ = __shfl_xor(rb, 32);
			if (!pr.o) {
				double acc = 0.0; // AvgRstr.h:95-102 order: x then y, each /(1 << D)
				acc += ra, acc += rb, acc += pa_, acc += pb_;
				cp[q + cs * (y >> 1)] = acc;
			}
		}
	} else {
		for (int i = tid; i < h * h; i += blockDim.x) { // as k_resid_restrict2d_lds
			const int hx = i % h, hy = i / h;
			double    acc = 0.0;
#pragma unroll
			for (int dy = 0; dy < 2; dy++) {
				const int     y = 2 * hy + dy, x = 2 * hx;
				const double2 fv = *reinterpret_cast<const double2 *>(fp + x + n * y);
				const double  c0 = T.at(x, y), c1 = T.at(x + 1, y);
				const double  xl = x > 0 ? T.at(x - 1, y) : gh(k0, c0), xr = x + 2 < n ? T.at(x + 2, y) : gh(k1, c1);
				const double  d0 = y > 0 ? T.at(x, y - 1) : gh(k2, c0), d1 = y > 0 ? T.at(x + 1, y - 1) : gh(k2, c1);
				const double  u0 = y + 1 < n ? T.at(x, y + 1) : gh(k3, c0), u1 = y + 1 < n ? T.at(x + 1, y + 1) : gh(k3, c1);
				acc += (fv.x - lap2d(xl, c0, c1, d0, u0, rhx, rhy)) / 4;
				acc += (fv.y - lap2d(c0, c1, xr, d1, u1, rhx, rhy)) / 4;
			}
			cp[hx + cs * hy] = acc;
		}
	}
}
// the ghost terms the kernel above left out (Fixup2D), as a pass of its own: faces in the order W,E,S,N. A parent on another rank:
// the block in `remote`.
static __global__ __launch_bounds__(128) void k_restrict_fixup2d(Level2D L, const double *__restrict__ u, const double *__restrict__ e4,
                                                          Prolong2D dst, double *__restrict__ coarse, double *__restrict__ remote,
                                                          const int64_t *__restrict__ remote_off, bool own = false)
{
	// own (interfaceResidRestrict2d): the restricted residual IS these terms -- the block starts from zero, written here (no memset
	// pass over the coarse vector; the waves of a workgroup see each other's global stores after a barrier)
	const int n = L.n, nn = n * n, h = n / 2, p = blockIdx.x;
	const int pa = dst.parent[p], o = dst.orth[p];
	double   *cb = pa >= 0 ? coarse + (size_t) pa * nn + ((o & 1) ? h : 0) + n * ((o & 2) ? h : 0) : remote + remote_off[-(pa + 2)];
	const int     cs = pa >= 0 ? n : h;
	if (own) {
		for (int i = threadIdx.x; i < h * h; i += blockDim.x) cb[i % h + cs * (i / h)] = 0.0;
		__syncthreads();
	}
	const Fixup2D fx{L, u, e4, p, cb, cs, own};
	auto          term = [&](int s, int kind, int i) { return fx.term(s, kind, i); };
	auto          cell = [&](int s, int i) { return fx.cell(s, i); };
	if (h >= 2 && h <= 32) {
		// one 32-lane group per face, all four sums in flight at once; W and E touch different cells, S and N too: two rounds of
		// additions (a corner cell takes its W/E term, then its S/N term -- the order of the face-by-face loop below)
		const int  s = threadIdx.x >> 5, i = threadIdx.x & 31, kind = L.face_kind[p * 4 + s];
		const bool act = i < h && kind >= FACE_LOCAL;
		const double acc = act ? term(s, kind, i) : 0.0;
		if (act && s < 2) cb[cell(s, i)] += acc;
		__syncthreads();
		if (act && s >= 2) cb[cell(s, i)] += acc;
		return;
	}
	for (int s = 0; s < 4; s++) {
		const int kind = L.face_kind[p * 4 + s];
		if (kind >= FACE_LOCAL)
			for (int i = threadIdx.x; i < h; i += blockDim.x) cb[cell(s, i)] += term(s, kind, i);
		__syncthreads();
	}
}
// the edge layers other ranks need, from e4 (the iterate itself was never stored)
static __global__ void k_pack_edges2d(int n, const int32_t *__restrict__ faces, const double *__restrict__ e4, double *__restrict__ sendbuf)
{
	const int p = faces[2 * blockIdx.x], s = faces[2 * blockIdx.x + 1];
	for (int i = threadIdx.x; i < n; i += blockDim.x) sendbuf[(size_t) blockIdx.x * n + i] = e4[((size_t) p * 4 + s) * n + i];
}
// restricted blocks that arrived from children on other ranks: into their quadrant of the coarse patch (as k_restrict_unpack3d)
static __global__ __launch_bounds__(256) void k_restrict_unpack2d(int n, const int32_t *__restrict__ desc, const int64_t *__restrict__ off,
                                                           const double *__restrict__ buf, double *__restrict__ coarse)
{
	const int     nn = n * n, h = n / 2;
	const int     pc = desc[2 * blockIdx.x], o = desc[2 * blockIdx.x + 1];
	const double *b  = buf + off[blockIdx.x];
	double       *cp = coarse + (size_t) pc * nn;
	if (o < 0) {
		for (int i = threadIdx.x; i < nn; i += blockDim.x) cp[i] = b[i];
	} else {
		const int bx = (o & 1) ? h : 0, by = (o & 2) ? h : 0;
		for (int i = threadIdx.x; i < h * h; i += blockDim.x) cp[bx + i % h + n * (by + i / h)] = b[i];
	}
}
// opts.fuse = 3, post-smoothing: out = S(v + P(coarse), f) with v = S(0, f) recomputed in LDS (its neighbours' edges
// come from e4): read f, 1/4 coarse, edges; write u -- 18.5 B per site instead of 26, and the pre-sweep kernel writes
// no u at all. Same arithmetic as k_rbgs_zero_resid2d_lds followed by k_rbgs2d_lds<false, true>: bit-identical.
template <int NC, int TPB = 256>
__global__ __launch_bounds__(TPB) void k_rbgs_resweep_prolong2d_lds(Level2D L, const double *__restrict__ f, const double *__restrict__ e4,
                                                                    double *__restrict__ out, Prolong2D ps)
{
	extern __shared__ __attribute__((aligned(16))) double tile2d[];
	const int     n = NC ? NC : L.n, nn = n * n;
	const int     p = blockIdx.x, tid = threadIdx.x;
	const double *fp = f + (size_t) p * nn;
	const Tile2D  T(tile2d, n);
	double       *idg = tile2d + 2 * T.cs;
	const double  rhx = L.rh2[p * 3], rhy = L.rh2[p * 3 + 1];
	F2D           fr;
	double        cr[F2D_MAX]; // the coarse correction of each pair, requested before the recompute needs the memory pipeline
	double        rv[F2D_MAX], bv[F2D_MAX], mid[2];
	loadF2d<TPB>(fp, nn, fr);
#pragma unroll
	for (int k = 0; k < F2D_MAX; k++) {
		const int i = tid + k * TPB;
		cr[k]       = (i < nn / 2) ? coarseAt2d(ps, n, p, (2 * i) % n, (2 * i) / n) : 0.0;
	}
	// the halo ring of the second sweep: the neighbours' facing values of v + P(coarse); physical faces folded -> 0. It goes
	// into the tile right away: the zero-guess sweep masks its ghosts instead of reading them.
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const int i = tid + k * TPB;
		if (i < 4 * n) {
			const int s = i / n, t = i % n;
			double    hv = 0.0;
			const int kind = L.face_kind[p * 4 + s];
			if (kind == FACE_LOCAL) {
				const int src = L.face_src[p * 4 + s];
				hv = e4[((size_t) src * 4 + (s ^ 1)) * n + t];
				hv += coarseAt2d(ps, n, src, s == 0 ? n - 1 : (s == 1 ? 0 : t), s == 2 ? n - 1 : (s == 3 ? 0 : t));
			} else if (kind == FACE_GHOST) { // the neighbour's rank sent the same sum (k_pack_faces_prolong2d)
				hv = L.ghost[(size_t) L.face_src[p * 4 + s] * n + t];
			}
			T.at(s == 0 ? -1 : (s == 1 ? n : t), s == 2 ? -1 : (s == 3 ? n : t)) = hv;
		}
	}
	idiagSetup2d<NC, TPB>(L, p, rhx, rhy, idg, mid, n);
	zeroSweep2d<NC, TPB, true>(T, idg, mid, fr, cr, n, rhx, rhy, rv, bv); // the black plane now holds v + P(coarse)
	lastSweep2d<NC, TPB>(T, idg, mid, fr, n, rhx, rhy, rv, bv);
	storePairs2d<NC, TPB>(out + (size_t) p * nn, rv, bv, n);
}

// ghost slots of coarse/fine faces: 2*gamma - m, weights of BilinearInterpolator.cpp:76-115.
// desc[8] = {patch, side, kind (2 = my neighbour is coarser, 3 = finer), half of the coarse face, nbr0, nbr1, -, -}
static __global__ void k_cf_ghost2d(int n, const int32_t *__restrict__ desc, const int32_t *__restrict__ slots,
                             const double *__restrict__ u, double *__restrict__ ghost)
{
	const int32_t *d = desc + (size_t) blockIdx.x * 8;
	const int      p = d[0], s = d[1], kind = d[2], q = d[3];
	const int      ax = s >> 1;
	const int      sa = (ax == 0) ? n : 1, sn = (ax == 0) ? 1 : n;
	const int      mine = (s & 1) ? (n - 1) * sn : 0, oth = (s & 1) ? 0 : (n - 1) * sn;
	double        *g  = ghost + (size_t) slots[blockIdx.x] * n;
	const double  *up = u + (size_t) p * n * n;
	for (int a = threadIdx.x; a < n; a += blockDim.x) {
		const double m = up[mine + a * sa];
		double       gamma;
		if (kind == 2) {
			const double other = up[mine + (a ^ 1) * sa];
			const int    ca    = (a + (q ? n : 0)) / 2;
			const double C     = d[4] >= 0 ? u[(size_t) d[4] * n * n + oth + ca * sa] : ghost[(size_t) (-(d[4] + 2)) * n + ca];
			gamma              = 5.0 / 6 * m - 1.0 / 6 * other + 2.0 / 6 * C;
		} else {
			const int    qa = (a >= n / 2), nbq = d[4 + qa];
			const int    fa = 2 * (a - qa * (n / 2));
			const double f0 = nbq >= 0 ? u[(size_t) nbq * n * n + oth + fa * sa] : ghost[(size_t) (-(nbq + 2)) * n + fa];
			const double f1 = nbq >= 0 ? u[(size_t) nbq * n * n + oth + (fa + 1) * sa] : ghost[(size_t) (-(nbq + 2)) * n + fa + 1];
			gamma           = 1.0 / 3 * m + (1.0 / 3 * f0 + 1.0 / 3 * f1);
		}
		g[a] = 2 * gamma - m;
	}
}

static __global__ void k_pack_faces2d(int n, const int32_t *__restrict__ faces, const double *__restrict__ u,
                               double *__restrict__ sendbuf)
{
	const int     p = faces[2 * blockIdx.x], s = faces[2 * blockIdx.x + 1];
	const int     ax = s >> 1, sa = (ax == 0) ? n : 1, sn = (ax == 0) ? 1 : n;
	const double *up = u + (size_t) p * n * n + ((s & 1) ? (n - 1) * sn : 0);
	double       *o  = sendbuf + (size_t) blockIdx.x * n;
	for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = up[i * sa];
}

// the facing values of u + P(coarse) -- u stored, or (e4 != null) only its edge layers -- for neighbours on other ranks: the
// sums a local neighbour would form itself (k_rbgs2d_lds<false, true>, k_rbgs_resweep_prolong2d_lds), same operands, same order
static __global__ void k_pack_faces_prolong2d(int n, const int32_t *__restrict__ faces, const double *__restrict__ u, const double *__restrict__ e4,
                                       Prolong2D ps, double *__restrict__ sendbuf)
{
	const int p = faces[2 * blockIdx.x], s = faces[2 * blockIdx.x + 1];
	for (int i = threadIdx.x; i < n; i += blockDim.x) {
		const int x = s == 0 ? 0 : (s == 1 ? n - 1 : i), y = s == 2 ? 0 : (s == 3 ? n - 1 : i);
		double    v = e4 ? e4[((size_t) p * 4 + s) * n + i] : u[(size_t) p * n * n + x + n * y];
		v += coarseAt2d(ps, n, p, x, y);
		sendbuf[(size_t) blockIdx.x * n + i] = v;
	}
}

__device__ __forceinline__ double restrictCell2d(const double *fp, int n, int hx, int hy)
{
	double acc = 0.0; // AvgRstr.h:95-102 order: x then y, each /(1<<D)
	acc += fp[2 * hx + n * (2 * hy)] / 4;
	acc += fp[2 * hx + 1 + n * (2 * hy)] / 4;
	acc += fp[2 * hx + n * (2 * hy + 1)] / 4;
	acc += fp[2 * hx + 1 + n * (2 * hy + 1)] / 4;
	return acc;
}
static __global__ __launch_bounds__(256) void k_restrict2d(int n, int Pc, const int32_t *__restrict__ child,
                                                    const int32_t *__restrict__ copy, const double *__restrict__ fine,
                                                    const double *__restrict__ remote, const int64_t *__restrict__ remote_off,
                                                    double *__restrict__ coarse)
{
	const int    nn = n * n, h = n / 2;
	const size_t total = (size_t) Pc * nn;
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t) gridDim.x * blockDim.x) {
		const int pc = (int) (idx / nn), c = (int) (idx % nn), x = c % n, y = c / n;
		if (copy[pc]) {
			const int src = child[(size_t) pc * 4];
			coarse[idx]   = 0.0 + (src >= 0 ? fine[(size_t) src * nn + c] : remote[remote_off[-(src + 2)] + c]);
			continue;
		}
		const int ox = x >= h, oy = y >= h, hx = x - ox * h, hy = y - oy * h;
		const int src = child[(size_t) pc * 4 + ox + 2 * oy];
		coarse[idx]   = (src >= 0) ? restrictCell2d(fine + (size_t) src * nn, n, hx, hy) : remote[remote_off[-(src + 2)] + hx + h * hy];
	}
}
static __global__ __launch_bounds__(256) void k_restrict_pack2d(int n, const int32_t *__restrict__ desc, const int64_t *__restrict__ off,
                                                         const double *__restrict__ fine, double *__restrict__ buf)
{
	const int     nn = n * n, h = n / 2;
	const int     p = desc[2 * blockIdx.x], o = desc[2 * blockIdx.x + 1];
	const double *fp = fine + (size_t) p * nn;
	double       *b  = buf + off[blockIdx.x];
	if (o < 0)
		for (int i = threadIdx.x; i < nn; i += blockDim.x) b[i] = fp[i];
	else
		for (int i = threadIdx.x; i < h * h; i += blockDim.x) b[i] = restrictCell2d(fp, n, i % h, i / h);
}
static __global__ __launch_bounds__(256) void k_prolong2d(int n, int Pf, const int32_t *__restrict__ parent,
                                                   const int32_t *__restrict__ orth, const double *__restrict__ coarse,
                                                   const double *__restrict__ remote, const int64_t *__restrict__ remote_off,
                                                   double *__restrict__ fine)
{
	const int    nn = n * n, h = n / 2;
	const size_t total = (size_t) Pf * nn;
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t) gridDim.x * blockDim.x) {
		const int pf = (int) (idx / nn), c = (int) (idx % nn), x = c % n, y = c / n;
		const int o = orth[pf], pa = parent[pf];
		double    cv;
		if (o >= 0) {
			if (pa >= 0)
				cv = coarse[(size_t) pa * nn + (x + ((o & 1) ? n : 0)) / 2 + n * ((y + ((o & 2) ? n : 0)) / 2)];
			else
				cv = remote[remote_off[-(pa + 2)] + x / 2 + h * (y / 2)];
		} else {
			cv = (pa >= 0) ? coarse[(size_t) pa * nn + c] : remote[remote_off[-(pa + 2)] + c];
		}
		fine[idx] += cv;
	}
}
static __global__ __launch_bounds__(256) void k_prolong_pack2d(int n, const int32_t *__restrict__ desc, const int64_t *__restrict__ off,
                                                        const double *__restrict__ coarse, double *__restrict__ buf)
{
	const int     nn = n * n, h = n / 2;
	const int     pc = desc[2 * blockIdx.x], o = desc[2 * blockIdx.x + 1];
	const double *cp = coarse + (size_t) pc * nn;
	double       *b  = buf + off[blockIdx.x];
	if (o < 0) {
		for (int i = threadIdx.x; i < nn; i += blockDim.x) b[i] = cp[i];
	} else {
		const int bx = (o & 1) ? h : 0, by = (o & 2) ? h : 0;
		for (int i = threadIdx.x; i < h * h; i += blockDim.x) b[i] = cp[bx + i % h + n * (by + i / h)];
	}
}

// ---- reference block-Jacobi patch solve in 2D (FftwPatchSolver<2>): rhs, then 4 dense transform passes
static __global__ __launch_bounds__(256) void k_patch_rhs2d(Level2D L, const double *__restrict__ u, const double *__restrict__ f,
                                                     double *__restrict__ rhs)
{
	const int    n = L.n, nn = n * n;
	const size_t total = (size_t) L.P * nn;
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t) gridDim.x * blockDim.x) {
		const int p = (int) (idx / nn), c = (int) (idx % nn), x = c % n, y = c / n;
		double    v = f[idx];
		const double m = u[idx];
		const int    xy[2] = {x, y};
#pragma unroll
		for (int ax = 0; ax < 2; ax++)
#pragma unroll
			for (int side = 0; side < 2; side++) {
				if (xy[ax] != (side ? n - 1 : 0)) continue;
				const int s = 2 * ax + side;
				if (L.face_kind[p * 4 + s] < FACE_LOCAL) continue;
				const double gh = ghost2d(L, u, p, s, xy[1 - ax], m, false);
				v -= 2.0 * L.rh2[p * 3 + ax] * (0.5 * m + 0.5 * gh);
			}
		rhs[idx] = v;
	}
}
// ---- the reference's OTHER patch solver in 2D: PatchSolvers/BiCGStabSolver.h:114-132 (apps/2d/steady.cpp:326-327, --patch_solver bcgs)
// Per patch: the right-hand side with its interface terms (addInterfaceToRHS: `rhs`, formed by k_patch_rhs2d from the old iterate),
// then BiCGStab<2>::solve (BiCGStab.h:45-106, no preconditioner, statement for statement) on the ONE patch with
// StarPatchOp<2>::apply (StarPatchOp.h:204-319: faces with a neighbour closed as homogeneous Dirichlet) from the patch's current
// values as the initial guess, to `tol` on ||resid|| / ||resid_0|| or `max_it` iterations. One workgroup of 256 threads owns a patch
// for the whole solve. A thread owns CPT consecutive cells of one row (ceil(n / CPT) segments per row, n rows: <= 256 threads
// for n <= 64 with CPT = 16, n <= 32 with 4, n <= 16 with 1): the seven Krylov vectors live in its registers, an operator
// application takes the x-neighbours from registers and only the rows above / below plus two segment-end cells from an LDS tile
// (segments CPT + 2 doubles apart: 128-bit accesses, and the 16 lanes of one LDS pass fall into 16 different bank groups), the
// closure of the patch operator on its four sides (ghost = -cell, or +cell on a Neumann side) is formed in registers, dot
// products are block reductions in a fixed order (a thread's cells, waveSum64, four partial sums through LDS). An iteration
// costs five barriers. u is updated in place (its neighbours' old values are in `rhs` already). its[p] = iterations taken.
// FULL: n is a multiple of CPT (every size the drivers use): no partly filled segments.
// the sum of v over the 64 lanes of a wave, the same value in every lane: butterflies inside each row of 16 lanes, then the rows
// combined (row 1 += row 0, row 3 += row 2; rows 2, 3 += lane 31) and lane 63 read back -- data-parallel-primitive moves on the
// vector ALU in a fixed order. (__shfl_down compiles to ds_bpermute_b32: 24 LDS-pipe instructions per pair of sums; the solve
// below waited for the LDS pipe 30 % of its time with them, SQ_WAIT_INST_LDS.)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dppMove0(double v)
{
	int lo = __double2loint(v), hi = __double2hiint(v);
	lo     = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true); // (rows outside ROW_MASK, lanes without a source: 0)
	hi     = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double waveSum64(double v)
{
	v += dppMove0<0xB1, 0xF>(v);  // quad_perm [1, 0, 3, 2]
	v += dppMove0<0x4E, 0xF>(v);  // quad_perm [2, 3, 0, 1]
	v += dppMove0<0x141, 0xF>(v); // row_half_mirror
	v += dppMove0<0x140, 0xF>(v); // row_mirror: every lane holds its row's sum
	v += dppMove0<0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
	v += dppMove0<0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
	return __hiloint2double(hi, lo);
}

template <int CPT, bool FULL>
__global__ __launch_bounds__(256, 2) void k_patch_bcgs2d(Level2D L, const double *__restrict__ rhs, double *__restrict__ u, double tol, int max_it,
                                                      int32_t *__restrict__ its)
{
	extern __shared__ __attribute__((aligned(16))) double tile2d[]; // n rows of spr segments of SS doubles
	constexpr int SS = CPT > 1 ? CPT + 2 : 1;
	const int     n = L.n, p = blockIdx.x, tid = threadIdx.x;
	const int     spr = (n + CPT - 1) / CPT, lw = spr * SS, row = tid / spr, seg = tid - row * spr, x0 = seg * CPT;
	const bool    act = row < n;
	const int     nv  = !act ? 0 : (FULL ? CPT : min(CPT, n - x0)); // valid cells of the segment (cells beyond hold zeros throughout)
	const bool    last_seg = seg == spr - 1;
	const double  rhx = L.rh2[p * 3], rhy = L.rh2[p * 3 + 1];
	double        sg[4];
#pragma unroll
	for (int s2 = 0; s2 < 4; s2++) sg[s2] = (L.face_kind[p * 4 + s2] == FACE_NEUMANN) ? 1.0 : -1.0;
	double       *mine = tile2d + row * lw + seg * SS;
	double        x[CPT], r[CPT], pv[CPT], ap[CPT], sv[CPT], as[CPT];
	// rhat never changes after the start: with 16 cells per thread it lives in LDS behind the tile and is streamed into the two dot
	// products that read it (32 registers less: two workgroups fit a CU, each hiding the other's latencies). Pair k / 2 of thread t
	// sits at 16-byte slot (k / 2) * 256 + t: the lanes of an access read consecutive slots. (A thread's sixteen values side by
	// side -- 128 bytes from lane to lane -- put eight lanes of every 16-lane group on one slot: SQ_LDS_BANK_CONFLICT was 54 % of
	// the kernel's LDS cycles.)
	constexpr bool RH_LDS = CPT >= 16;
	double         rh_reg[RH_LDS ? 1 : CPT];
	double2       *rh_lds = reinterpret_cast<double2 *>(tile2d + n * lw) + tid;
	auto           loadRh = [&](double(&o)[CPT]) {
        if (RH_LDS) {
#pragma unroll
            for (int k = 0; k < CPT; k += 2) {
                const double2 t = rh_lds[(k / 2) * 256];
                o[k] = t.x, o[k + (CPT > 1)] = t.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < CPT; k++) o[k] = rh_reg[RH_LDS ? 0 : k];
        }
	};
	double       *up = u + (size_t) p * n * n + (size_t) row * n + x0;
	const double *bp = rhs + (size_t) p * n * n + (size_t) row * n + x0;
	// out = A_patch v. (The tile's previous readers are done: a dot product's barrier lies between any two applications.)
	auto apply = [&](const double(&v)[CPT], double(&out)[CPT]) {
		if (act) {
			if (CPT > 1) {
#pragma unroll
				for (int k = 0; k < CPT; k += 2) *reinterpret_cast<double2 *>(mine + k) = make_double2(v[k], v[k + (CPT > 1)]);
			} else
				mine[0] = v[0];
		}
		__syncthreads();
		if (act) {
			double ym[CPT], yp[CPT];
			if (row > 0) {
				if (CPT > 1) {
#pragma unroll
					for (int k = 0; k < CPT; k += 2) {
						const double2 t = *reinterpret_cast<const double2 *>(mine - lw + k);
						ym[k] = t.x, ym[k + (CPT > 1)] = t.y;
					}
				} else
					ym[0] = mine[-lw];
			} else {
#pragma unroll
				for (int k = 0; k < CPT; k++) ym[k] = sg[2] * v[k];
			}
			if (row < n - 1) {
				if (CPT > 1) {
#pragma unroll
					for (int k = 0; k < CPT; k += 2) {
						const double2 t = *reinterpret_cast<const double2 *>(mine + lw + k);
						yp[k] = t.x, yp[k + (CPT > 1)] = t.y;
					}
				} else
					yp[0] = mine[lw];
			} else {
#pragma unroll
				for (int k = 0; k < CPT; k++) yp[k] = sg[3] * v[k];
			}
			const double left  = seg > 0 ? mine[-SS + CPT - 1] : sg[0] * v[0];
			const double right = last_seg ? sg[1] * v[CPT - 1] : mine[SS]; // (FULL, or a full segment: its last cell is cell CPT - 1)
#pragma unroll
			for (int k = 0; k < CPT; k++) {
				const double l = k == 0 ? left : v[k > 0 ? k - 1 : 0];
				double       rt = k == CPT - 1 ? right : v[k < CPT - 1 ? k + 1 : 0];
				if (!FULL && k + 1 == nv) rt = sg[1] * v[k]; // the patch's east side inside a partly filled (hence last) segment
				out[k] = lap2d(l, v[k], rt, ym[k], yp[k], rhx, rhy);
				if (!FULL && k >= nv) out[k] = 0.0;
			}
		} else {
#pragma unroll
			for (int k = 0; k < CPT; k++) out[k] = 0.0;
		}
	};
	// (sum a b, sum c d) over the patch, the same value in every thread; partial sums alternate between two LDS slots, so one
	// barrier per call separates a slot's writers from the readers of its previous use
	__shared__ double red[2][2][4];
	int               par = 0;
	auto dot2 = [&](const double(&a)[CPT], const double(&b)[CPT], const double(&c2)[CPT], const double(&d)[CPT], double &o0, double &o1) {
		double s0 = 0.0, s1 = 0.0;
#pragma unroll
		for (int k = 0; k < CPT; k++) {
			s0 += a[k] * b[k];
			s1 += c2[k] * d[k];
		}
		s0 = waveSum64(s0), s1 = waveSum64(s1);
		if ((tid & 63) == 0) red[par][0][tid >> 6] = s0, red[par][1][tid >> 6] = s1;
		__syncthreads();
		o0 = ((red[par][0][0] + red[par][0][1]) + red[par][0][2]) + red[par][0][3];
		o1 = ((red[par][1][0] + red[par][1][1]) + red[par][1][2]) + red[par][1][3];
		par ^= 1;
	};
#pragma unroll
	for (int k = 0; k < CPT; k++) x[k] = k < nv ? up[k] : 0.0;
	apply(x, r); // A->apply(x, resid)
#pragma unroll
	for (int k = 0; k < CPT; k++) {
		r[k]  = k < nv ? -1.0 * r[k] + bp[k] : 0.0; // resid->scaleThenAdd(-1, b)
		pv[k] = r[k];                               // rhat->copy(resid); p->copy(resid)
		if (!RH_LDS) rh_reg[RH_LDS ? 0 : k] = r[k];
	}
	if (RH_LDS) {
#pragma unroll
		for (int k = 0; k < CPT; k += 2) rh_lds[(k / 2) * 256] = make_double2(r[k], r[k + (CPT > 1)]);
	}
	double rr, rho;
	dot2(r, r, r, r, rr, rho); // (rhat == resid here: rho = rhat . resid)
	rho = rr;
	const double r0_norm = sqrt(rr);
	int          num     = 0;
	while (sqrt(rr) / r0_norm > tol && num < max_it) {
		apply(pv, ap);
		double d0, d1;
		{
			double rh[CPT];
			loadRh(rh);
			dot2(rh, ap, rh, ap, d0, d1);
		}
		const double alpha = rho / d0;
#pragma unroll
		for (int k = 0; k < CPT; k++) sv[k] = r[k] + -alpha * ap[k]; // s->copy(resid); s->addScaled(-alpha, ap)
		apply(sv, as);
		dot2(as, sv, as, as, d0, d1);
		const double omega = d0 / d1;
#pragma unroll
		for (int k = 0; k < CPT; k++) {
			x[k] = x[k] + (alpha * pv[k] + omega * sv[k]);   // x->addScaled(alpha, p, omega, s)
			r[k] = r[k] + (-alpha * ap[k] + -omega * as[k]); // resid->addScaled(-alpha, ap, -omega, as)
		}
		double rho_new;
		{
			double rh[CPT];
			loadRh(rh);
			dot2(r, rh, r, r, rho_new, rr);
		}
		const double beta = rho_new * alpha / (rho * omega);
#pragma unroll
		for (int k = 0; k < CPT; k++) {
			pv[k] = pv[k] + -omega * ap[k]; // p->addScaled(-omega, ap)
			pv[k] = beta * pv[k] + r[k];    // p->scaleThenAdd(beta, resid)
		}
		num++;
		rho = rho_new;
	}
#pragma unroll
	for (int k = 0; k < CPT; k++)
		if (k < nv) up[k] = x[k];
	if (tid == 0 && its) its[p] = num;
}

// STAGE 0,1 forward x,y; 2,3 inverse x,y. mats: [nplans][4][n*n] row-major; lam: [nplans][2][n]
template <int STAGE>
__global__ __launch_bounds__(256) void k_dst_axis2d(int n, int P, const int32_t *__restrict__ plan,
                                                    const double *__restrict__ mats, const double *__restrict__ lam,
                                                    const int32_t *__restrict__ zero_mode, const double *__restrict__ rh2,
                                                    const double *__restrict__ in, double *__restrict__ out)
{
	constexpr int AX = STAGE % 2;
	const int     nn = n * n, st = AX ? n : 1;
	const size_t  total = (size_t) P * nn;
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t) gridDim.x * blockDim.x) {
		const int     p = (int) (idx / nn), c = (int) (idx % nn);
		const int     pl = plan[p];
		const double *M  = mats + ((size_t) pl * 4 + STAGE) * nn;
		const int     i = (c / st) % n, base = c - i * st;
		const double *ip = in + (size_t) p * nn + base;
		double        acc = 0.0;
		for (int j = 0; j < n; j++) acc += M[(size_t) i * n + j] * ip[(size_t) j * st];
		if (STAGE == 1) {
			const double *lm = lam + (size_t) pl * 2 * n;
			acc /= -(lm[c % n] * rh2[p * 3] + lm[n + c / n] * rh2[p * 3 + 1]);
			if (zero_mode[pl] && c == 0) acc = 0.0;
		}
		if (STAGE == 3) acc *= 4.0 / ((double) n * n);
		out[idx] = acc;
	}
}

// The same pass on the fp64 matrix cores for patches too large for LDS (n a multiple of 16; config C1: one 256^2 patch): a wave
// per 16x16 tile of the output, K = n in steps of four, both operands straight from global memory (the patch and the matrix
// stay in L2). x passes: Out = In M^T, y passes: Out = M In. Sums in the MFMA's order: equal to k_dst_axis2d to rounding.
template <int STAGE>
__global__ __launch_bounds__(256) void k_dst_axis2d_mfma(int n, int P, const int32_t *__restrict__ plan,
                                                         const double *__restrict__ mats, const double *__restrict__ lam,
                                                         const int32_t *__restrict__ zero_mode, const double *__restrict__ rh2,
                                                         const double *__restrict__ in, double *__restrict__ out)
{
	constexpr int AX = STAGE % 2;
	const int     nn = n * n, tn = n / 16, tp = tn * tn;
	const int     t  = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (t >= P * tp) return;
	const int     p = t / tp, tr = (t % tp) / tn, tc = (t % tp) % tn; // tile row / column of the output
	const int     l = threadIdx.x & 63, j = l & 15, g = l >> 4;
	const int     pl = plan[p];
	const double *M  = mats + ((size_t) pl * 4 + STAGE) * nn;
	const double *ip = in + (size_t) p * nn;
	// AX = 0: A[i = row y][k] = In[y][k], B[k][col i] = M[i][k];   AX = 1: A[i = row][k] = M[row][k], B[k][col x] = In[k][x]
	const double *ap = AX == 0 ? ip + (size_t) (16 * tr + j) * n + g : M + (size_t) (16 * tr + j) * n + g;
	const double *bp = AX == 0 ? M + (size_t) (16 * tc + j) * n + g : ip + (size_t) g * n + 16 * tc + j;
	const size_t  bstep = AX == 0 ? 4 : (size_t) 4 * n;
	typedef double v4 __attribute__((ext_vector_type(4)));
	v4 d = v4{0, 0, 0, 0};
	for (int k0 = 0; k0 < n / 4; k0 += 4) { // (n is a multiple of 16) four steps' operands in flight together
		double a[4], b[4];
#pragma unroll
		for (int q = 0; q < 4; q++) a[q] = ap[4 * (k0 + q)], b[q] = bp[(k0 + q) * bstep];
#pragma unroll
		for (int q = 0; q < 4; q++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], d, 0, 0, 0);
	}
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const int row = 16 * tr + g + 4 * r, col = 16 * tc + j;
		double    v   = d[r];
		if (STAGE == 1) {
			const double *lm = lam + (size_t) pl * 2 * n;
			v /= -(lm[col] * rh2[p * 3] + lm[n + row] * rh2[p * 3 + 1]);
			if (zero_mode[pl] && row == 0 && col == 0) v = 0.0;
		}
		if (STAGE == 3) v *= 4.0 / ((double) n * n);
		out[(size_t) p * nn + (size_t) row * n + col] = v;
	}
}

// The same exact patch solve for patches that fit in LDS twice (n <= 64), ONE launch: right-hand side with the interface
// terms, the four dense transform passes and the eigenvalue division on two LDS tiles, one workgroup per patch -- 16 B per
// site instead of 24 + 4 x 16, and one kernel's latency instead of five on the coarsest level of a cycle (a single patch:
// 78 us for five launches). Sums in the order of k_dst_axis2d. matsT: the same matrices transposed (x passes: lanes run
// along the output index, so the matrix element must be contiguous across lanes). ZERO: the iterate is zero (no interface
// term, u never read). Out of place: block Jacobi reads the neighbours' OLD face values, and another workgroup may be done
// with its patch before this one starts.
// TPB: threads per workgroup (256, or 1024 on levels with few patches: a single patch is a serial chain of four passes, and
// sixteen waves shorten each of them four times)
template <bool ZERO, int NC, int TPB = 256>
__global__ __launch_bounds__(TPB) void k_patch_solve2d_lds(Level2D L, const int32_t *__restrict__ plan, const double *__restrict__ mats,
                                                           const double *__restrict__ matsT, const double *__restrict__ lam,
                                                           const int32_t *__restrict__ zero_mode, const double *__restrict__ f,
                                                           const double *__restrict__ u, double *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) double tile2d[]; // two tiles of n*n
	const int     n = NC ? NC : L.n, nn = n * n;
	const int     p = blockIdx.x, tid = threadIdx.x, pl = plan[p];
	double       *A = tile2d, *B = tile2d + nn;
	const double  rhx = L.rh2[p * 3], rhy = L.rh2[p * 3 + 1];
	const double *fp = f + (size_t) p * nn, *up = u + (size_t) p * nn;
	for (int c = tid; c < nn; c += TPB) { // k_patch_rhs2d
		double v = fp[c];
		if (!ZERO) {
			const int    x = c % n, y = c / n, xy[2] = {x, y};
			const double m = up[c];
#pragma unroll
			for (int ax = 0; ax < 2; ax++)
#pragma unroll
				for (int side = 0; side < 2; side++) {
					if (xy[ax] != (side ? n - 1 : 0)) continue;
					const int s = 2 * ax + side;
					if (L.face_kind[p * 4 + s] < FACE_LOCAL) continue;
					const double gh = ghost2d(L, u, p, s, xy[1 - ax], m, false);
					v -= 2.0 * L.rh2[p * 3 + ax] * (0.5 * m + 0.5 * gh);
				}
		}
		A[c] = v;
	}
	__syncthreads();
	const double *lm = lam + (size_t) pl * 2 * n;
	if (NC == 64) {
		// lane l = one matrix row index, wave w: 16 outputs per thread whose sums advance together. Per j: ONE coalesced
		// matrix load (the transposed matrix for both kinds of pass: lanes run along the OUTPUT index) and 16 LDS
		// broadcast reads; sums in j order as k_dst_axis2d.
		constexpr int WV = TPB / 64, KPT = 64 / WV; // waves, outputs per thread
		const int     l = tid & 63, w = tid >> 6;
#pragma unroll 1
		for (int stage = 0; stage < 4; stage++) {
			const int     ax = stage & 1;
			const double *MT = matsT + ((size_t) pl * 4 + stage) * nn;
			const double *in = (stage & 1) ? B : A;
			double       *o  = (stage & 1) ? A : B;
			double        acc[KPT];
#pragma unroll
			for (int k = 0; k < KPT; k++) acc[k] = 0.0;
			// x pass: output (i = l, y = w + WV k) = sum_j M[l][j] in[y][j];  y pass: output (i = l, x = w + WV k) = sum_j M[l][j] in[j][x]
			const int bs = ax ? 1 : 64, js = ax ? 64 : 1; // input element of output k at step j: in[(w + WV k) * bs + j * js]
			// this lane's matrix column in chunks of eight, the next chunk requested while the current one is used (with one
			// workgroup on the chip -- the coarsest level -- a load per step would cost a memory round trip per step)
			double ma[8], mb[8];
#pragma unroll
			for (int jj = 0; jj < 8; jj++) ma[jj] = MT[jj * 64 + l];
#pragma unroll 1
			for (int jc = 0; jc < 64; jc += 8) {
				const int jn = (jc + 8 < 64) ? jc + 8 : jc;
#pragma unroll
				for (int jj = 0; jj < 8; jj++) mb[jj] = MT[(jn + jj) * 64 + l];
#pragma unroll
				for (int jj = 0; jj < 8; jj++) {
#pragma unroll
					for (int k = 0; k < KPT; k++) acc[k] += ma[jj] * in[(w + WV * k) * bs + (jc + jj) * js];
				}
#pragma unroll
				for (int jj = 0; jj < 8; jj++) ma[jj] = mb[jj];
			}
#pragma unroll
			for (int k = 0; k < KPT; k++) {
				const int c = ax ? l * 64 + (w + WV * k) : (w + WV * k) * 64 + l; // cell x + n y of the output
				double    v = acc[k];
				if (stage == 1) {
					v /= -(lm[c % 64] * rhx + lm[64 + c / 64] * rhy);
					if (zero_mode[pl] && c == 0) v = 0.0;
				}
				if (stage == 3)
					out[(size_t) p * nn + c] = v * (4.0 / ((double) n * n));
				else
					o[c] = v;
			}
			__syncthreads();
		}
		return;
	}
#pragma unroll 1
	for (int stage = 0; stage < 4; stage++) { // any n: the plain form
		const int     ax = stage & 1, st = ax ? n : 1;
		const double *M  = (ax ? mats : matsT) + ((size_t) pl * 4 + stage) * nn;
		const double *in = (stage & 1) ? B : A;
		double       *o  = (stage & 1) ? A : B;
		for (int c = tid; c < nn; c += TPB) {
			const int i = (c / st) % n, base = c - i * st;
			double    acc = 0.0;
			if (ax) {
				for (int j = 0; j < n; j++) acc += M[(size_t) i * n + j] * in[base + j * st];
			} else {
				for (int j = 0; j < n; j++) acc += M[(size_t) j * n + i] * in[base + j];
			}
			if (stage == 1) {
				acc /= -(lm[c % n] * rhx + lm[n + c / n] * rhy);
				if (zero_mode[pl] && c == 0) acc = 0.0;
			}
			if (stage == 3)
				out[(size_t) p * nn + c] = acc * (4.0 / ((double) n * n));
			else
				o[c] = acc;
		}
		__syncthreads();
	}
}

// The right-hand side of a 64^2 patch's solve in an LDS tile [y][x] (row stride PS2D_LD): f, minus the interface terms of the old
// iterate on the edge cells (k_patch_rhs2d's expressions in its order: a corner cell takes its x-face term, then its y-face
// term). 256 threads: all of them bring f in; then thread (s, t) = (tid >> 6, tid & 63) forms the term of position t on side s,
// so the chain of dependent loads (face tables -> own and neighbour's value -> their coarse values) is one deep per workgroup.
// (With a thread owning cells tid + 256 k, the 62 cells of a west or east column belonged to four threads: sixteen chains in a
// row on each, 100 us of the 257 us post-sweep at 4096 patches.) PROLONG: the old iterate is u + P(coarse), see below.
// No barrier at the end: the caller's.
constexpr int PS2D_LD = 65; // padded row length of the LDS tiles
template <bool ZERO, bool PROLONG>
__device__ __forceinline__ void patchRhsTile2d(const Level2D &L, int p, const double *__restrict__ fp, const double *__restrict__ up,
                                               const double *__restrict__ u, const Prolong2D &ps, double *T, int tid)
{
	constexpr int n = 64, nn = n * n, LD = PS2D_LD;
	for (int c = tid; c < nn; c += 256) T[(c / n) * LD + c % n] = fp[c];
	if (ZERO) return;
	const int  s = tid >> 6, t = tid & 63, kind = L.face_kind[p * 4 + s];
	const int  x = s == 0 ? 0 : (s == 1 ? n - 1 : t), y = s == 2 ? 0 : (s == 3 ? n - 1 : t);
	const bool has = kind >= FACE_LOCAL;
	double     term = 0.0;
	if (has) {
		double m = up[x + n * y];
		if (PROLONG) m += coarseAt2d(ps, n, p, x, y);
		double gh = ghost2d(L, u, p, s, t, m, false);
		if (PROLONG && kind == FACE_LOCAL)
			gh += coarseAt2d(ps, n, L.face_src[p * 4 + s], s == 0 ? n - 1 : (s == 1 ? 0 : t), s == 2 ? n - 1 : (s == 3 ? 0 : t));
		term = 2.0 * L.rh2[p * 3 + (s >> 1)] * (0.5 * m + 0.5 * gh);
	}
	__syncthreads(); // f is in the tile
	if (has && s < 2) T[y * LD + x] -= term;
	__syncthreads();
	if (has && s >= 2) T[y * LD + x] -= term;
}

// The exact 2D patch solve for 64^2 patches on the fp64 matrix cores: four 64x64x64 products (x forward, y forward +
// eigenvalue division, x inverse, y inverse) as v_mfma_f64_16x16x4 tiles, D(16x16) += A(16x4) B(4x16) with lane
// (j = l & 15, g = l >> 4) holding A[i = j][k = g], B[k = g][col = j], D[row = g + 4r][col = j]. Wave w owns the output rows
// 16w .. 16w+15 of every stage (four 16x16 tiles, 16 k-steps each: 64 MFMAs per wave and stage); the data crosses between
// stages through one LDS tile with padded rows (a row per lane: stride 65 doubles, conflict-free), the matrices come
// transposed (matsT) so that sixteen lanes read 128 contiguous bytes. One workgroup per patch, out of place (block Jacobi
// reads the neighbours' OLD values). Sums run in the MFMA's order, not k_dst_axis2d's: equal to rounding.
typedef double v4f64_2d __attribute__((ext_vector_type(4)));
// PF: the matrix fragments of the next stage are fetched into registers a stage ahead (246 VGPRs, one workgroup per CU: for
// levels of few patches, where the latency of the matrix loads is all there is); without it they are read where they are used
// and two workgroups share a CU.
// PROLONG (opts.fuse >= 2, the post-sweep of a V-cycle): the old iterate is u + P(coarse) with the prolongation never carried out --
// a block-Jacobi sweep reads the old iterate only through its interface terms, i.e. on the patch's own edge cells and its
// neighbours' facing cells: both take their coarse value here (a neighbour on another rank sent the sum, k_pack_faces_prolong2d)
// and the result overwrites u. No k_prolong2d pass (17 B per site).
template <bool ZERO, bool PF, bool PROLONG = false>
__global__ __launch_bounds__(256) void k_patch_solve2d_mfma(Level2D L, const int32_t *__restrict__ plan, const double *__restrict__ matsT,
                                                            const double *__restrict__ lam, const int32_t *__restrict__ zero_mode,
                                                            const double *__restrict__ f, const double *__restrict__ u,
                                                            double *__restrict__ out, Prolong2D ps = Prolong2D(),
                                                            const int32_t *__restrict__ list = nullptr)
{
	static_assert(!(ZERO && PROLONG), "a zero iterate has no correction to take");
	constexpr int n = 64, nn = n * n, LD = PS2D_LD;
	extern __shared__ __attribute__((aligned(16))) double tile2d[]; // one tile of 64 x 65: stages 0 and 2 work on a wave's own rows in place
	double       *A = tile2d, *B = tile2d;
	const int     p = list ? list[blockIdx.x] : blockIdx.x, tid = threadIdx.x, pl = plan[p]; // (list: the patches the half-size kernel leaves)
	const int     w = tid >> 6, l = tid & 63, j = l & 15, g = l >> 4;
	const double  rhx = L.rh2[p * 3], rhy = L.rh2[p * 3 + 1];
	const double *fp = f + (size_t) p * nn, *up = u + (size_t) p * nn;
	const double *MT = matsT + (size_t) pl * 4 * nn; // MT[stage][k * 64 + i] = M[stage][i][k]
	const double *lm = lam + (size_t) pl * 2 * n;
	double        mc[PF ? 64 : 1], mr[PF ? 16 : 1]; // matrix fragments of a B-side stage (4 column tiles x 16 k-steps) and of an A-side stage
	auto          loadB = [&](int stage) {
        if constexpr (PF) {
#pragma unroll
            for (int ks = 0; ks < 16; ks++)
#pragma unroll
                for (int ct = 0; ct < 4; ct++) mc[ks * 4 + ct] = MT[stage * nn + (4 * ks + g) * n + 16 * ct + j];
        }
	};
	auto loadA = [&](int stage) {
		if constexpr (PF) {
#pragma unroll
			for (int ks = 0; ks < 16; ks++) mr[ks] = MT[stage * nn + (4 * ks + g) * n + 16 * w + j];
		}
	};
	auto fragB = [&](int stage, int ks, int ct) { return PF ? mc[ks * 4 + ct] : MT[stage * nn + (4 * ks + g) * n + 16 * ct + j]; };
	auto fragA = [&](int stage, int ks) { return PF ? mr[ks] : MT[stage * nn + (4 * ks + g) * n + 16 * w + j]; };
	loadB(0); // stage 0's matrix, in flight while the right-hand side is formed
	patchRhsTile2d<ZERO, PROLONG>(L, p, fp, up, u, ps, A, tid);
	__syncthreads();
	v4f64_2d d[4];
	// ---- stage 0: Y1[y][kx] = sum_x X[y][x] Fx[kx][x]: A operand = X rows (LDS), B operand = Fx^T (registers)
	loadA(1); // stage 1's matrix, in flight during stage 0
#pragma unroll
	for (int ct = 0; ct < 4; ct++) d[ct] = v4f64_2d{0, 0, 0, 0};
#pragma clang loop unroll_count(PF ? 16 : 4)
	for (int ks = 0; ks < 16; ks++) {
		const double a = A[(16 * w + j) * LD + 4 * ks + g];
#pragma unroll
		for (int ct = 0; ct < 4; ct++) d[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fragB(0, ks, ct), d[ct], 0, 0, 0);
	}
#pragma unroll
	for (int ct = 0; ct < 4; ct++)
#pragma unroll
		for (int r = 0; r < 4; r++) B[(16 * w + g + 4 * r) * LD + 16 * ct + j] = d[ct][r];
	loadB(2); // stage 2's matrix, in flight during stage 1
	__syncthreads();
	// ---- stage 1: Y2[ky][kx] = sum_y Fy[ky][y] Y1[y][kx], divided by the eigenvalues: A = Fy (registers), B = Y1 (LDS)
#pragma unroll
	for (int ct = 0; ct < 4; ct++) d[ct] = v4f64_2d{0, 0, 0, 0};
#pragma clang loop unroll_count(PF ? 16 : 4)
	for (int ks = 0; ks < 16; ks++) {
		const double a = fragA(1, ks);
#pragma unroll
		for (int ct = 0; ct < 4; ct++) d[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[(4 * ks + g) * LD + 16 * ct + j], d[ct], 0, 0, 0);
	}
	loadA(3);
	__syncthreads(); // every wave has read all rows of Y1 before any overwrites its own
#pragma unroll
	for (int ct = 0; ct < 4; ct++)
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int ky = 16 * w + g + 4 * r, kx = 16 * ct + j;
			double    v  = d[ct][r] / -(lm[kx] * rhx + lm[n + ky] * rhy);
			if (zero_mode[pl] && kx == 0 && ky == 0) v = 0.0;
			A[ky * LD + kx] = v;
		}
	__syncthreads();
	// ---- stage 2: Y3[ky][x] = sum_kx Y2[ky][kx] Gx[x][kx]: A = Y2 rows (LDS), B = Gx^T (registers)
#pragma unroll
	for (int ct = 0; ct < 4; ct++) d[ct] = v4f64_2d{0, 0, 0, 0};
#pragma clang loop unroll_count(PF ? 16 : 4)
	for (int ks = 0; ks < 16; ks++) {
		const double a = A[(16 * w + j) * LD + 4 * ks + g];
#pragma unroll
		for (int ct = 0; ct < 4; ct++) d[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fragB(2, ks, ct), d[ct], 0, 0, 0);
	}
#pragma unroll
	for (int ct = 0; ct < 4; ct++)
#pragma unroll
		for (int r = 0; r < 4; r++) B[(16 * w + g + 4 * r) * LD + 16 * ct + j] = d[ct][r];
	__syncthreads();
	// ---- stage 3: U[y][x] = sum_ky Gy[y][ky] Y3[ky][x], scaled: A = Gy (registers), B = Y3 (LDS)
#pragma unroll
	for (int ct = 0; ct < 4; ct++) d[ct] = v4f64_2d{0, 0, 0, 0};
#pragma clang loop unroll_count(PF ? 16 : 4)
	for (int ks = 0; ks < 16; ks++) {
		const double a = fragA(3, ks);
#pragma unroll
		for (int ct = 0; ct < 4; ct++) d[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[(4 * ks + g) * LD + 16 * ct + j], d[ct], 0, 0, 0);
	}
	double *op = out + (size_t) p * nn;
#pragma unroll
	for (int ct = 0; ct < 4; ct++)
#pragma unroll
		for (int r = 0; r < 4; r++) op[(16 * w + g + 4 * r) * n + 16 * ct + j] = d[ct][r] * (4.0 / ((double) n * n));
}
// ---- the same solve with HALF-SIZE transforms (the 2D twin of k_ps_sym): on an axis whose two sides close the same way
// (Dirichlet / neighbour on both, or Neumann on both: DST-II/III resp. DCT-II/III) basis function k is even about the patch
// centre for even k and odd for odd k, so a transform of length 64 is two of length 32 on the sums and differences of mirrored
// entries: forward  Y[k even] = sum_{j<32} F[k][j] (x[j] + x[63-j]),  Y[k odd] = sum_{j<32} F[k][j] (x[j] - x[63-j]);
//          inverse  x[j], x[63-j] = E[j] +- O[j],  E[j] = sum_{k even} G[j][k] y[k],  O[j] = sum_{k odd} G[j][k] y[k].
// Each of the four stages is 32 instead of 64 v_mfma_f64_16x16x4 per wave (the kernel is bound by them: 54 of 78.6 TFLOP/s on the
// full-size products); the butterflies ride on the LDS reads and on the accumulators. Transformed data sits in parity-split order
// (position c < 32: wave number 2c, c >= 32: 2(c - 32) + 1) between the stages, along both axes.
// sym [plan][stage 4][k-step 8][t 4][lane 64]: the matrix fragments in MFMA operand order (built by the host, gmg_core.hip):
//   stage 0 (B side): t = column tile;            F_x[kx(t, j)][4 ks + g]
//   stage 1 (A side): t = wave;                   F_y[ky(w, j)][4 ks + g]
//   stage 2 (B side): t = parity * 2 + column tile of x' < 32;   G_x[16 ct + j][2 (4 ks + g) + parity]
//   stage 3 (A side): t = parity * 2 + row tile of y' < 32;      G_y[16 rt + j][2 (4 ks + g) + parity]
// Only patches whose plan has two pure axes (list, n of them); the others take k_patch_solve2d_mfma. Sums run in another order
// than there: equal to rounding (<= 1e-13).
constexpr int PS2S_STAGE = 8 * 4 * 64, PS2S_PLAN = 4 * PS2S_STAGE;
template <bool ZERO, bool PF, bool PROLONG = false>
__global__ __launch_bounds__(256, PF ? 2 : 4) void k_patch_solve2d_sym(Level2D L, const int32_t *__restrict__ plan, const double *__restrict__ sym,
                                                           const double *__restrict__ inv, const int32_t *__restrict__ itab,
                                                           const double *__restrict__ f, const double *__restrict__ u,
                                                           double *__restrict__ out, const int32_t *__restrict__ list, Prolong2D ps = Prolong2D())
{
	static_assert(!(ZERO && PROLONG), "a zero iterate has no correction to take");
	constexpr int n = 64, nn = n * n, LD = PS2D_LD;
	extern __shared__ __attribute__((aligned(16))) double tile2d[]; // one tile of 64 x 65
	double       *T = tile2d;
	const int     p = list ? list[blockIdx.x] : blockIdx.x, tid = threadIdx.x, pl = plan[p];
	const int     w = tid >> 6, l = tid & 63, j = l & 15, g = l >> 4;
	const double *fp = f + (size_t) p * nn, *up = u + (size_t) p * nn;
	const double *S  = sym + (size_t) pl * PS2S_PLAN;
	// inv: per distinct (plan, spacings) of the level one table [position r'][position c'] of scale / -(lx[kx(c')] + ly[ky(r')]), the
	// reciprocal eigenvalue of the patch operator times the transforms' scale 4 / n^2 = 2^-10 (a power of two: the same bits as a
	// multiplication of the result), 0 at the zero mode of an all-Neumann patch; itab[p] = the patch's table. The IEEE divisions these
	// replace (16 per lane, ~ 15 vector instructions each, behind two dependent table loads) ran on the unit the fp64 matrix
	// instruction executes on (tools/mfma_valu.hip): a sixth of the kernel's pipe time.
	const double *ivt = inv + (size_t) itab[p] * nn;
	auto          frag = [&](int stage, int ks, int t) { return S[stage * PS2S_STAGE + (ks * 4 + t) * 64 + l]; };
	// PF: every stage's fragments are fetched a stage ahead; otherwise only those of the A-side stages (8 and 16 values per lane:
	// 124 -> 140 registers, three workgroups per CU either way), a B-side stage's 32 where it uses them
	double        mb[PF ? 32 : 1], ma[16]; // fragments of a B-side stage (8 k-steps x 4) and of an A-side stage (8, or 2 x 8)
	auto          loadB = [&](int stage) {
        if constexpr (PF) {
#pragma unroll
            for (int ks = 0; ks < 8; ks++)
#pragma unroll
                for (int t = 0; t < 4; t++) mb[ks * 4 + t] = frag(stage, ks, t);
        }
	};
	auto fB = [&](int stage, int ks, int t) { return PF ? mb[ks * 4 + t] : frag(stage, ks, t); };
	loadB(0);
	patchRhsTile2d<ZERO, PROLONG>(L, p, fp, up, u, ps, T, tid);
	__syncthreads();
	v4f64_2d d[4];
	// ---- stage 0: Y1[y][c'] = sum_{x'<32} (X[y][x'] +- X[y][63-x']) Fx[kx(c')][x']: tiles 0, 1 even kx (sums), 2, 3 odd kx (differences)
#pragma unroll
	for (int ks = 0; ks < 8; ks++) ma[ks] = frag(1, ks, w);
#pragma unroll
	for (int ct = 0; ct < 4; ct++) d[ct] = v4f64_2d{0, 0, 0, 0};
#pragma unroll // (also without PF: the eight k-steps' fragment loads are then in flight together, 586 -> 547 us per C5 cycle)
	for (int ks = 0; ks < 8; ks++) {
		const double xl = T[(16 * w + j) * LD + 4 * ks + g], xh = T[(16 * w + j) * LD + 63 - 4 * ks - g];
		const double ae = xl + xh, ao = xl - xh;
#pragma unroll
		for (int ct = 0; ct < 4; ct++) d[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(ct < 2 ? ae : ao, fB(0, ks, ct), d[ct], 0, 0, 0);
	}
#pragma unroll
	for (int ct = 0; ct < 4; ct++)
#pragma unroll
		for (int r = 0; r < 4; r++) T[(16 * w + g + 4 * r) * LD + 16 * ct + j] = d[ct][r]; // (the wave's own rows, in place)
	loadB(2);
	double iv[16]; // (requested a stage ahead of their use: they come from L2, every workgroup of the level reads the same few tables)
#pragma unroll
	for (int ct = 0; ct < 4; ct++)
#pragma unroll
		for (int r = 0; r < 4; r++) iv[ct * 4 + r] = ivt[(16 * w + g + 4 * r) * n + 16 * ct + j];
	__syncthreads();
	// ---- stage 1: Y2[r'][c'] = sum_{y'<32} Fy[ky(r')][y'] (Y1[y'][c'] +- Y1[63-y'][c']): waves 0, 1 even ky, 2, 3 odd ky
#pragma unroll
	for (int ct = 0; ct < 4; ct++) d[ct] = v4f64_2d{0, 0, 0, 0};
#pragma unroll // (also without PF: the eight k-steps' fragment loads are then in flight together, 586 -> 547 us per C5 cycle)
	for (int ks = 0; ks < 8; ks++) {
		const double a = ma[ks];
#pragma unroll
		for (int ct = 0; ct < 4; ct++) {
			const double yl = T[(4 * ks + g) * LD + 16 * ct + j], yh = T[(63 - 4 * ks - g) * LD + 16 * ct + j];
			d[ct]           = __builtin_amdgcn_mfma_f64_16x16x4f64(a, w < 2 ? yl + yh : yl - yh, d[ct], 0, 0, 0);
		}
	}
#pragma unroll
	for (int ks = 0; ks < 8; ks++) ma[ks] = frag(3, ks, w & 1), ma[8 + ks] = frag(3, ks, 2 + (w & 1));
	__syncthreads(); // every wave has read all rows of Y1 before any overwrites its own
#pragma unroll
	for (int ct = 0; ct < 4; ct++)
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int rp = 16 * w + g + 4 * r, cp = 16 * ct + j; // positions (the wave numbers behind them: the table's business)
			T[rp * LD + cp] = d[ct][r] * iv[ct * 4 + r];
		}
	__syncthreads();
	// ---- stage 2: E[r'][x'] = sum_{kx even} Y2 Gx[x'][kx], O[r'][x'] = sum_{kx odd} ...; Y3[r'][x'], Y3[r'][63-x'] = E +- O
	v4f64_2d e2[2], o2[2];
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++) e2[c2] = o2[c2] = v4f64_2d{0, 0, 0, 0};
#pragma unroll // (also without PF: the eight k-steps' fragment loads are then in flight together, 586 -> 547 us per C5 cycle)
	for (int ks = 0; ks < 8; ks++) {
		const double ae = T[(16 * w + j) * LD + 4 * ks + g], ao = T[(16 * w + j) * LD + 32 + 4 * ks + g];
#pragma unroll
		for (int c2 = 0; c2 < 2; c2++) {
			e2[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, fB(2, ks, c2), e2[c2], 0, 0, 0);
			o2[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ao, fB(2, ks, 2 + c2), o2[c2], 0, 0, 0);
		}
	}
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int rp = 16 * w + g + 4 * r, xp = 16 * c2 + j;
			T[rp * LD + xp]      = e2[c2][r] + o2[c2][r];
			T[rp * LD + 63 - xp] = e2[c2][r] - o2[c2][r];
		}
	__syncthreads();
	// ---- stage 3: E[y'][x] = sum_{ky even} Gy[y'][ky] Y3[ky][x], O likewise; U[y'][x], U[63-y'][x] = (E +- O) * scale.
	// Wave w: row tile w & 1 of y' < 32, column tiles 2 (w >> 1) and 2 (w >> 1) + 1
	const int rt = w & 1, cb = 2 * (w >> 1);
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++) e2[c2] = o2[c2] = v4f64_2d{0, 0, 0, 0};
#pragma unroll // (also without PF: the eight k-steps' fragment loads are then in flight together, 586 -> 547 us per C5 cycle)
	for (int ks = 0; ks < 8; ks++) {
		const double ae = ma[ks], ao = ma[8 + ks];
#pragma unroll
		for (int c2 = 0; c2 < 2; c2++) {
			e2[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, T[(4 * ks + g) * LD + 16 * (cb + c2) + j], e2[c2], 0, 0, 0);
			o2[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ao, T[(32 + 4 * ks + g) * LD + 16 * (cb + c2) + j], o2[c2], 0, 0, 0);
		}
	}
	double      *op = out + (size_t) p * nn;
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int yp = 16 * rt + g + 4 * r, x = 16 * (cb + c2) + j;
			op[yp * n + x]        = e2[c2][r] + o2[c2][r]; // (the scale 4 / n^2 rides in the reciprocal table)
			op[(63 - yp) * n + x] = e2[c2][r] - o2[c2][r];
		}
}
} // namespace te
