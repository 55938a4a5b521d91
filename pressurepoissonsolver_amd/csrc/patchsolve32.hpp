// Reference block-Jacobi smoother, fast path for 32^3 patches: the exact per-patch Dirichlet /
// Neumann solve of FftwPatchSolver.h:173-206 (transforms = DftPatchSolver.h:237-289 matrices) as
// three HBM passes of dense 32x32 transforms on the fp64 matrix cores.
//
// This is the one GEMM-shaped piece of the path (64 flop per site per axis, 12.6 MFLOP per patch),
// so unlike the stencils it belongs on MFMA: v_mfma_f64_16x16x4_f64, D(16x16) += A(16x4) B(4x16),
// lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D[row = (l>>4) + 4r][col = l&15], r = 0..3.
// The transform matrices live in registers as A (or B) fragments for the whole kernel; data
// fragments come straight from global memory / an LDS plane. A D tile is reused as the next
// product's B operand without moving data by letting k-step (mb, r) stand for row 16mb + g + 4r
// (the sum over k does not care about the order; the other operand is built in that order).
//
//   k_ps_xy<INV=false> : per z-plane, x then y forward transform            (8 B read + 8 B write / site)
//   k_ps_z             : z forward, eigenvalue divide, z inverse             (8 + 8)
//   k_ps_xy<INV=true>  : per z-plane, x then y inverse transform, (2/N)^3    (8 + 8)
#pragma once
#include "march3d.hpp"

namespace te
{
typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f64 mfma_f64(double a, double b, v4f64 c)
{
	return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Interface term of the patch right-hand side, StarPatchOp::addInterfaceToRHS (StarPatchOp.h:185-203):
// corr[p][s][a + N b] = (2/h^2) gamma = rh2 * (m + ghost) on faces that have a neighbour, 0 on
// physical faces. One workgroup per patch face; only face layers are touched (6/N of the sites).
// PROLONG: the iterate is u + P(coarse) (DrctIntp.h:99-106) without that sum ever being stored: a block-
// Jacobi sweep overwrites u and reads the old iterate ONLY through these interface terms, so the
// prolongation has to be evaluated on the face layers alone (levels without coarse/fine faces, see ProlongSrc).
template <int N, bool PROLONG>
__global__ __launch_bounds__(256) void k_face_corr3d(LevelDev L, const double *__restrict__ u, double *__restrict__ corr,
                                                     ProlongSrc ps)
{
	constexpr int NN = N * N, NNN = N * N * N;
	const int     p = blockIdx.x / 6, s = blockIdx.x % 6, ax = s >> 1;
	const int     kind = L.face_kind[(size_t) p * 6 + s], src = L.face_src[(size_t) p * 6 + s];
	const int     sa = (ax == 0) ? N : 1, sb = (ax == 2) ? N : NN, sn = (ax == 0) ? 1 : (ax == 1 ? N : NN);
	const int     mine = (s & 1) ? (N - 1) * sn : 0, oth = (s & 1) ? 0 : (N - 1) * sn;
	const double  rh = L.rh2[(size_t) p * 3 + ax];
	double       *c  = corr + ((size_t) p * 6 + s) * NN;
	// PROLONG: P e at a cell through coarseAtCell (octant children and patches that copy through alike); a ghost slot
	// (neighbour on another rank, coarse/fine face) holds values of u + P e already
	for (int i = threadIdx.x; i < NN; i += blockDim.x) {
		double v = 0.0;
		if (kind >= FACE_LOCAL) {
			const int a = i % N, b = i / N;
			const int cell = a * sa + b * sb;
			double    m, gh;
			if (L.f6) { // the old iterate exists only as its six face layers (k_ps_sym<false, FACES>)
				m  = L.f6[((size_t) p * 6 + s) * NN + i];
				gh = (kind == FACE_LOCAL) ? L.f6[((size_t) src * 6 + (s ^ 1)) * NN + i] : L.ghost[(size_t) src * NN + i];
			} else if (ax == 0 && L.xf) { // x faces from the compact columns the producer of u exported (LevelDev.xf): no strided reads
				m  = L.xf[((size_t) p * 2 + (s & 1)) * NN + i];
				gh = (kind == FACE_LOCAL) ? L.xf[((size_t) src * 2 + ((s & 1) ^ 1)) * NN + i] : L.ghost[(size_t) src * NN + i];
			} else {
				m  = u[(size_t) p * NNN + mine + cell];
				gh = (kind == FACE_LOCAL) ? u[(size_t) src * NNN + oth + cell] : L.ghost[(size_t) src * NN + i];
			}
			if (PROLONG) {
				m += coarseAtCell<N>(ps, p, mine + cell);
				if (kind == FACE_LOCAL) gh += coarseAtCell<N>(ps, src, oth + cell);
			}
			v = 2.0 * rh * (0.5 * m + 0.5 * gh);
		}
		c[i] = v;
	}
}

// mats: [nplans][6][32*32] row-major (forward x,y,z then inverse x,y,z); y_i = sum_j M[i*32+j] x_j
// mfrag: the same matrices in the order the three-pass kernels' lanes hold them, [nplans][6][chunk 8][lane 64][2]: a lane's 16
// values of a matrix are 8 coalesced 16-byte loads (from the row-major store they were 16 eight-byte loads touching 16 cache
// lines each: on the few-patch levels these kernels run on, fetching the 2 x 8 KiB of matrices took longer than the 64 MFMAs --
// profiles/r06_tail_stamps.txt). Element e = 2 chunk + {0, 1} of lane (j = lane & 15, g = lane >> 4):
//   matrices 0, 3 (x):       B operand  bx[nb = e >> 3][ks = e & 7]               = M[(2j + nb) * 32 + 4 ks + g]
//   matrices 1, 4, 5 (y, z): A operand  ay[mo = e >> 3][mb = (e >> 2) & 1][r = e & 3] = M[(16 mo + j) * 32 + 16 mb + g + 4 r]
//   matrix 2 (z forward):    A operand  af[mb = e >> 3][ks = e & 7]               = M[(16 mb + j) * 32 + 4 ks + g]
// (matFragIndex below is what gmg_core.hip fills the table with.) The values and the order of the products are unchanged.
__host__ __device__ inline int matFragSource(int m, int lane, int e)
{
	const int j = lane & 15, g = lane >> 4;
	if (m == 0 || m == 3) return (2 * j + (e >> 3)) * 32 + 4 * (e & 7) + g;
	if (m == 2) return (16 * (e >> 3) + j) * 32 + 4 * (e & 7) + g;
	return (16 * (e >> 3) + j) * 32 + 16 * ((e >> 2) & 1) + g + 4 * (e & 3);
}
// a lane's 16 values of matrix m of plan pl
__device__ __forceinline__ void loadMatFrag(const double *__restrict__ mfrag, int pl, int m, int lane, double (&v)[16])
{
	const double2 *F = reinterpret_cast<const double2 *>(mfrag + ((size_t) pl * 6 + m) * 1024) + lane;
#pragma unroll
	for (int c = 0; c < 8; c++) {
		const double2 t = F[c * 64];
		v[2 * c] = t.x, v[2 * c + 1] = t.y;
	}
}
// corr (forward pass only, may be null = zero initial guess, no interface term): see k_face_corr3d.
template <bool INV, bool CORR = false>
__global__ __launch_bounds__(256) void k_ps_xy(int P, const int32_t *__restrict__ plan,
                                               const double *__restrict__ mfrag, const double *__restrict__ in,
                                               const double *__restrict__ corr, double *__restrict__ out TE_STAMP_PARAM)
{
	constexpr int N = 32, NN = N * N;
	TE_STAMP_DECL;
	TE_STAMP(0, false);
	const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, g = l >> 4;
	// one workgroup per patch and segment: wave w owns planes w + 4 it, it in [it0, it0 + 8 / gridDim.y)
	// (gridDim.y > 1 spreads the few patches of a coarse level over more CUs)
	const int pid = blockIdx.x, its = 8 / gridDim.y, it0 = blockIdx.y * its;
	if (pid >= P) return;
	const int     pl = plan[pid];
	TE_STAMP(1, true);
	// matrix fragments, loaded once per wave. Output column kx = 2j + nb (a lane's two columns are
	// adjacent in memory).
	double bx[2][8];    // B = Mx^T: B[k = x = 4ks + g][col kx] = Mx[kx][x]
	double ay[2][2][4]; // A = My: A[i = ky = 16mo + j][k-step (mb, r) = y = 16mb + g + 4r]
	{
		double vx[16], vy[16];
		loadMatFrag(mfrag, pl, INV ? 3 : 0, l, vx);
		loadMatFrag(mfrag, pl, INV ? 4 : 1, l, vy);
#pragma unroll
		for (int e = 0; e < 16; e++) bx[e >> 3][e & 7] = vx[e], ay[e >> 3][(e >> 2) & 1][e & 3] = vy[e];
	}

	constexpr double scale = INV ? 8.0 / (32.0 * 32.0 * 32.0) : 1.0; // (2/N)^3, DftPatchSolver.h:214
	// The data plane is read straight from global memory in the A layout, A[i = y = 16mb + j][k = x =
	// 4ks + g]: a load touches 16 rows x 32 B; the 8 k-steps together consume each row's 256 B, served
	// from L1 after the first touch. No LDS, no barrier, the next plane is in flight during the MFMAs.
	const int     aoff = j * N + g;
	const double *cr   = CORR ? corr + (size_t) pid * 6 * NN : nullptr;
	static_assert(!(INV && CORR), "interface terms belong to the forward pass");
	// element (y = 16mb + j, x = 4ks + g) of plane z, minus the interface terms of the faces it lies on,
	// subtracted in the reference's side order W/E, S/N, B/T
	// The x- and y-face terms are loaded unconditionally and masked (no divergent branch around a load,
	// so the plane prefetch stays in flight); only the two z-face planes take a wave-uniform branch.
	const double mW = (g == 0) ? 1.0 : 0.0, mE = (g == 3) ? 1.0 : 0.0, mS = (j == 0) ? 1.0 : 0.0, mN = (j == 15) ? 1.0 : 0.0;
	auto fetch = [&](double(&dst)[2][8], int z) {
		const double *ip = in + ((size_t) pid * N + z) * NN + aoff;
#pragma unroll
		for (int mb = 0; mb < 2; mb++)
#pragma unroll
			for (int ks = 0; ks < 8; ks++) {
				double v = ip[16 * mb * N + 4 * ks];
				if (CORR) { // compile-time: a run-time test here puts every load in its own block with its own wait
					const int y = 16 * mb + j, x = 4 * ks + g;
					if (ks == 0) v -= mW * cr[0 * NN + y + N * z];
					if (ks == 7) v -= mE * cr[1 * NN + y + N * z];
					if (mb == 0) v -= mS * cr[2 * NN + x + N * z];
					if (mb == 1) v -= mN * cr[3 * NN + x + N * z];
				}
				dst[mb][ks] = v;
			}
		if (CORR && (z == 0 || z == N - 1)) {
			const double *cz = cr + (z == 0 ? 4 : 5) * NN;
#pragma unroll
			for (int mb = 0; mb < 2; mb++)
#pragma unroll
				for (int ks = 0; ks < 8; ks++) dst[mb][ks] -= cz[(4 * ks + g) + N * (16 * mb + j)];
		}
	};
	double nxt[2][8];
	fetch(nxt, wave + 4 * it0);
#if TE_STAMPS
#pragma unroll
	for (int e = 0; e < 16; e++) {
		TE_STAMP_PIN(bx[e >> 3][e & 7]);
		TE_STAMP_PIN(ay[e >> 3][(e >> 2) & 1][e & 3]);
		TE_STAMP_PIN(nxt[e >> 3][e & 7]);
	}
#endif
	TE_STAMP(2, true);
#pragma unroll 1
	for (int it = it0; it < it0 + its; it++) {
		const int z = wave + 4 * it;
		double    a[2][8];
#pragma unroll
		for (int mb = 0; mb < 2; mb++)
#pragma unroll
			for (int ks = 0; ks < 8; ks++) a[mb][ks] = nxt[mb][ks];
		fetch(nxt, (it + 1 < it0 + its) ? z + 4 : z);
		// x transform: D1[row y = 16mb + g + 4r][col kx] = sum_x X[y][x] Mx[kx][x]
		v4f64 d1[2][2];
#pragma unroll
		for (int mb = 0; mb < 2; mb++) {
			d1[mb][0] = d1[mb][1] = v4f64{0, 0, 0, 0};
#pragma unroll
			for (int ks = 0; ks < 8; ks++) {
				d1[mb][0] = mfma_f64(a[mb][ks], bx[0][ks], d1[mb][0]);
				d1[mb][1] = mfma_f64(a[mb][ks], bx[1][ks], d1[mb][1]);
			}
		}
#if TE_STAMPS
		TE_STAMP_PIN(d1[0][0][0]); TE_STAMP_PIN(d1[0][1][0]); TE_STAMP_PIN(d1[1][0][0]); TE_STAMP_PIN(d1[1][1][0]);
		if (it == it0) TE_STAMP(3, false);
#endif
		// y transform: D2[row ky = 16mo + g + 4r][col kx] = sum_y My[ky][y] D1[y][kx]
		double *op = out + ((size_t) pid * N + z) * NN;
#pragma unroll
		for (int mo = 0; mo < 2; mo++) {
			v4f64 e0 = v4f64{0, 0, 0, 0}, e1 = v4f64{0, 0, 0, 0};
#pragma unroll
			for (int mb = 0; mb < 2; mb++)
#pragma unroll
				for (int r = 0; r < 4; r++) {
					e0 = mfma_f64(ay[mo][mb][r], d1[mb][0][r], e0);
					e1 = mfma_f64(ay[mo][mb][r], d1[mb][1][r], e1);
				}
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const int ky = 16 * mo + g + 4 * r;
				reinterpret_cast<double2 *>(op + ky * N)[j] = double2{e0[r] * scale, e1[r] * scale};
			}
		}
	}
	TE_STAMP(5, false);
	TE_STAMP(6, true);
	TE_STAMP_FLUSH(stamp_dst, blockIdx.x + gridDim.x * blockIdx.y);
}

// The same pass for levels of at most eight patches (the coarsest levels of a cycle: one patch, eight patches), where a launch is
// one dependent chain per wave -- arguments, plan, matrices + plane, 64 products, stores -- and what bounds it is what ONE compute
// unit can have in flight (profiles/r06_tail_stamps.txt: 2.1-2.5 us for the 48 KiB a four-wave workgroup requests, then 2 us of
// products). One wave per workgroup and per HALF plane: the output columns kx = 2j + nb of one nb -- the x products of a column set
// and the y products that follow are independent of the other set's, so every value is formed by the same instructions in the same
// order as in k_ps_xy (bit-identical); 64 workgroups per patch instead of 8, 20 KiB requested and 32 products per wave.
// grid (P, 64): blockIdx.y = 2 z + nb; 64 threads.
template <bool INV, bool CORR = false>
__global__ __launch_bounds__(64) void k_ps_xy_half(int P, const int32_t *__restrict__ plan, const double *__restrict__ mfrag,
                                                   const double *__restrict__ in, const double *__restrict__ corr,
                                                   double *__restrict__ out TE_STAMP_PARAM)
{
	constexpr int N = 32, NN = N * N;
	TE_STAMP_DECL;
	TE_STAMP(0, false);
	const int l = threadIdx.x & 63, j = l & 15, g = l >> 4;
	const int pid = blockIdx.x, z = blockIdx.y >> 1, nb = blockIdx.y & 1;
	if (pid >= P) return;
	const int pl = plan[pid];
	TE_STAMP(1, true);
	double bx[8], ay[2][2][4];
	{
		// (chunks 4 nb .. 4 nb + 3 of the x matrix are bx[nb][0..7] of k_ps_xy)
		const double2 *F = reinterpret_cast<const double2 *>(mfrag + ((size_t) pl * 6 + (INV ? 3 : 0)) * 1024) + (size_t) nb * 4 * 64 + l;
#pragma unroll
		for (int c = 0; c < 4; c++) {
			const double2 t = F[c * 64];
			bx[2 * c] = t.x, bx[2 * c + 1] = t.y;
		}
		double vy[16];
		loadMatFrag(mfrag, pl, INV ? 4 : 1, l, vy);
#pragma unroll
		for (int e = 0; e < 16; e++) ay[e >> 3][(e >> 2) & 1][e & 3] = vy[e];
	}
	constexpr double scale = INV ? 8.0 / (32.0 * 32.0 * 32.0) : 1.0;
	static_assert(!(INV && CORR), "interface terms belong to the forward pass");
	const double *cr = CORR ? corr + (size_t) pid * 6 * NN : nullptr;
	const double  mW = (g == 0) ? 1.0 : 0.0, mE = (g == 3) ? 1.0 : 0.0, mS = (j == 0) ? 1.0 : 0.0, mN = (j == 15) ? 1.0 : 0.0;
	double        a[2][8];
	{ // (k_ps_xy's fetch)
		const double *ip = in + ((size_t) pid * N + z) * NN + j * N + g;
#pragma unroll
		for (int mb = 0; mb < 2; mb++)
#pragma unroll
			for (int ks = 0; ks < 8; ks++) {
				double v = ip[16 * mb * N + 4 * ks];
				if (CORR) {
					const int y = 16 * mb + j, x = 4 * ks + g;
					if (ks == 0) v -= mW * cr[0 * NN + y + N * z];
					if (ks == 7) v -= mE * cr[1 * NN + y + N * z];
					if (mb == 0) v -= mS * cr[2 * NN + x + N * z];
					if (mb == 1) v -= mN * cr[3 * NN + x + N * z];
				}
				a[mb][ks] = v;
			}
		if (CORR && (z == 0 || z == N - 1)) {
			const double *cz = cr + (z == 0 ? 4 : 5) * NN;
#pragma unroll
			for (int mb = 0; mb < 2; mb++)
#pragma unroll
				for (int ks = 0; ks < 8; ks++) a[mb][ks] -= cz[(4 * ks + g) + N * (16 * mb + j)];
		}
	}
#if TE_STAMPS
#pragma unroll
	for (int e = 0; e < 16; e++) {
		TE_STAMP_PIN(bx[e & 7]);
		TE_STAMP_PIN(ay[e >> 3][(e >> 2) & 1][e & 3]);
		TE_STAMP_PIN(a[e >> 3][e & 7]);
	}
#endif
	TE_STAMP(2, true);
	v4f64 d1[2];
#pragma unroll
	for (int mb = 0; mb < 2; mb++) {
		d1[mb] = v4f64{0, 0, 0, 0};
#pragma unroll
		for (int ks = 0; ks < 8; ks++) d1[mb] = mfma_f64(a[mb][ks], bx[ks], d1[mb]);
	}
#if TE_STAMPS
	TE_STAMP_PIN(d1[0][0]); TE_STAMP_PIN(d1[1][0]);
	TE_STAMP(3, false);
#endif
	double *op = out + ((size_t) pid * N + z) * NN + nb;
#pragma unroll
	for (int mo = 0; mo < 2; mo++) {
		v4f64 e = v4f64{0, 0, 0, 0};
#pragma unroll
		for (int mb = 0; mb < 2; mb++)
#pragma unroll
			for (int r = 0; r < 4; r++) e = mfma_f64(ay[mo][mb][r], d1[mb][r], e);
#pragma unroll
		for (int r = 0; r < 4; r++) op[(16 * mo + g + 4 * r) * N + 2 * j] = e[r] * scale;
	}
	TE_STAMP(5, false);
	TE_STAMP(6, true);
	TE_STAMP_FLUSH(stamp_dst, blockIdx.x + gridDim.x * blockIdx.y);
}

// lam: [nplans][3][32] = 4 sin^2(.), eigenvalue = -(lam_x rh2x + lam_y rh2y + lam_z rh2z)
// (FftwPatchSolver.h:143-168). One wave = one x-row (fixed y) of a patch, all z.
static __global__ __launch_bounds__(256) void k_ps_z(int P, const int32_t *__restrict__ plan, const double *__restrict__ mfrag,
                                              const double *__restrict__ lam, const int32_t *__restrict__ zero_mode,
                                              const double *__restrict__ rh2, const double *__restrict__ in,
                                              double *__restrict__ out TE_STAMP_PARAM)
{
	constexpr int N = 32, NN = N * N, NNN = N * N * N;
	TE_STAMP_DECL;
	TE_STAMP(0, false);
	const int     wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, g = l >> 4;
	const int     pid  = blockIdx.x; // one workgroup per patch and segment (see k_ps_xy); wave w owns rows y = w + 4 it
	const int     its = 8 / gridDim.y, it0 = blockIdx.y * its;
	if (pid >= P) return;
	const int     pl = plan[pid];
	TE_STAMP(1, true);
	double af[2][8];    // A[i = kz = 16mb + j][k = z = 4ks + g]
	double ai[2][2][4]; // A[i = z = 16mo + j][k-step (mb, r) = kz = 16mb + g + 4r]
	{
		double vf[16], vi[16];
		loadMatFrag(mfrag, pl, 2, l, vf);
		loadMatFrag(mfrag, pl, 5, l, vi);
#pragma unroll
		for (int e = 0; e < 16; e++) af[e >> 3][e & 7] = vf[e], ai[e >> 3][(e >> 2) & 1][e & 3] = vi[e];
	}

	const double *lm = lam + (size_t) pl * 3 * N;
	const double *rh = rh2 + (size_t) pid * 3;
	const double  lx0 = lm[2 * j] * rh[0], lx1 = lm[2 * j + 1] * rh[0];
	double        ez[2][4];
#pragma unroll
	for (int mb = 0; mb < 2; mb++)
#pragma unroll
		for (int r = 0; r < 4; r++) ez[mb][r] = lm[2 * N + 16 * mb + g + 4 * r] * rh[2];
	const bool zmp = zero_mode[pl] != 0;

	double2 nxt[8]; // B[k = z = 4ks + g][cols x = 2j, 2j+1]
	{
		const double *ip = in + (size_t) pid * NNN + (wave + 4 * it0) * N;
#pragma unroll
		for (int ks = 0; ks < 8; ks++) nxt[ks] = reinterpret_cast<const double2 *>(ip + (4 * ks + g) * NN)[j];
	}
	TE_STAMP(2, true);
#pragma unroll 1
	for (int it = it0; it < it0 + its; it++) {
		const int y = wave + 4 * it;
		double2   v[8];
#pragma unroll
		for (int ks = 0; ks < 8; ks++) v[ks] = nxt[ks];
		{
			const int     yn = (it + 1 < it0 + its) ? y + 4 : y;
			const double *ip = in + (size_t) pid * NNN + yn * N;
#pragma unroll
			for (int ks = 0; ks < 8; ks++) nxt[ks] = reinterpret_cast<const double2 *>(ip + (4 * ks + g) * NN)[j];
		}
		const double ly   = lm[N + y] * rh[1];
		const double exy0 = lx0 + ly, exy1 = lx1 + ly;
		const bool   zm   = zmp && y == 0 && j == 0;
		v4f64        d0[2], d1[2]; // D[row kz = 16mb + g + 4r][col]: even / odd x
#pragma unroll
		for (int mb = 0; mb < 2; mb++) {
			d0[mb] = d1[mb] = v4f64{0, 0, 0, 0};
#pragma unroll
			for (int ks = 0; ks < 8; ks++) {
				d0[mb] = mfma_f64(af[mb][ks], v[ks].x, d0[mb]);
				d1[mb] = mfma_f64(af[mb][ks], v[ks].y, d1[mb]);
			}
#pragma unroll
			for (int r = 0; r < 4; r++) {
				d0[mb][r] /= -(exy0 + ez[mb][r]);
				d1[mb][r] /= -(exy1 + ez[mb][r]);
				if (zm && 16 * mb + g + 4 * r == 0) d0[mb][r] = 0.0; // FftwPatchSolver.h:197
			}
		}
		double *op = out + (size_t) pid * NNN + y * N;
#pragma unroll
		for (int mo = 0; mo < 2; mo++) {
			v4f64 e0 = v4f64{0, 0, 0, 0}, e1 = v4f64{0, 0, 0, 0};
#pragma unroll
			for (int mb = 0; mb < 2; mb++)
#pragma unroll
				for (int r = 0; r < 4; r++) {
					e0 = mfma_f64(ai[mo][mb][r], d0[mb][r], e0);
					e1 = mfma_f64(ai[mo][mb][r], d1[mb][r], e1);
				}
#pragma unroll
			for (int r = 0; r < 4; r++) reinterpret_cast<double2 *>(op + (16 * mo + g + 4 * r) * NN)[j] = double2{e0[r], e1[r]};
		}
	}
	TE_STAMP(5, false);
	TE_STAMP(6, true);
	TE_STAMP_FLUSH(stamp_dst, blockIdx.x + gridDim.x * blockIdx.y);
}

// k_ps_z for levels of at most eight patches (see k_ps_xy_half): one wave per workgroup and per half row -- the columns x = 2j + xs
// of one parity xs; grid (P, 64): blockIdx.y = 2 y + xs; 64 threads. Bit-identical to k_ps_z.
static __global__ __launch_bounds__(64) void k_ps_z_half(int P, const int32_t *__restrict__ plan, const double *__restrict__ mfrag,
                                                         const double *__restrict__ lam, const int32_t *__restrict__ zero_mode,
                                                         const double *__restrict__ rh2, const double *__restrict__ in,
                                                         double *__restrict__ out TE_STAMP_PARAM)
{
	constexpr int N = 32, NN = N * N, NNN = N * N * N;
	TE_STAMP_DECL;
	TE_STAMP(0, false);
	const int l = threadIdx.x & 63, j = l & 15, g = l >> 4;
	const int pid = blockIdx.x, y = blockIdx.y >> 1, xs = blockIdx.y & 1;
	if (pid >= P) return;
	const int pl = plan[pid];
	TE_STAMP(1, true);
	double af[2][8], ai[2][2][4];
	{
		double vf[16], vi[16];
		loadMatFrag(mfrag, pl, 2, l, vf);
		loadMatFrag(mfrag, pl, 5, l, vi);
#pragma unroll
		for (int e = 0; e < 16; e++) af[e >> 3][e & 7] = vf[e], ai[e >> 3][(e >> 2) & 1][e & 3] = vi[e];
	}
	const double *lm = lam + (size_t) pl * 3 * N;
	const double *rh = rh2 + (size_t) pid * 3;
	const double  lx = lm[2 * j + xs] * rh[0];
	double        ez[2][4];
#pragma unroll
	for (int mb = 0; mb < 2; mb++)
#pragma unroll
		for (int r = 0; r < 4; r++) ez[mb][r] = lm[2 * N + 16 * mb + g + 4 * r] * rh[2];
	const bool zm = zero_mode[pl] != 0 && y == 0 && j == 0 && xs == 0;
	double     v[8];
	{
		const double *ip = in + (size_t) pid * NNN + y * N + 2 * j + xs;
#pragma unroll
		for (int ks = 0; ks < 8; ks++) v[ks] = ip[(4 * ks + g) * NN];
	}
#if TE_STAMPS
#pragma unroll
	for (int e = 0; e < 16; e++) {
		TE_STAMP_PIN(af[e >> 3][e & 7]);
		TE_STAMP_PIN(ai[e >> 3][(e >> 2) & 1][e & 3]);
		TE_STAMP_PIN(v[e & 7]);
	}
#endif
	TE_STAMP(2, true);
	const double ly = lm[N + y] * rh[1];
	const double exy = lx + ly;
	v4f64        d[2];
#pragma unroll
	for (int mb = 0; mb < 2; mb++) {
		d[mb] = v4f64{0, 0, 0, 0};
#pragma unroll
		for (int ks = 0; ks < 8; ks++) d[mb] = mfma_f64(af[mb][ks], v[ks], d[mb]);
#pragma unroll
		for (int r = 0; r < 4; r++) {
			d[mb][r] /= -(exy + ez[mb][r]);
			if (zm && 16 * mb + g + 4 * r == 0) d[mb][r] = 0.0; // FftwPatchSolver.h:197
		}
	}
#if TE_STAMPS
	TE_STAMP_PIN(d[0][0]); TE_STAMP_PIN(d[1][0]);
	TE_STAMP(3, false);
#endif
	double *op = out + (size_t) pid * NNN + y * N + 2 * j + xs;
#pragma unroll
	for (int mo = 0; mo < 2; mo++) {
		v4f64 e = v4f64{0, 0, 0, 0};
#pragma unroll
		for (int mb = 0; mb < 2; mb++)
#pragma unroll
			for (int r = 0; r < 4; r++) e = mfma_f64(ai[mo][mb][r], d[mb][r], e);
#pragma unroll
		for (int r = 0; r < 4; r++) op[(16 * mo + g + 4 * r) * NN] = e[r];
	}
	TE_STAMP(5, false);
	TE_STAMP(6, true);
	TE_STAMP_FLUSH(stamp_dst, blockIdx.x + gridDim.x * blockIdx.y);
}

// ---- single-pass patch solve -------------------------------------------------------------------
// The three kernels above move every site through HBM three times (48 B/site). k_ps_fused keeps the
// whole 32^3 patch on chip: one workgroup of 8 waves per patch, the patch lives in
// registers (64 doubles per lane) and crosses between the "plane" distribution (x,y transforms: wave
// w owns planes w, w+8, w+16, w+24) and the "column" distribution (z transform: wave w owns the
// (z, kx) slabs of ky = 2w, 2w+1, 16+2w, 17+2w) through one 144 KiB LDS image, half a patch at a time:
//   A   per plane: load f (minus interface terms), x,y forward            [as k_ps_xy<false>]
//   X1  planes -> columns, ky halves (rows of the y-transform's two output tiles)
//   Z   per slab: z forward, eigenvalue divide, zero mode, z inverse       [as k_ps_z]
//   X2  columns -> planes, z halves; per plane x,y inverse, scale, store   [as k_ps_xy<true>]
// 16 B/site of HBM traffic (read f, write u) + the face terms. The arithmetic (operands, MFMA order)
// is that of the three-pass kernels, so the result is bit-identical to them.
constexpr int PSF_S1        = 32 * 32;          // X1 image: [ky_local 16][z 32][kx 32], slab stride in doubles
constexpr int PSF_P2        = 36;               // X2 image: [z_local 16][ky 32][kx 32 (+4)]: the A-layout reads of
constexpr int PSF_S2        = 32 * PSF_P2;      //   phase C step 16 rows at once; pitch 36 spreads them over the banks
constexpr int PSF_LDS_BYTES = 16 * PSF_S2 * 8; // 147456
#ifdef PSF_TIMING // tools/psf_bench.hip: shader-clock stamps of workgroup 0 at the phase boundaries
static __device__ long long psf_stamp[8][12];
#define PSF_STAMP(k) do { if (blockIdx.x == 2048 && l == 0) psf_stamp[wave][k] = clock64(); } while (0)
#else
#define PSF_STAMP(k)
#endif

template <bool CORR>
__global__ __launch_bounds__(512) void k_ps_fused(int P, const int32_t *__restrict__ plan, const double *__restrict__ mats,
                                                  const double *__restrict__ lam, const int32_t *__restrict__ zero_mode,
                                                  const double *__restrict__ rh2, const double *__restrict__ in,
                                                  const double *__restrict__ corr, double *__restrict__ out,
                                                  const int32_t *__restrict__ list /* may be null: patches list[0..P) */)
{
	constexpr int N = 32, NN = N * N;
	extern __shared__ __attribute__((aligned(16))) double xbuf[];
	const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, g = l >> 4;
	const int slot = xcdRemap(blockIdx.x, P);
	if (slot >= P) return;
	const int pid = list ? list[slot] : slot;
	const int     pl = plan[pid];
	const double *M  = mats + (size_t) pl * 6 * NN;
	PSF_STAMP(0);

	// ---- A: x,y forward of this wave's four planes ---------------------------------------------
	v4f64 hi[4][2]; // ky >= 16 half of the y-transform output, parked until the image is free again
	{
		const double *Mx = M, *My = M + NN;
		double        bx[2][8], ay[2][2][4];
#pragma unroll
		for (int nb = 0; nb < 2; nb++)
#pragma unroll
			for (int ks = 0; ks < 8; ks++) bx[nb][ks] = Mx[(2 * j + nb) * N + 4 * ks + g];
#pragma unroll
		for (int mo = 0; mo < 2; mo++)
#pragma unroll
			for (int mb = 0; mb < 2; mb++)
#pragma unroll
				for (int r = 0; r < 4; r++) ay[mo][mb][r] = My[(16 * mo + j) * N + 16 * mb + g + 4 * r];

		const int     aoff = j * N + g;
		const double *cr   = CORR ? corr + (size_t) pid * 6 * NN : nullptr;
		const double  mW = (g == 0) ? 1.0 : 0.0, mE = (g == 3) ? 1.0 : 0.0, mS = (j == 0) ? 1.0 : 0.0, mN = (j == 15) ? 1.0 : 0.0;
		auto fetch = [&](double(&dst)[2][8], int z) { // see k_ps_xy
			const double *ip = in + ((size_t) pid * N + z) * NN + aoff;
#pragma unroll
			for (int mb = 0; mb < 2; mb++)
#pragma unroll
				for (int ks = 0; ks < 8; ks++) {
					double v = ip[16 * mb * N + 4 * ks];
					if (CORR) {
						const int y = 16 * mb + j, x = 4 * ks + g;
						if (ks == 0) v -= mW * cr[0 * NN + y + N * z];
						if (ks == 7) v -= mE * cr[1 * NN + y + N * z];
						if (mb == 0) v -= mS * cr[2 * NN + x + N * z];
						if (mb == 1) v -= mN * cr[3 * NN + x + N * z];
					}
					dst[mb][ks] = v;
				}
			if (CORR && (z == 0 || z == N - 1)) {
				const double *cz = cr + (z == 0 ? 4 : 5) * NN;
#pragma unroll
				for (int mb = 0; mb < 2; mb++)
#pragma unroll
					for (int ks = 0; ks < 8; ks++) dst[mb][ks] -= cz[(4 * ks + g) + N * (16 * mb + j)];
			}
		};
		double a[2][8];
		fetch(a, wave);
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int z = wave + 8 * i;
			v4f64     d1[2][2];
#pragma unroll
			for (int mb = 0; mb < 2; mb++) {
				d1[mb][0] = d1[mb][1] = v4f64{0, 0, 0, 0};
#pragma unroll
				for (int ks = 0; ks < 8; ks++) {
					d1[mb][0] = mfma_f64(a[mb][ks], bx[0][ks], d1[mb][0]);
					d1[mb][1] = mfma_f64(a[mb][ks], bx[1][ks], d1[mb][1]);
				}
			}
			if (i < 3) fetch(a, z + 8); // the next plane's loads fly behind the y transform
#pragma unroll
			for (int mo = 0; mo < 2; mo++) {
				v4f64 e0 = v4f64{0, 0, 0, 0}, e1 = v4f64{0, 0, 0, 0};
#pragma unroll
				for (int mb = 0; mb < 2; mb++)
#pragma unroll
					for (int r = 0; r < 4; r++) {
						e0 = mfma_f64(ay[mo][mb][r], d1[mb][0][r], e0);
						e1 = mfma_f64(ay[mo][mb][r], d1[mb][1][r], e1);
					}
				if (mo == 0) {
#pragma unroll
					for (int r = 0; r < 4; r++)
						*reinterpret_cast<double2 *>(xbuf + (g + 4 * r) * PSF_S1 + z * N + 2 * j) = double2{e0[r], e1[r]};
				} else {
					hi[i][0] = e0, hi[i][1] = e1;
				}
			}
		}
	}
	PSF_STAMP(1);
	ldsBarrier();
	PSF_STAMP(2);

	// ---- X1: this wave's four (z, kx) slabs; V[t] = slab of ky = 16 (t >> 1) + 2 wave + (t & 1) ----
	double2 V[4][8]; // B[k = z = 4ks + g][cols kx = 2j, 2j + 1]
#pragma unroll
	for (int t = 0; t < 2; t++)
#pragma unroll
		for (int ks = 0; ks < 8; ks++)
			V[t][ks] = *reinterpret_cast<const double2 *>(xbuf + (2 * wave + t) * PSF_S1 + (4 * ks + g) * N + 2 * j);
	ldsBarrier();
#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int r = 0; r < 4; r++)
			*reinterpret_cast<double2 *>(xbuf + (g + 4 * r) * PSF_S1 + (wave + 8 * i) * N + 2 * j) = double2{hi[i][0][r], hi[i][1][r]};
	ldsBarrier();
#pragma unroll
	for (int t = 0; t < 2; t++)
#pragma unroll
		for (int ks = 0; ks < 8; ks++)
			V[2 + t][ks] = *reinterpret_cast<const double2 *>(xbuf + (2 * wave + t) * PSF_S1 + (4 * ks + g) * N + 2 * j);

	PSF_STAMP(3);
	// ---- Z: forward, divide by the eigenvalue, inverse -------------------------------------------
	v4f64 D[4][2][2]; // [slab][mb][even / odd kx]: rows kz = 16mb + g + 4r
	{
		const double *Mf = M + 2 * NN;
		double        af[2][8];
#pragma unroll
		for (int mb = 0; mb < 2; mb++)
#pragma unroll
			for (int ks = 0; ks < 8; ks++) af[mb][ks] = Mf[(16 * mb + j) * N + 4 * ks + g];
		const double *lm = lam + (size_t) pl * 3 * N;
		const double *rh = rh2 + (size_t) pid * 3;
		const double  lx0 = lm[2 * j] * rh[0], lx1 = lm[2 * j + 1] * rh[0];
		double        ez[2][4];
#pragma unroll
		for (int mb = 0; mb < 2; mb++)
#pragma unroll
			for (int r = 0; r < 4; r++) ez[mb][r] = lm[2 * N + 16 * mb + g + 4 * r] * rh[2];
		const bool zmp = zero_mode[pl] != 0;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const int    ky   = 16 * (t >> 1) + 2 * wave + (t & 1);
			const double ly   = lm[N + ky] * rh[1];
			const double exy0 = lx0 + ly, exy1 = lx1 + ly;
			const bool   zm   = zmp && ky == 0 && j == 0;
#pragma unroll
			for (int mb = 0; mb < 2; mb++) {
				v4f64 d0 = v4f64{0, 0, 0, 0}, d1 = v4f64{0, 0, 0, 0};
#pragma unroll
				for (int ks = 0; ks < 8; ks++) {
					d0 = mfma_f64(af[mb][ks], V[t][ks].x, d0);
					d1 = mfma_f64(af[mb][ks], V[t][ks].y, d1);
				}
#pragma unroll
				for (int r = 0; r < 4; r++) {
					d0[r] /= -(exy0 + ez[mb][r]);
					d1[r] /= -(exy1 + ez[mb][r]);
					if (zm && 16 * mb + g + 4 * r == 0) d0[r] = 0.0; // FftwPatchSolver.h:197
				}
				D[t][mb][0] = d0, D[t][mb][1] = d1;
			}
		}
	}
	PSF_STAMP(4);
	v4f64 E[4][2][2]; // [slab][mo][even / odd kx]: rows z = 16mo + g + 4r
	{
		const double *Mi = M + 5 * NN;
		double        ai[2][2][4];
#pragma unroll
		for (int mo = 0; mo < 2; mo++)
#pragma unroll
			for (int mb = 0; mb < 2; mb++)
#pragma unroll
				for (int r = 0; r < 4; r++) ai[mo][mb][r] = Mi[(16 * mo + j) * N + 16 * mb + g + 4 * r];
#pragma unroll
		for (int t = 0; t < 4; t++)
#pragma unroll
			for (int mo = 0; mo < 2; mo++) {
				v4f64 e0 = v4f64{0, 0, 0, 0}, e1 = v4f64{0, 0, 0, 0};
#pragma unroll
				for (int mb = 0; mb < 2; mb++)
#pragma unroll
					for (int r = 0; r < 4; r++) {
						e0 = mfma_f64(ai[mo][mb][r], D[t][mb][0][r], e0);
						e1 = mfma_f64(ai[mo][mb][r], D[t][mb][1][r], e1);
					}
				E[t][mo][0] = e0, E[t][mo][1] = e1;
			}
	}

	PSF_STAMP(5);
	// ---- X2 + C: back to planes, z halves; x,y inverse, scale, store ------------------------------
	{
		const double *Mx = M + 3 * NN, *My = M + 4 * NN;
		double        bx[2][8], ay[2][2][4];
#pragma unroll
		for (int nb = 0; nb < 2; nb++)
#pragma unroll
			for (int ks = 0; ks < 8; ks++) bx[nb][ks] = Mx[(2 * j + nb) * N + 4 * ks + g];
#pragma unroll
		for (int mo = 0; mo < 2; mo++)
#pragma unroll
			for (int mb = 0; mb < 2; mb++)
#pragma unroll
				for (int r = 0; r < 4; r++) ay[mo][mb][r] = My[(16 * mo + j) * N + 16 * mb + g + 4 * r];
		constexpr double scale = 8.0 / (32.0 * 32.0 * 32.0); // (2/N)^3, DftPatchSolver.h:214
#pragma unroll
		for (int zh = 0; zh < 2; zh++) {
			ldsBarrier(); // every wave is done reading the previous image
			PSF_STAMP(6 + 3 * zh);
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const int ky = 16 * (t >> 1) + 2 * wave + (t & 1);
#pragma unroll
				for (int r = 0; r < 4; r++)
					*reinterpret_cast<double2 *>(xbuf + (g + 4 * r) * PSF_S2 + ky * PSF_P2 + 2 * j) =
					    double2{E[t][zh][0][r], E[t][zh][1][r]};
			}
			ldsBarrier();
			PSF_STAMP(7 + 3 * zh);
#pragma unroll
			for (int ii = 0; ii < 2; ii++) {
				const int     zl = wave + 8 * ii, z = 16 * zh + zl;
				const double *ip = xbuf + zl * PSF_S2 + j * PSF_P2 + g; // A[i = ky = 16mb + j][k = kx = 4ks + g]
				double        a[2][8];
#pragma unroll
				for (int mb = 0; mb < 2; mb++)
#pragma unroll
					for (int ks = 0; ks < 8; ks++) a[mb][ks] = ip[16 * mb * PSF_P2 + 4 * ks];
				v4f64 d1[2][2];
#pragma unroll
				for (int mb = 0; mb < 2; mb++) {
					d1[mb][0] = d1[mb][1] = v4f64{0, 0, 0, 0};
#pragma unroll
					for (int ks = 0; ks < 8; ks++) {
						d1[mb][0] = mfma_f64(a[mb][ks], bx[0][ks], d1[mb][0]);
						d1[mb][1] = mfma_f64(a[mb][ks], bx[1][ks], d1[mb][1]);
					}
				}
				double *op = out + ((size_t) pid * N + z) * NN;
#pragma unroll
				for (int mo = 0; mo < 2; mo++) {
					v4f64 e0 = v4f64{0, 0, 0, 0}, e1 = v4f64{0, 0, 0, 0};
#pragma unroll
					for (int mb = 0; mb < 2; mb++)
#pragma unroll
						for (int r = 0; r < 4; r++) {
							e0 = mfma_f64(ay[mo][mb][r], d1[mb][0][r], e0);
							e1 = mfma_f64(ay[mo][mb][r], d1[mb][1][r], e1);
						}
#pragma unroll
					for (int r = 0; r < 4; r++)
						reinterpret_cast<double2 *>(op + (16 * mo + g + 4 * r) * N)[j] = double2{e0[r] * scale, e1[r] * scale};
				}
			}
			PSF_STAMP(8 + 3 * zh);
		}
	}
}
} // namespace te
