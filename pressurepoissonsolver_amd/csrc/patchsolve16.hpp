// Reference block-Jacobi smoother for 16^3 patches: the exact per-patch solve of FftwPatchSolver.h:173-206 (transforms =
// DftPatchSolver.h:237-289 matrices) in ONE launch and one pass over HBM. A 16x16 transform is exactly one
// v_mfma_f64_16x16x4 tile (four k-steps), a 16^3 patch is 32 KiB: one workgroup of four waves per patch keeps it in LDS.
//   A  per z-plane (wave w: planes w, w+4, w+8, w+12): f (minus the interface terms, CORR) straight from global memory as
//      the A operand, x forward, y forward chained in registers (a D tile is the next product's B operand when k-step r
//      stands for row g + 4r), result to the LDS image T[z][ky][kx]
//   Z  per ky-slice (wave w: slices w, w+4, ...): z forward, eigenvalue division, zero mode, z inverse, chained; in place
//   C  per z-plane: x inverse (A operand from the image), y inverse chained, (2/N)^3, store
// 16 B/site + face terms instead of the seven passes of k_patch_rhs3d + 6 x k_dst_axis3d (112 B/site). Sums run in the MFMA's
// order: equal to the one-axis-per-launch form to rounding (tests: <= 1e-11 against the oracle).
#pragma once
#include "patchsolve32.hpp"

namespace te
{
constexpr int PS16_ROW = 17;            // row pitch of the image (A-operand reads step 16 rows at once)
constexpr int PS16_ZS  = 16 * PS16_ROW; // plane pitch: 272 = 16 mod 32, the two z's of a half-wave read disjoint banks

template <bool CORR>
__global__ __launch_bounds__(256) void k_ps16(int P, const int32_t *__restrict__ plan, const double *__restrict__ mats,
                                              const double *__restrict__ lam, const int32_t *__restrict__ zero_mode,
                                              const double *__restrict__ rh2, const double *__restrict__ in,
                                              const double *__restrict__ corr, double *__restrict__ out)
{
	constexpr int N = 16, NN = N * N, NNN = N * N * N;
	__shared__ double T[N * PS16_ZS];
	const int pid = blockIdx.x;
	if (pid >= P) return;
	const int     wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, g = l >> 4;
	const int     pl = plan[pid];
	const double *M  = mats + (size_t) pl * 6 * NN; // forward x, y, z, inverse x, y, z; row-major y_i = sum_j M[i][j] x_j
	// matrix fragments: as B operand B[k = 4ks + g][col j] = M[j][4ks + g] (x forward, x inverse); as A operand of a chained
	// product A[i = j][k-step r <-> g + 4r] = M[j][g + 4r] (y forward, z inverse, y inverse); z forward takes its data from the
	// image in natural k order: A[i = j][k = 4ks + g]
	double bxf[4], ayf[4], azf[4], azi[4], bxi[4], ayi[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		bxf[k] = M[0 * NN + j * N + 4 * k + g];
		ayf[k] = M[1 * NN + j * N + g + 4 * k];
		azf[k] = M[2 * NN + j * N + 4 * k + g];
		bxi[k] = M[3 * NN + j * N + 4 * k + g];
		ayi[k] = M[4 * NN + j * N + g + 4 * k];
		azi[k] = M[5 * NN + j * N + g + 4 * k];
	}
	const double *ip = in + (size_t) pid * NNN;
	const double *cr = CORR ? corr + (size_t) pid * 6 * NN : nullptr;
	// ---- A: x, y forward per plane
#pragma unroll 1
	for (int i = 0; i < 4; i++) {
		const int z = wave + 4 * i;
		double    a[4];
#pragma unroll
		for (int ks = 0; ks < 4; ks++) {
			const int x = 4 * ks + g, y = j;
			double    v = ip[z * NN + y * N + x];
			if (CORR) { // the reference's side order W/E, S/N, B/T (StarPatchOp.h:185-203)
				if (x == 0) v -= cr[0 * NN + y + N * z];
				if (x == N - 1) v -= cr[1 * NN + y + N * z];
				if (y == 0) v -= cr[2 * NN + x + N * z];
				if (y == N - 1) v -= cr[3 * NN + x + N * z];
				if (z == 0) v -= cr[4 * NN + x + N * y];
				if (z == N - 1) v -= cr[5 * NN + x + N * y];
			}
			a[ks] = v;
		}
		v4f64 d1 = v4f64{0, 0, 0, 0}, d2 = v4f64{0, 0, 0, 0};
#pragma unroll
		for (int ks = 0; ks < 4; ks++) d1 = mfma_f64(a[ks], bxf[ks], d1); // D1[row y = g + 4r][col kx = j]
#pragma unroll
		for (int r = 0; r < 4; r++) d2 = mfma_f64(ayf[r], d1[r], d2); // D2[row ky = g + 4r][col kx = j]
#pragma unroll
		for (int r = 0; r < 4; r++) T[z * PS16_ZS + (g + 4 * r) * PS16_ROW + j] = d2[r];
	}
	__syncthreads();
	// ---- Z: z forward, eigenvalues, z inverse per ky-slice
	const double *lm = lam + (size_t) pl * 3 * N;
	const double  rhx = rh2[(size_t) pid * 3], rhy = rh2[(size_t) pid * 3 + 1], rhz = rh2[(size_t) pid * 3 + 2];
	const double  lx  = lm[j] * rhx;
	const bool    zmp = zero_mode[pl] != 0;
#pragma unroll 1
	for (int i = 0; i < 4; i++) {
		const int ky = wave + 4 * i;
		v4f64     d3 = v4f64{0, 0, 0, 0}, d4 = v4f64{0, 0, 0, 0};
#pragma unroll
		for (int ks = 0; ks < 4; ks++) d3 = mfma_f64(azf[ks], T[(4 * ks + g) * PS16_ZS + ky * PS16_ROW + j], d3); // D3[row kz = g + 4r][col kx]
		const double exy = lx + lm[N + ky] * rhy;
#pragma unroll
		for (int r = 0; r < 4; r++) {
			d3[r] /= -(exy + lm[2 * N + g + 4 * r] * rhz);
			if (zmp && j == 0 && ky == 0 && g + 4 * r == 0) d3[r] = 0.0; // FftwPatchSolver.h:197
		}
#pragma unroll
		for (int r = 0; r < 4; r++) d4 = mfma_f64(azi[r], d3[r], d4); // D4[row z = g + 4r][col kx]
#pragma unroll
		for (int r = 0; r < 4; r++) T[(g + 4 * r) * PS16_ZS + ky * PS16_ROW + j] = d4[r];
	}
	__syncthreads();
	// ---- C: x, y inverse per plane, scale, store
	constexpr double scale = 8.0 / ((double) N * N * N); // (2/N)^3, DftPatchSolver.h:214
	double          *op    = out + (size_t) pid * NNN;
#pragma unroll 1
	for (int i = 0; i < 4; i++) {
		const int z  = wave + 4 * i;
		v4f64     d5 = v4f64{0, 0, 0, 0}, d6 = v4f64{0, 0, 0, 0};
#pragma unroll
		for (int ks = 0; ks < 4; ks++) d5 = mfma_f64(T[z * PS16_ZS + j * PS16_ROW + 4 * ks + g], bxi[ks], d5); // D5[row ky = g + 4r][col x = j]
#pragma unroll
		for (int r = 0; r < 4; r++) d6 = mfma_f64(ayi[r], d5[r], d6); // D6[row y = g + 4r][col x = j]
#pragma unroll
		for (int r = 0; r < 4; r++) op[z * NN + (g + 4 * r) * N + j] = d6[r] * scale;
	}
}

// 4^3 and 8^3 patches (the reference's test meshes): the whole solve in one launch as well, one workgroup per patch, the six
// one-axis transforms of k_dst_axis3d back to back on two LDS copies of the patch (same sums in the same order), interface
// terms subtracted while loading (k_face_corr3d) -- one launch and 16 B/site instead of seven and 112.
template <int N, bool CORR>
__global__ __launch_bounds__(256) void k_ps_small(int P, const int32_t *__restrict__ plan, const double *__restrict__ mats,
                                                  const double *__restrict__ lam, const int32_t *__restrict__ zero_mode,
                                                  const double *__restrict__ rh2, const double *__restrict__ in,
                                                  const double *__restrict__ corr, double *__restrict__ out)
{
	static_assert(N <= 8, "two copies of the patch and six matrices in static LDS");
	constexpr int NN = N * N, NNN = N * N * N;
	__shared__ double buf[2][NNN], Ms[6 * NN];
	const int pid = blockIdx.x;
	if (pid >= P) return;
	const int pl = plan[pid];
	for (int i = threadIdx.x; i < 6 * NN; i += 256) Ms[i] = mats[(size_t) pl * 6 * NN + i];
	const double *cr = CORR ? corr + (size_t) pid * 6 * NN : nullptr;
	for (int c = threadIdx.x; c < NNN; c += 256) {
		const int x = c % N, y = (c / N) % N, z = c / NN;
		double    v = in[(size_t) pid * NNN + c];
		if (CORR) { // W/E, S/N, B/T (StarPatchOp.h:185-203)
			if (x == 0) v -= cr[0 * NN + y + N * z];
			if (x == N - 1) v -= cr[1 * NN + y + N * z];
			if (y == 0) v -= cr[2 * NN + x + N * z];
			if (y == N - 1) v -= cr[3 * NN + x + N * z];
			if (z == 0) v -= cr[4 * NN + x + N * y];
			if (z == N - 1) v -= cr[5 * NN + x + N * y];
		}
		buf[0][c] = v;
	}
	__syncthreads();
	const double *lm = lam + (size_t) pl * 3 * N;
	const double *rh = rh2 + (size_t) pid * 3;
#pragma unroll
	for (int stage = 0; stage < 6; stage++) {
		const int     ax = stage % 3, st = (ax == 0) ? 1 : (ax == 1 ? N : NN);
		const double *src = buf[stage & 1], *M = Ms + stage * NN;
		double       *dst = buf[(stage & 1) ^ 1];
		for (int c = threadIdx.x; c < NNN; c += 256) {
			const int i = (c / st) % N, base = c - i * st;
			double    acc = 0.0;
#pragma unroll
			for (int j = 0; j < N; j++) acc += M[i * N + j] * src[base + j * st];
			if (stage == 2) {
				const int x = c % N, y = (c / N) % N, z = c / NN;
				acc /= -(lm[x] * rh[0] + lm[N + y] * rh[1] + lm[2 * N + z] * rh[2]);
				if (zero_mode[pl] && c == 0) acc = 0.0;
			}
			if (stage == 5)
				out[(size_t) pid * NNN + c] = acc * (8.0 / ((double) N * N * N));
			else
				dst[c] = acc;
		}
		__syncthreads();
	}
}
} // namespace te
