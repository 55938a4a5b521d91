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
// plane by plane; z-neighbours live in registers, x/y-neighbours in a double-buffered LDS
// plane with a one-cell halo; HBM sees each interior cell once per operand.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace te
{
enum FaceKind : int32_t { FACE_DIRICHLET = 0, FACE_NEUMANN = 1, FACE_LOCAL = 2, FACE_GHOST = 3 };
enum StencilMode : int { MODE_APPLY = 0, MODE_RESID = 1, MODE_JACOBI = 2 };

struct LevelDev {
	int32_t        P;         // local patches
	const int32_t *face_kind; // [P*6]
	const int32_t *face_src;  // [P*6] FACE_LOCAL: neighbour patch; FACE_GHOST: ghost plane slot
	const double  *face_kadj; // [P*6] change of the diagonal's per-axis factor 2 at that face
	const double  *rh2;       // [P*3] 1/h^2
	const double  *ghost;     // [nslots*N*N]
};

template <int N> struct Tile {
	static constexpr int TPB = (N * N < 256) ? N * N : 256;
	static constexpr int CPT = (N * N) / TPB;
	static constexpr int LW  = N + 2; // LDS row length incl. halo
	static constexpr int LSZ = LW * LW;
	static_assert((N * N) % TPB == 0, "plane must tile the workgroup");
};

// Blocks b, b+8, b+16, ... share an XCD (observed round-robin dispatch); give every XCD one
// contiguous run of patches so that x/y/z-neighbour faces are mostly served from that XCD's
// L2. Speed only; correctness never depends on placement.
__device__ __forceinline__ int xcdRemap(int b, int nblocks)
{
	int chunk = (nblocks + 7) >> 3;
	return (b & 7) * chunk + (b >> 3);
}

// value a cell just outside the patch takes, given the cell just inside (`own`)
template <int N>
__device__ __forceinline__ double ghostValue(int kind, int src, double own, const double *__restrict__ u,
                                             const double *__restrict__ ghost, int nbr_cell, int slot_cell)
{
	switch (kind) {
		case FACE_DIRICHLET: return -own;
		case FACE_NEUMANN: return own;
		case FACE_LOCAL: return u[(size_t) src * (N * N * N) + nbr_cell];
		default: return ghost[(size_t) src * (N * N) + slot_cell];
	}
}

// MODE_APPLY : out = A u                     (SchurHelper.h:360-376 + StarPatchOp.h:28-184)
// MODE_RESID : out = f - A u                 (+ Cycle.h:60-61)
// MODE_JACOBI: out = u + omega (f - A u)/diag(A)
// grid: 8*ceil(P*ZS/8) blocks of Tile<N>::TPB threads; ZS z-slabs per patch.
template <int N, int MODE, int ZS>
__global__ __launch_bounds__(Tile<N>::TPB) void k_stencil3d(LevelDev L, const double *__restrict__ u,
                                                            const double *__restrict__ f,
                                                            double *__restrict__ out, double omega)
{
	using T                = Tile<N>;
	constexpr int TPB      = T::TPB;
	constexpr int CPT      = T::CPT;
	constexpr int LW       = T::LW;
	constexpr int NN       = N * N;
	constexpr int NNN      = N * N * N;
	constexpr int ZL       = N / ZS; // planes per slab
	const int     nblocks  = L.P * ZS;
	const int     work     = xcdRemap(blockIdx.x, nblocks);
	if (work >= nblocks) return;
	const int pid = work / ZS;
	const int z0  = (work % ZS) * ZL;
	const int tid = threadIdx.x;

	__shared__ double tile[2][T::LSZ];
	__shared__ double idiag[27];

	const int32_t *fk = L.face_kind + (size_t) pid * 6;
	const int32_t *fs = L.face_src + (size_t) pid * 6;
	const double   rhx = L.rh2[(size_t) pid * 3], rhy = L.rh2[(size_t) pid * 3 + 1],
	             rhz = L.rh2[(size_t) pid * 3 + 2];
	const double *up = u + (size_t) pid * NNN;
	const double *fp = (MODE != MODE_APPLY) ? f + (size_t) pid * NNN : nullptr;
	double       *op = out + (size_t) pid * NNN;

	if (MODE == MODE_JACOBI) {
		for (int e = tid; e < 27; e += TPB) { // TPB may be < 27 for tiny patches
			const double *ka = L.face_kadj + (size_t) pid * 6;
			int           cx = e % 3, cy = (e / 3) % 3, cz = e / 9;
			double        kx = 2.0 + (cx == 0 ? ka[0] : 0.0) + (cx == 2 ? ka[1] : 0.0);
			double        ky = 2.0 + (cy == 0 ? ka[2] : 0.0) + (cy == 2 ? ka[3] : 0.0);
			double        kz = 2.0 + (cz == 0 ? ka[4] : 0.0) + (cz == 2 ? ka[5] : 0.0);
			idiag[e]         = -1.0 / (kx * rhx + ky * rhy + kz * rhz);
		}
	}

	// ---- x/y halo ownership: thread h < 4N owns halo entry (side = h/N, t = h%N) ----------
	const bool has_halo = tid < 4 * N;
	int        h_kind = 0, h_src = 0, h_own = 0, h_nbr = 0, h_lds = 0, h_slot_mul = 0;
	if (has_halo) {
		const int side = tid / N, t = tid % N;
		h_kind = fk[side];
		h_src  = fs[side];
		switch (side) {
			case 0: // west: own (0,t), neighbour's (N-1,t)
				h_own = t * N;
				h_nbr = t * N + (N - 1);
				h_lds = (t + 1) * LW;
				break;
			case 1:
				h_own = t * N + (N - 1);
				h_nbr = t * N;
				h_lds = (t + 1) * LW + N + 1;
				break;
			case 2: // south: own (t,0), neighbour's (t,N-1)
				h_own = t;
				h_nbr = (N - 1) * N + t;
				h_lds = t + 1;
				break;
			default:
				h_own = (N - 1) * N + t;
				h_nbr = t;
				h_lds = (N + 1) * LW + t + 1;
				break;
		}
		h_slot_mul = t; // ghost plane cell (a,b) = (t, z)
	}
	auto haloLoad = [&](int z) -> double {
		double own = (h_kind <= FACE_NEUMANN) ? up[z * NN + h_own] : 0.0;
		return ghostValue<N>(h_kind, h_src, own, u, L.ghost, z * NN + h_nbr, h_slot_mul + N * z);
	};

	// ---- register pipeline over z ------------------------------------------------------------
	double um[CPT], uc[CPT], un[CPT], un2[CPT], fc[CPT], fn[CPT];
	const int kb = fk[4], sb = fs[4], kt = fk[5], st = fs[5];
#pragma unroll
	for (int k = 0; k < CPT; k++) {
		const int i = tid + k * TPB;
		uc[k]       = up[z0 * NN + i];
		if (z0 == 0)
			um[k] = ghostValue<N>(kb, sb, uc[k], u, L.ghost, (N - 1) * NN + i, i);
		else
			um[k] = up[(z0 - 1) * NN + i];
		if (z0 + 1 < N)
			un[k] = up[(z0 + 1) * NN + i];
		else
			un[k] = ghostValue<N>(kt, st, uc[k], u, L.ghost, i, i);
		if (MODE != MODE_APPLY) fc[k] = fp[z0 * NN + i];
	}
	double hv = has_halo ? haloLoad(z0) : 0.0;

#pragma unroll 1
	for (int zz = 0; zz < ZL; zz++) {
		const int z = z0 + zz;
		// prefetch plane z+2 (or the top ghost) and the next halo / rhs plane
#pragma unroll
		for (int k = 0; k < CPT; k++) {
			const int i = tid + k * TPB;
			if (z + 2 < N)
				un2[k] = up[(z + 2) * NN + i];
			else if (z + 2 == N)
				un2[k] = ghostValue<N>(kt, st, un[k], u, L.ghost, i, i);
			else
				un2[k] = 0.0;
			if (MODE != MODE_APPLY) fn[k] = (zz + 1 < ZL) ? fp[(z + 1) * NN + i] : 0.0;
		}
		double hvn = (has_halo && zz + 1 < ZL) ? haloLoad(z + 1) : 0.0;

		double *tl = tile[zz & 1];
#pragma unroll
		for (int k = 0; k < CPT; k++) {
			const int i = tid + k * TPB;
			const int x = i % N, y = i / N;
			tl[(y + 1) * LW + x + 1] = uc[k];
		}
		if (has_halo) tl[h_lds] = hv;
		__syncthreads();
#pragma unroll
		for (int k = 0; k < CPT; k++) {
			const int    i = tid + k * TPB;
			const int    x = i % N, y = i / N;
			const double c = uc[k];
			const double *t0 = tl + (y + 1) * LW + x + 1;
			double lap = (t0[-1] - 2 * c + t0[1]) * rhx;
			lap += (t0[-LW] - 2 * c + t0[LW]) * rhy;
			lap += (um[k] - 2 * c + un[k]) * rhz;
			double r;
			if (MODE == MODE_APPLY) {
				r = lap;
			} else if (MODE == MODE_RESID) {
				r = fc[k] - lap;
			} else {
				const int cx = (x == 0) ? 0 : (x == N - 1 ? 2 : 1);
				const int cy = (y == 0) ? 0 : (y == N - 1 ? 2 : 1);
				const int cz = (z == 0) ? 0 : (z == N - 1 ? 2 : 1);
				r            = c + omega * (fc[k] - lap) * idiag[cx + 3 * cy + 9 * cz];
			}
			op[z * NN + i] = r;
		}
#pragma unroll
		for (int k = 0; k < CPT; k++) {
			um[k] = uc[k];
			uc[k] = un[k];
			un[k] = un2[k];
			if (MODE != MODE_APPLY) fc[k] = fn[k];
		}
		hv = hvn;
	}
}

// Patch-local red-black Gauss-Seidel sweep with neighbour ghosts frozen at the old iterate
// (hybrid GS: Gauss-Seidel inside the patch, Jacobi across patch faces), out-of-place:
// out = S(u, f). Red = (x+y+z) even. Plane z gets its red update from old black values;
// plane z-1 then gets its black update from new red values, so output lags one plane.
template <int N>
__global__ __launch_bounds__(Tile<N>::TPB) void k_rbgs3d(LevelDev L, const double *__restrict__ u,
                                                         const double *__restrict__ f,
                                                         double *__restrict__ out)
{
	using T           = Tile<N>;
	constexpr int TPB = T::TPB;
	constexpr int CPT = T::CPT;
	constexpr int LW  = T::LW;
	constexpr int NN  = N * N;
	constexpr int NNN = N * N * N;
	const int     pid = xcdRemap(blockIdx.x, L.P);
	if (pid >= L.P) return;
	const int tid = threadIdx.x;

	__shared__ double tile[3][T::LSZ]; // planes z-1, z, z+1 rotate through three buffers
	__shared__ double kfac[9]; // per axis: k at lo face, interior, hi face (physical faces change the diagonal)

	const int32_t *fk = L.face_kind + (size_t) pid * 6;
	const int32_t *fs = L.face_src + (size_t) pid * 6;
	const double   rhx = L.rh2[(size_t) pid * 3], rhy = L.rh2[(size_t) pid * 3 + 1],
	             rhz = L.rh2[(size_t) pid * 3 + 2];
	const double *up = u + (size_t) pid * NNN;
	const double *fp = f + (size_t) pid * NNN;
	double       *op = out + (size_t) pid * NNN;

	if (tid < 9) {
		int    ax = tid / 3, c = tid % 3;
		double k  = 2.0;
		if (c != 1) {
			int kind = fk[2 * ax + (c == 2)];
			if (kind == FACE_DIRICHLET) k = 3.0;
			if (kind == FACE_NEUMANN) k = 1.0;
		}
		kfac[tid] = k;
	}

	const bool has_halo = tid < 4 * N;
	int        h_kind = 0, h_src = 0, h_nbr = 0, h_lds = 0, h_t = 0;
	if (has_halo) {
		const int side = tid / N, t = tid % N;
		h_kind = fk[side];
		h_src  = fs[side];
		h_t    = t;
		switch (side) {
			case 0:
				h_nbr = t * N + (N - 1);
				h_lds = (t + 1) * LW;
				break;
			case 1:
				h_nbr = t * N;
				h_lds = (t + 1) * LW + N + 1;
				break;
			case 2:
				h_nbr = (N - 1) * N + t;
				h_lds = t + 1;
				break;
			default:
				h_nbr = t;
				h_lds = (N + 1) * LW + t + 1;
				break;
		}
	}
	// physical faces contribute nothing to the off-diagonal sum (their ghost is folded into k)
	auto haloLoad = [&](int z) -> double {
		if (h_kind == FACE_LOCAL) return u[(size_t) h_src * NNN + z * NN + h_nbr];
		if (h_kind == FACE_GHOST) return L.ghost[(size_t) h_src * NN + h_t + N * z];
		return 0.0;
	};
	const int kb = fk[4], sb = fs[4], kt = fk[5], st = fs[5];
	auto zGhost = [&](int kind, int src, int nbr_cell, int i) -> double {
		if (kind == FACE_LOCAL) return u[(size_t) src * NNN + nbr_cell];
		if (kind == FACE_GHOST) return L.ghost[(size_t) src * NN + i];
		return 0.0;
	};

	// planes: umm = z-2, um = z-1, uc = z, un = z+1, un2 = z+2 (values are updated in place)
	double umm[CPT], um[CPT], uc[CPT], un[CPT], un2[CPT], fm[CPT], fc[CPT], fn[CPT];
#pragma unroll
	for (int k = 0; k < CPT; k++) {
		const int i = tid + k * TPB;
		uc[k]       = up[i];
		um[k]       = zGhost(kb, sb, (N - 1) * NN + i, i);
		un[k]       = up[NN + i];
		fc[k]       = fp[i];
		umm[k]      = 0.0;
		fm[k]       = 0.0;
	}
	double hv = has_halo ? haloLoad(0) : 0.0;
	__syncthreads(); // kfac

	auto relax = [&](double *tl, int z, int colour, double *cen, const double *below, const double *above,
	                 const double *rhs) {
		const int cz = (z == 0) ? 0 : (z == N - 1 ? 2 : 1);
#pragma unroll
		for (int k = 0; k < CPT; k++) {
			const int i = tid + k * TPB;
			const int x = i % N, y = i / N;
			if (((x + y + z) & 1) != colour) continue;
			const int     cx = (x == 0) ? 0 : (x == N - 1 ? 2 : 1);
			const int     cy = (y == 0) ? 0 : (y == N - 1 ? 2 : 1);
			const double *t0 = tl + (y + 1) * LW + x + 1;
			double        o  = (t0[-1] + t0[1]) * rhx + (t0[-LW] + t0[LW]) * rhy + (below[k] + above[k]) * rhz;
			double        d  = kfac[cx] * rhx + kfac[3 + cy] * rhy + kfac[6 + cz] * rhz;
			double        v  = (o - rhs[k]) / d;
			cen[k]           = v;
			tl[(y + 1) * LW + x + 1] = v;
		}
	};

#pragma unroll 1
	for (int z = 0; z <= N; z++) {
		double hvn = 0.0;
		if (z < N) {
#pragma unroll
			for (int k = 0; k < CPT; k++) {
				const int i = tid + k * TPB;
				if (z + 2 < N)
					un2[k] = up[(z + 2) * NN + i];
				else if (z + 2 == N)
					un2[k] = zGhost(kt, st, i, i);
				else
					un2[k] = 0.0;
				fn[k] = (z + 1 < N) ? fp[(z + 1) * NN + i] : 0.0;
			}
			hvn        = (has_halo && z + 1 < N) ? haloLoad(z + 1) : 0.0;
			double *tl = tile[z % 3];
#pragma unroll
			for (int k = 0; k < CPT; k++) {
				const int i = tid + k * TPB;
				tl[(i / N + 1) * LW + i % N + 1] = uc[k];
			}
			if (has_halo) tl[h_lds] = hv;
		}
		// one barrier per plane: buffer z%3 was last read two iterations ago (black of plane z-3)
		__syncthreads();
		if (z < N) relax(tile[z % 3], z, 0, uc, um, un, fc); // red cells of plane z from old black values
		if (z > 0) {
			double *tl = tile[(z - 1) % 3];
			// black cells of plane z-1: x/y neighbours = new red in LDS; z neighbours = umm (new red,
			// or the frozen bottom ghost) and uc (new red, or the frozen top ghost when z == N)
			relax(tl, z - 1, 1, um, umm, uc, fm);
#pragma unroll
			for (int k = 0; k < CPT; k++) op[(z - 1) * NN + tid + k * TPB] = um[k];
		}
#pragma unroll
		for (int k = 0; k < CPT; k++) {
			umm[k] = um[k];
			um[k]  = uc[k];
			uc[k]  = un[k];
			un[k]  = un2[k];
			fm[k]  = fc[k];
			fc[k]  = fn[k];
		}
		hv = hvn;
	}
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
			double    C  = u[(size_t) d[4] * NNN + oth + ca * sa + cb * sb];
			gamma        = (11 * m - sum) / 12.0 + 4.0 * C / 12.0;
		} else {
			const int     qa = (a >= N / 2), qb = (b >= N / 2);
			const double *fn = u + (size_t) d[4 + qa + 2 * qb] * NNN;
			const int     fa = 2 * (a - qa * (N / 2)), fb = 2 * (b - qb * (N / 2));
			double        sum = 0;
			for (int bb = 0; bb < 2; bb++)
				for (int aa = 0; aa < 2; aa++) sum += 1.0 / 6.0 * fn[oth + (fa + aa) * sa + (fb + bb) * sb];
			gamma = 2.0 / 6.0 * m + sum;
		}
		g[i] = 2 * gamma - m;
	}
}

// AvgRstr.h:78-113, gathered per coarse cell (no atomics, no zero-fill pass): the eight fine
// cells are summed in the order the reference's scatter loop visits them (x, then y, then z),
// each divided by 2^D first, so the result is bit-identical.
// child[pc*8 + o] = fine patch holding orthant o, or child[pc*8] = source, copy[pc] = 1.
template <int N>
__global__ __launch_bounds__(256) void k_restrict3d(int Pc, const int32_t *__restrict__ child,
                                                    const int32_t *__restrict__ copy,
                                                    const double *__restrict__ fine,
                                                    double *__restrict__ coarse)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	const size_t  total = (size_t) Pc * NNN;
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total;
	     idx += (size_t) gridDim.x * blockDim.x) {
		const int pc = (int) (idx / NNN), c = (int) (idx % NNN);
		const int x = c % N, y = (c / N) % N, z = c / NN;
		if (copy[pc]) {
			coarse[idx] = 0.0 + fine[(size_t) child[(size_t) pc * 8] * NNN + c];
			continue;
		}
		const int     ox = x >= H, oy = y >= H, oz = z >= H;
		const double *fp = fine + (size_t) child[(size_t) pc * 8 + ox + 2 * oy + 4 * oz] * NNN;
		const int     fx = 2 * (x - ox * H), fy = 2 * (y - oy * H), fz = 2 * (z - oz * H);
		double        acc = 0.0;
#pragma unroll
		for (int dz = 0; dz < 2; dz++)
#pragma unroll
			for (int dy = 0; dy < 2; dy++) {
				const double2 v = *reinterpret_cast<const double2 *>(fp + fx + N * (fy + dy) + NN * (fz + dz));
				acc += v.x / 8;
				acc += v.y / 8;
			}
		coarse[idx] = acc;
	}
}

// DrctIntp.h:80-113: fine += coarse[parent][(c + orthant offset)/2]
template <int N>
__global__ __launch_bounds__(256) void k_prolong3d(int Pf, const int32_t *__restrict__ parent,
                                                   const int32_t *__restrict__ orth,
                                                   const double *__restrict__ coarse,
                                                   double *__restrict__ fine)
{
	constexpr int NN = N * N, NNN = N * N * N;
	const size_t  total = (size_t) Pf * (NNN / 2);
	for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < total;
	     idx += (size_t) gridDim.x * blockDim.x) {
		const int     pf = (int) (idx / (NNN / 2)), c = (int) (idx % (NNN / 2)) * 2;
		const int     x = c % N, y = (c / N) % N, z = c / NN;
		const int     o  = orth[pf];
		const double *cp = coarse + (size_t) parent[pf] * NNN;
		double2      *fp = reinterpret_cast<double2 *>(fine + (size_t) pf * NNN + c);
		double2       v  = *fp;
		if (o >= 0) {
			const int cx = (x + ((o & 1) ? N : 0)) / 2, cy = (y + ((o & 2) ? N : 0)) / 2,
			          cz = (z + ((o & 4) ? N : 0)) / 2;
			const double cv = cp[cx + N * cy + NN * cz];
			v.x += cv;
			v.y += cv;
		} else {
			v.x += cp[c];
			v.y += cp[c + 1];
		}
		*fp = v;
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
} // namespace te
