// Single-pass patch solve with half-size transforms, for patches whose three axes are "pure": DST-II/III
// (interface or Dirichlet on both sides) or DCT-II/III (Neumann on both sides). Those 32x32 matrices are
// (anti)symmetric under n -> 31-n, F[k][31-n] = (-1)^k F[k][n] (same for the inverse in its row index),
// so a 32-point transform is a butterfly plus two 16x16 products:
//     forward  y[2m+p] = sum_{n<16} F[2m+p][n] (x[n] + (-1)^p x[31-n])
//     inverse  x[n], x[31-n] = P[n] +- Q[n],  P = sum_m G[n][2m] y[2m],  Q = sum_m G[n][2m+1] y[2m+1]
// which halves the work on the fp64 matrix cores (MI355X: the same peak as the vector ALUs, 64 cycles per
// v_mfma_f64_16x16x4, so the dense form is bound by them). The butterflies stay inside a lane because
// every tile that feeds one is produced with its upper half in mirrored order (rows 31-r instead of 16+r),
// which costs nothing: row order is set by load addresses or by the previous product's matrix fragment.
// Frequencies come out split by parity (even block, odd block); only the eigenvalue lookup notices.
//
// Data flow (one 8-wave workgroup per CU walks over patches; lane = (j = l & 15, g = l >> 4)):
//   A    wave w owns planes z = w, w+8, w+16, w+24. Per plane: load f in the y transform's A-operand layout
//        (sixteen lanes read 128 contiguous bytes), subtract the interface terms, y forward (data as A
//        operand: x moves to the rows), x forward (B operand). Result rows kx = 2r+parity, cols ky.
//        Even kx go to the LDS image [kx/2][z][ky] (128 KiB = half a patch), odd kx wait in registers.
//   Z    wave w owns the (z, ky) slabs kx/2 = 2w, 2w+1 of the image: z forward, eigenvalue divide, zero mode,
//        z inverse, written back IN PLACE. The same image therefore serves both exchanges: planes write rows
//        (slab, z, :), slabs read (slab, :, :), planes read their rows back.
//   C1   per plane: P = the even-kx half of the x inverse (the parity split of the half transforms is the
//        split of the image), kept in registers; then the odd-kx half goes through the image (Z again) and
//   C2   per plane: Q = odd-kx half, x = P +- Q, y inverse, scale, store.
// xf_out (may be null) receives the x-face columns of the result, which the next interface-term and residual
// kernels read instead of gathering them with stride 256 B.
// The matrix fragments (24 KiB per plan) sit in LDS, so the loop issues no loads but the planes themselves:
// loads return in order, and a table load queued behind a plane from HBM would stall the transforms.
// Only one patch fits a CU, so nothing but this workgroup can hide its own load latency: the first two
// planes of the NEXT patch are requested during the second z stage (registers are free, no stores are
// queued behind them), the other two while the first ones are processed. Interface terms of a plane's ring
// (128 doubles) arrive with two coalesced loads per lane one plane ahead and are dealt out through a
// per-wave LDS strip. Mixed Dirichlet/Neumann axes (type-IV transforms) have no such symmetry: those levels
// use k_ps_fused.
#pragma once
#include "patchsolve32.hpp"

#ifndef PSS_NTL
#define PSS_NTL 1 // ... and non-temporal loads of the right-hand side (re-read one whole sweep later)
#endif
#ifndef PSS_BAR3
#define PSS_BAR3 0 // 1: a barrier between C1 and the odd half's write into the image (not needed: see there)
#endif
#ifndef PSS_BAREND
#define PSS_BAREND 0 // 1: a barrier at the end of every patch (not needed: see there)
#endif
#ifndef PSS_NT
#define PSS_NT 1 // non-temporal stores of the result (nothing re-reads a 1 GiB vector before it has left the caches)
#endif
namespace te
{
// fragment-ordered half matrices, per plan: [transform 6][parity 2][k-step 4][lane 64]
//   0 y fwd  B[k = n = 4q+g][col j]  = Fy[2j+p][4q+g]        3 x inv  B[k = m = 4q+g][col j] = Gx[j][2(4q+g)+p]
//   1 x fwd  A[i = j][k = n = g+4q]  = Fx[2j+p][g+4q]        4 y inv  A[i = j][k = m = g+4q] = Gy[j][2(g+4q)+p]
//   2 z fwd  A[i = j][k = n = 4q+g]  = Fz[2j+p][4q+g]        5 z inv  A[i = j][k = m = g+4q] = Gz[j][2(g+4q)+p]
constexpr int PSS_FRAG      = 6 * 2 * 4 * 64;
constexpr int PSS_SLAB      = 32 * 32;                             // image: [kx/2 16][z 32][ky 32] doubles
constexpr int PSS_RING      = 16 * PSS_SLAB;                       // per-wave ring strips: [wave 8][128]
constexpr int PSS_TAB       = PSS_RING + 8 * 128;                  // the current plan's fragment table
constexpr int PSS_LDS_BYTES = (PSS_TAB + PSS_FRAG) * 8;            // 163840: all of a CU's LDS
constexpr int PSS_INV       = 32 * 32 * 32;                        // one table of reciprocal eigenvalue sums
#ifdef PSF_TIMING
static __device__ long long pss_stamp[8][12];
#define PSS_STAMP(k) do { if (blockIdx.x == 100 && l == 0 && it == (int) (blockIdx.x + 8 * gridDim.x)) pss_stamp[wave][k] = clock64(); } while (0)
#else
#define PSS_STAMP(k)
#endif

// One plane as the A operand of the y transform, A[i <-> x][k <-> y]: lane (j, g) holds rows y = 4q+g ("l") and
// 31-y ("h"), q = 0..3, of columns x = j ("l") and 31-j ("h"). Sixteen lanes read 128 contiguous bytes.
// n / d for well-scaled operands (eigenvalues of the patch operator: no denormals, no overflow): hardware
// reciprocal, two Newton steps, one residual correction -- the sequence of the IEEE division without its
// scaling and fix-up instructions, a third of the issue cycles. The z stages are bound by these divisions
// (32 per lane and slab on the vector ALU) rather than by their 32 MFMAs.
__device__ __forceinline__ double pssDiv(double n, double d)
{
	double x = __builtin_amdgcn_rcp(d);
	double e = __builtin_fma(-d, x, 1.0);
	x        = __builtin_fma(x, e, x);
	e        = __builtin_fma(-d, x, 1.0);
	x        = __builtin_fma(x, e, x);
	const double q = n * x;
	return __builtin_fma(__builtin_fma(-d, q, n), x, q);
}

struct PssPlane {
	double ll[4], lh[4], hl[4], hh[4]; // [row half][column half][q]
};

// list (may be null): the kernel works on patches list[0..P) instead of 0..P (levels where only some patches
// have pure axes; the others take k_ps_fused).
// FACES (opts.fuse = 3, the zero-guess pre-sweep of a V-cycle): only the six face layers of the result are stored, into
// f6_out [patch][6][N*N] (the layout of the RB-GS face layers: W,E at (y + N z), S,N at (x + N z), B,T at (x + N y)) -- the
// residual of a block-Jacobi sweep from zero lives on the faces (interfaceResidRestrict) and the post-sweep overwrites the
// iterate, reading the old one only through its interface terms (k_face_corr3d): 8 + 1.5 B per site instead of 16.
// inv / itab: the reciprocals 1 / -(lx[kx] + ly[ky] + lz[kz]) of the eigenvalue sums, one table of PSS_INV doubles per distinct (plan,
// spacings) of the level, in the order the z stages consume them ([half][slab][kz parity][r][ky parity][lane]: a wave's load is 512
// contiguous bytes), 0 at the zero mode of an all-Neumann patch (FftwPatchSolver.h:197); itab[patch] = the patch's table. The divisions
// they replace ran on the vector ALU -- which on gfx950 is the unit the fp64 matrix instruction executes on (tools/mfma_valu.hip:
// v_fma_f64 issued by one wave of a SIMD and v_mfma_f64 issued by the other take the SUM of their times): 64 divisions of ~ 11
// instructions per lane and patch were a sixth of the kernel's pipe time.
template <bool CORR, bool FACES = false>
__global__ __launch_bounds__(512) void k_ps_sym(int P, const int32_t *__restrict__ plan, const double *__restrict__ frag,
                                                const double *__restrict__ inv, const int32_t *__restrict__ itab,
                                                const double *__restrict__ in,
                                                const double *__restrict__ corr, double *__restrict__ out,
                                                double *__restrict__ xf_out, const int32_t *__restrict__ list,
                                                double *__restrict__ f6_out = nullptr)
{
	constexpr int N = 32, NN = N * N;
	extern __shared__ __attribute__((aligned(16))) double xbuf[];
	// The walk over patches is a loop, and everything derived from the lane id is loop-invariant: left alone
	// the compiler hoists dozens of LDS/global offsets out of the loop and then spills them (scratch
	// reloads sit in the same in-order vmcnt queue as the prefetches and stall on them). So the wave id is
	// made scalar and each phase re-derives its lane coordinates from an opaque copy of the lane id.
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
	struct Lane {
		int l, j, g;
	};
	auto lane = [&]() {
		int v = l;
		asm volatile("" : "+v"(v));
		return Lane{v, v & 15, v >> 4};
	};
	// this wave's planes in phase A's order: a z-face plane (0 for wave 0, 31 for wave 7) comes last so that
	// its whole-plane interface term can be requested a plane ahead like everything else
	const int     zrot  = (wave == 0) ? 1 : 0;
	const bool    zface = (wave == 0 || wave == 7);
	auto          zof   = [&](int i) { return wave + 8 * ((i + zrot) & 3); };
	double       *ring  = xbuf + PSS_RING + wave * 128;
	const double *tab   = xbuf + PSS_TAB;
	auto          frg   = [&](const Lane &q, int tr, int p, int k) { return tab[((tr * 2 + p) * 4 + k) * 64 + q.l]; };

	auto loadPlane = [&](const Lane &q, PssPlane &d, const double *base) { // base: a plane stored [y][x]
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const double *rl = base + (4 * k + q.g) * N, *rh = base + (N - 1 - 4 * k - q.g) * N;
#if PSS_NTL
			d.ll[k] = __builtin_nontemporal_load(rl + q.j), d.lh[k] = __builtin_nontemporal_load(rl + N - 1 - q.j);
			d.hl[k] = __builtin_nontemporal_load(rh + q.j), d.hh[k] = __builtin_nontemporal_load(rh + N - 1 - q.j);
#else
			d.ll[k] = rl[q.j], d.lh[k] = rl[N - 1 - q.j];
			d.hl[k] = rh[q.j], d.hh[k] = rh[N - 1 - q.j];
#endif
		}
	};
	// ring terms of plane z: lane l < 32 holds W[l] and S[l], l >= 32 holds E[l-32] and N[l-32]
	auto loadRing = [&](const Lane &q, double(&c)[2], const double *cr, int z) {
		c[0] = cr[(q.l >> 5) * NN + N * z + (q.l & 31)];
		c[1] = cr[(2 + (q.l >> 5)) * NN + N * z + (q.l & 31)];
	};
	// d -= ring terms, per element in the reference's side order W/E, then S/N (StarPatchOp.h:185-203)
	auto applyRing = [&](const Lane &q, PssPlane &d, const double(&c)[2]) {
		const int    l = q.l, j = q.j, g = q.g, y = l & 31;
		const double m0 = (g == 0) ? 1.0 : 0.0, mj = (j == 0) ? 1.0 : 0.0;
		// W/E strips: the eight values a lane needs (y = 4q+g, then 31-y) adjacent: [g][l q 0..3 | h q 0..3]
		ring[(l >> 5) * 32 + (y < 16 ? (y & 3) * 8 + (y >> 2) : (3 - (y & 3)) * 8 + 4 + (7 - (y >> 2)))] = c[0];
		ring[64 + l] = c[1];
		double wv[8], ev[8];
#pragma unroll
		for (int k = 0; k < 8; k++) wv[k] = ring[g * 8 + k], ev[k] = ring[32 + g * 8 + k];
		const double s0 = ring[64 + j], s1 = ring[64 + 31 - j], n0 = ring[96 + j], n1 = ring[96 + 31 - j];
#pragma unroll
		for (int k = 0; k < 4; k++) {
			d.ll[k] -= mj * wv[k], d.hl[k] -= mj * wv[4 + k]; // x = 0
			d.lh[k] -= mj * ev[k], d.hh[k] -= mj * ev[4 + k]; // x = 31
		}
		d.ll[0] -= m0 * s0, d.lh[0] -= m0 * s1; // y = 0
		d.hl[0] -= m0 * n0, d.hh[0] -= m0 * n1; // y = 31
	};

	// planes in flight: slot 0 = plane z_0 then z_2 (then the z-face term), slot 1 = z_1 then z_3
	PssPlane s0, s1;
	double   c[2] = {0.0, 0.0};
	int      it = blockIdx.x, cur_plan = -1; // position in the walk; pid = the patch there
	auto     patchAt = [&](int i) { return list ? list[i] : i; };
	if (it < P) {
		const Lane q   = lane();
		const int  pid = patchAt(it);
		loadPlane(q, s0, in + ((size_t) pid * N + zof(0)) * NN);
		loadPlane(q, s1, in + ((size_t) pid * N + zof(1)) * NN);
		if (CORR) loadRing(q, c, corr + (size_t) pid * 6 * NN, zof(0));
	}
	// A CU keeps about one L1's worth (32 KiB) of loads in flight, so a burst of plane requests takes several
	// memory latencies and holds up every store queued behind it (the memory pipeline is in order): the next
	// patch's first two planes are requested apart, at the start and at the end of the second z stage.
	auto prefetch = [&](int which) {
		if (it + (int) gridDim.x >= P) return;
		const int  np = patchAt(it + gridDim.x);
		const Lane qn = lane();
		if (which == 0) {
			loadPlane(qn, s0, in + ((size_t) np * N + zof(0)) * NN);
			if (CORR) loadRing(qn, c, corr + (size_t) np * 6 * NN, zof(0));
		} else {
			loadPlane(qn, s1, in + ((size_t) np * N + zof(1)) * NN);
		}
	};

#pragma unroll 1
	for (; it < P; it += gridDim.x) {
		const int pid = patchAt(it);
		const int pl = plan[pid];
		if (pl != cur_plan) { // (re)load the plan's fragment table: only when every wave is past the previous patch
			if (cur_plan >= 0) ldsBarrier();
			const double *src = frag + (size_t) pl * PSS_FRAG;
			for (int i = threadIdx.x; i < PSS_FRAG; i += 512) xbuf[PSS_TAB + i] = src[i];
			cur_plan = pl;
			ldsBarrier();
		}
		const double *ip = in + (size_t) pid * N * NN;
		const double *cr = CORR ? corr + (size_t) pid * 6 * NN : nullptr;
		const double *ivt = inv + (size_t) itab[pid] * PSS_INV;
		double        iv[2][2][4];
		auto          ivLoad = [&](int half, int t) {
            const Lane    q   = lane();
            const double *ip2 = ivt + ((size_t) (half * 16 + 2 * wave + t) * 16) * 64 + q.l;
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    iv[p][0][r] = ip2[((p * 4 + r) * 2 + 0) * 64];
                    iv[p][1][r] = ip2[((p * 4 + r) * 2 + 1) * 64];
                }
		};
		PSS_STAMP(0);

		// ---- A: y,x forward of this wave's four planes ---------------------------------------------
		v4f64 hi[4][2]; // odd-kx half of the transformed planes, parked until the image is free again
		{
			const Lane q = lane();
			const int  j = q.j, g = q.g;
			v4f64      t[2][2]; // [x half][ky parity]: rows x = r (half 0) / 31 - r (half 1), r = g + 4r'; cols ky = 2j + parity
			auto       yfwd = [&](PssPlane &a) {
                t[0][0] = t[0][1] = t[1][0] = t[1][1] = v4f64{0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double b0 = frg(q, 0, 0, k), b1 = frg(q, 0, 1, k);
                    t[0][0] = mfma_f64(a.ll[k] + a.hl[k], b0, t[0][0]);
                    t[0][1] = mfma_f64(a.ll[k] - a.hl[k], b1, t[0][1]);
                    t[1][0] = mfma_f64(a.lh[k] + a.hh[k], b0, t[1][0]);
                    t[1][1] = mfma_f64(a.lh[k] - a.hh[k], b1, t[1][1]);
                }
			};
			auto xfwd = [&](int z, v4f64(&h)[2]) {
				v4f64       e[2][2]; // [kx parity][ky parity]: rows kx = 2 (g + 4r) + parity
				const v4f64 sx0 = t[0][0] + t[1][0], dx0 = t[0][0] - t[1][0], sx1 = t[0][1] + t[1][1], dx1 = t[0][1] - t[1][1];
				e[0][0] = e[0][1] = e[1][0] = e[1][1] = v4f64{0, 0, 0, 0};
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const double a0 = frg(q, 1, 0, r), a1 = frg(q, 1, 1, r);
					e[0][0] = mfma_f64(a0, sx0[r], e[0][0]);
					e[0][1] = mfma_f64(a0, sx1[r], e[0][1]);
					e[1][0] = mfma_f64(a1, dx0[r], e[1][0]);
					e[1][1] = mfma_f64(a1, dx1[r], e[1][1]);
				}
				// even kx -> the image now, odd kx wait in h
#pragma unroll
				for (int r = 0; r < 4; r++)
					*reinterpret_cast<double2 *>(xbuf + (g + 4 * r) * PSS_SLAB + z * N + 2 * j) = double2{e[0][0][r], e[0][1][r]};
				h[0] = e[1][0], h[1] = e[1][1];
			};
			double cn[2];
			// A plane's first instructions -- the butterflies of its y transform -- depend on nothing but its loads, and the instruction
			// scheduler likes to hoist them to right behind the request (seen in the ISA: s_waitcnt vmcnt(5) five instructions after
			// plane z_3 was requested -- a whole HBM latency exposed per patch). Nothing crosses a fence: a plane is consumed where
			// the source says, a plane's worth of matrix instructions behind its request.
#ifndef PSS_FENCE_MASK
#define PSS_FENCE_MASK 0x3FC // every class of instruction may cross but vector-ALU ones (a full fence, 0, costs registers: spills)
#endif
#ifndef PSS_FENCES
#define PSS_FENCES 7
#endif
#define PSS_FENCE(i) do { if (PSS_FENCES & (1 << (i))) __builtin_amdgcn_sched_barrier(PSS_FENCE_MASK); } while (0)
			// plane z_0 (slot 0), then request z_2 -> slot 0
			if (CORR) {
				loadRing(q, cn, cr, zof(1));
				applyRing(q, s0, c);
			}
			yfwd(s0);
			loadPlane(q, s0, ip + zof(2) * NN);
			xfwd(zof(0), hi[0]);
			PSS_FENCE(0);
			// plane z_1 (slot 1), then request z_3 -> slot 1
			if (CORR) {
				loadRing(q, c, cr, zof(2));
				applyRing(q, s1, cn);
			}
			yfwd(s1);
			loadPlane(q, s1, ip + zof(3) * NN);
			xfwd(zof(1), hi[1]);
			PSS_FENCE(1);
			// plane z_2 (slot 0), then the z-face term of z_3 -> slot 0
			if (CORR) {
				loadRing(q, cn, cr, zof(3));
				applyRing(q, s0, c);
			}
			yfwd(s0);
			if (CORR && zface) loadPlane(q, s0, cr + (wave == 0 ? 4 : 5) * NN);
			xfwd(zof(2), hi[2]);
			PSS_FENCE(2);
			// plane z_3 (slot 1)
			if (CORR) {
				applyRing(q, s1, cn);
				if (zface) {
#pragma unroll
					for (int k = 0; k < 4; k++)
						s1.ll[k] -= s0.ll[k], s1.lh[k] -= s0.lh[k], s1.hl[k] -= s0.hl[k], s1.hh[k] -= s0.hh[k];
				}
			}
			ivLoad(0, 0); // (the first z stage's first reciprocals: this plane's 32 matrix instructions ahead of their use)
			yfwd(s1);
			xfwd(zof(3), hi[3]);
		}
		PSS_STAMP(1);

		// ---- Z on one half of the image (kx = 2 slab + half), in place ----------------------------------
		auto zstage = [&](int half, bool fetch_next) {
			const Lane    q  = lane();
			const int     j = q.j, g = q.g;
			// The reciprocals of a slab (kx = 2 (2 wave + t) + half), iv[kz parity][ky parity][r], come from L2 (every CU walks the same
			// few tables): slab 0's were requested by the phase in front of this stage (ivLoad: a transform's worth of matrix
			// instructions earlier), slab 1's go into the same registers as soon as slab 0's have been used -- slab 0's inverse
			// and slab 1's forward transform (32 matrix instructions) lie between that request and its use. Slab by slab
			// rather than stage by stage: one slab's data and one set of reciprocals are live at a time (the kernel runs at 256
			// registers), and with a multiplication where the division was there is no vector work left to hide.
#pragma unroll
			for (int t = 0; t < 2; t++) {
				double *sp = xbuf + (2 * wave + t) * PSS_SLAB + 2 * j;
				v4f64   d[2][2]; // [kz parity][ky parity]: rows kz = 2 (g + 4r) + parity
				d[0][0] = d[0][1] = d[1][0] = d[1][1] = v4f64{0, 0, 0, 0};
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const double2 vl = *reinterpret_cast<const double2 *>(sp + (4 * k + g) * N);
					const double2 vh = *reinterpret_cast<const double2 *>(sp + (N - 1 - 4 * k - g) * N);
					const double  a0 = frg(q, 2, 0, k), a1 = frg(q, 2, 1, k);
					d[0][0] = mfma_f64(a0, vl.x + vh.x, d[0][0]);
					d[0][1] = mfma_f64(a0, vl.y + vh.y, d[0][1]);
					d[1][0] = mfma_f64(a1, vl.x - vh.x, d[1][0]);
					d[1][1] = mfma_f64(a1, vl.y - vh.y, d[1][1]);
				}
#pragma unroll
				for (int p = 0; p < 2; p++)
#pragma unroll
					for (int r = 0; r < 4; r++) {
						d[p][0][r] *= iv[p][0][r];
						d[p][1][r] *= iv[p][1][r];
					}
				if (t == 0) {
					ivLoad(half, 1);
					if (fetch_next) prefetch(0); // (behind the table request: loads return in order)
				}
				v4f64 pz[2], qz[2];
				pz[0] = pz[1] = qz[0] = qz[1] = v4f64{0, 0, 0, 0};
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const double a0 = frg(q, 5, 0, r), a1 = frg(q, 5, 1, r);
					pz[0] = mfma_f64(a0, d[0][0][r], pz[0]);
					pz[1] = mfma_f64(a0, d[0][1][r], pz[1]);
					qz[0] = mfma_f64(a1, d[1][0][r], qz[0]);
					qz[1] = mfma_f64(a1, d[1][1][r], qz[1]);
				}
				// rows z = g + 4r (P + Q) and 31 - z (P - Q); every read of this slab is complete (same wave, in order)
#pragma unroll
				for (int r = 0; r < 4; r++) {
					*reinterpret_cast<double2 *>(sp + (g + 4 * r) * N)         = double2{pz[0][r] + qz[0][r], pz[1][r] + qz[1][r]};
					*reinterpret_cast<double2 *>(sp + (N - 1 - g - 4 * r) * N) = double2{pz[0][r] - qz[0][r], pz[1][r] - qz[1][r]};
				}
			}
			if (fetch_next) prefetch(1);
		};
		// half of the x inverse of plane z from the image: A[i = j <-> ky = 2j (.x), 2j+1 (.y)][k = m = 4q+g <-> slab]
		auto xhalf = [&](const Lane &q, int z, int half, v4f64(&acc)[2]) {
			const double *rp = xbuf + q.g * PSS_SLAB + z * N + 2 * q.j;
			acc[0] = acc[1] = v4f64{0, 0, 0, 0};
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const double2 v = *reinterpret_cast<const double2 *>(rp + 4 * k * PSS_SLAB);
				const double  b = frg(q, 3, half, k);
				acc[0] = mfma_f64(v.x, b, acc[0]);
				acc[1] = mfma_f64(v.y, b, acc[1]);
			}
		};

		ldsBarrier();
		PSS_STAMP(2);
		zstage(0, false);
		PSS_STAMP(3);
		ldsBarrier();
		PSS_STAMP(4);
		v4f64 Pe[4][2]; // even-kx half of the x inverse of planes z = wave + 8i: rows ky = 2 (g + 4r) + parity, cols x
		{
			const Lane q = lane();
			ivLoad(1, 0); // (the second z stage's first reciprocals)
#pragma unroll
			for (int i = 0; i < 4; i++) xhalf(q, wave + 8 * i, 0, Pe[i]);
		}
		PSS_STAMP(5);
		// The odd half takes the even half's place WITHOUT a barrier: the rows a wave overwrites here -- (slab, z) for its own four
		// planes z, all slabs -- are exactly the rows it alone has just read in C1 (same wave, LDS operations in program order).
#if PSS_BAR3
		ldsBarrier();
#endif
		{
			const Lane q = lane();
#pragma unroll
			for (int i = 0; i < 4; i++)
#pragma unroll
				for (int r = 0; r < 4; r++)
					*reinterpret_cast<double2 *>(xbuf + (q.g + 4 * r) * PSS_SLAB + zof(i) * N + 2 * q.j) = double2{hi[i][0][r], hi[i][1][r]};
		}
		ldsBarrier();
		PSS_STAMP(6);
		zstage(1, true);
		PSS_STAMP(7);
		ldsBarrier();
		PSS_STAMP(8);

		// ---- C2: odd half of the x inverse, y inverse, scale, store --------------------------------------
		{
			const Lane       q = lane();
			const int        j = q.j, g = q.g;
			// (the scale (2/N)^3 = 2^-12 of DftPatchSolver.h:214 rides in the y-inverse fragments: a power of two, bit-identical)
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int z = wave + 8 * i;
				v4f64     Qo[2];
				xhalf(q, z, 1, Qo);
				double *op = out + ((size_t) pid * N + z) * NN;
#pragma unroll
				for (int xc = 0; xc < 2; xc++) { // columns x = j (P + Q) and 31 - j (P - Q)
					const v4f64 x0 = xc ? Pe[i][0] - Qo[0] : Pe[i][0] + Qo[0]; // rows ky = 2 (g + 4r)
					const v4f64 x1 = xc ? Pe[i][1] - Qo[1] : Pe[i][1] + Qo[1]; // rows ky = 2 (g + 4r) + 1
					v4f64       py = v4f64{0, 0, 0, 0}, qy = v4f64{0, 0, 0, 0};
#pragma unroll
					for (int r = 0; r < 4; r++) {
						py = mfma_f64(frg(q, 4, 0, r), x0[r], py);
						qy = mfma_f64(frg(q, 4, 1, r), x1[r], qy);
					}
					const int   x  = xc ? N - 1 - j : j;
					const v4f64 yl = py + qy, yh = py - qy;
					if (FACES) {
						double *fb = f6_out + (size_t) pid * 6 * NN;
						if (z == 0 || z == N - 1) { // a z face: the whole plane
							double *zp = fb + (z ? 5 : 4) * NN;
#pragma unroll
							for (int r = 0; r < 4; r++) zp[(g + 4 * r) * N + x] = yl[r], zp[(N - 1 - g - 4 * r) * N + x] = yh[r];
						}
						if (g == 0) fb[2 * NN + x + N * z] = yl[0], fb[3 * NN + x + N * z] = yh[0]; // rows y = 0 and y = N - 1
						if (j == 0) { // columns x = 0 (xc = 0) and x = N - 1
							double *xo = fb + xc * NN + N * z;
#pragma unroll
							for (int r = 0; r < 4; r++) xo[g + 4 * r] = yl[r], xo[N - 1 - g - 4 * r] = yh[r];
						}
						continue;
					}
#pragma unroll
					for (int r = 0; r < 4; r++) {
#if PSS_NT
						__builtin_nontemporal_store(yl[r], &op[(g + 4 * r) * N + x]);
						__builtin_nontemporal_store(yh[r], &op[(N - 1 - g - 4 * r) * N + x]);
#else
						op[(g + 4 * r) * N + x]         = yl[r];
						op[(N - 1 - g - 4 * r) * N + x] = yh[r];
#endif
					}
					if (xf_out && j == 0) { // the two x-face columns of the new iterate, compact (LevelDev.xf layout: [side][z][y])
						double *xo = xf_out + ((size_t) pid * 2 + xc) * NN + N * z;
#pragma unroll
						for (int r = 0; r < 4; r++) xo[g + 4 * r] = yl[r], xo[N - 1 - g - 4 * r] = yh[r];
					}
				}
			}
		}
		PSS_STAMP(9);
		// No barrier at the end of a patch either: the next patch's phase A rewrites rows of this wave's own planes, which only
		// this wave read in C2; every other wave meets the new data behind the barrier in front of the first z stage. (A change of
		// plan has its own barrier: the fragment table is shared. That reload must not overtake a wave still in C2 -- see there.)
#if PSS_BAREND
		ldsBarrier();
#endif
	}
}
} // namespace te
