// The plane-marching stencil kernels of the 3D path (operator apply, residual, weighted Jacobi, fused
// residual+restriction, and the patch-local red-black Gauss-Seidel sweep with its zero-guess and
// fused-prolongation variants). See kernels3d.hpp for layout, ghost formulation and helpers.
//
// Thread map (Tile3): thread (X, Yp) owns the x-pair X (cells 2X, 2X+1) of the TWO adjacent rows
// y = 2Yp and 2Yp+1 of every plane. Consequences that keep the instruction count low (these kernels
// are co-limited by VALU/LDS issue, not only by HBM):
//   * the y-neighbour inside the row pair is a register of the same thread; only the outer one is an
//     LDS read (1 ds_read_b128 per pair for the stencil modes, 1 ds_read_b64 per relaxed cell);
//   * row parity is the compile-time index k, so with the z loop unrolled by two the colour of every
//     cell is static: the red-black sweep has no selects and no run-time LDS addressing;
//   * the 2x2x2 children of a coarse cell sit in one thread (two planes apart in time), so the fused
//     restriction needs no cross-lane traffic.
#pragma once
#include "kernels3d.hpp"
#include <type_traits>

#ifndef TE_ZR_WIDE
#define TE_ZR_WIDE 1 // (0: tooling -- every variant of the fused pre-sweep at three workgroups per CU, as before round 6's end)
#endif
#ifndef TE_ZR_NT
#define TE_ZR_NT 0 // (non-temporal loads of f in the pre-sweep cost more in the post-sweep, which finds less of f in the Infinity Cache)
#endif
namespace te
{
// streaming accesses that nothing re-reads before they leave the caches (a 1 GiB vector per pass): non-temporal
template <bool NT> __device__ __forceinline__ double2 ldStream(const double2 *p)
{
	if (NT) {
		double2 v;
		v.x = __builtin_nontemporal_load(&p->x);
		v.y = __builtin_nontemporal_load(&p->y);
		return v;
	}
	return *p;
}
template <bool NT> __device__ __forceinline__ void stStream(double2 *p, double2 v)
{
	if (NT) {
		__builtin_nontemporal_store(v.x, &p->x);
		__builtin_nontemporal_store(v.y, &p->y);
	} else
		*p = v;
}
// A copy the compiler cannot see through: the source registers are dead afterwards, so the load that refills a ring slot can
// be given the slot's own registers. Without it the register allocator merges the COPY with the slot (the copy is free) and
// gives every refill fresh registers, which it then moves into the slot's registers at the loop's back edge -- a move that reads
// the newest loads and makes every iteration wait for them.
__device__ __forceinline__ double2 takeRegs(const double2 &src)
{
	double2 r;
	asm volatile("v_mov_b64 %0, %2\n\tv_mov_b64 %1, %3" : "=&v"(r.x), "=&v"(r.y) : "v"(src.x), "v"(src.y));
	return r;
}
__device__ __forceinline__ double takeReg(double src)
{
	double r;
	asm volatile("v_mov_b64 %0, %1" : "=&v"(r) : "v"(src));
	return r;
}
// The six face kinds of a patch, requested together: read one by one under conditions (position class != interior) they
// are three dependent waits at the start of every workgroup, and a slab kernel is six plane steps long.
struct FaceKinds {
	int32_t k[6];
	template <class FK> __device__ __forceinline__ explicit FaceKinds(const FK &fk)
	{
#pragma unroll
		for (int s = 0; s < 6; s++) k[s] = fk[s];
	}
	// ghost of the residual's stencil as a multiple of the cell just inside: -1 Dirichlet, +1 Neumann, 0 on a face with a neighbour
	__device__ __forceinline__ double phys(int s) const { return k[s] == FACE_DIRICHLET ? -1.0 : (k[s] == FACE_NEUMANN ? 1.0 : 0.0); }
};
// 1/diag of the patch-local relaxation per (x, y, z) position class (low face, interior, high face): physical faces are folded
// into the diagonal, k = 3 Dirichlet, 1 Neumann (StarPatchOp.h:39-65)
__device__ __forceinline__ void idiagTable(double *idiag, int tid, const FaceKinds &kinds, double rhx, double rhy, double rhz)
{
	if (tid < 27) {
		double    kf[3];
		const int cls[3] = {tid % 3, (tid / 3) % 3, tid / 9};
#pragma unroll
		for (int ax = 0; ax < 3; ax++) {
			const int kind = cls[ax] == 2 ? kinds.k[2 * ax + 1] : kinds.k[2 * ax];
			kf[ax]         = 2.0;
			if (cls[ax] != 1 && kind == FACE_DIRICHLET) kf[ax] = 3.0;
			if (cls[ax] != 1 && kind == FACE_NEUMANN) kf[ax] = 1.0;
		}
		idiag[tid] = 1.0 / (kf[0] * rhx + kf[1] * rhy + kf[2] * rhz);
	}
}
template <int N> struct Tile3 {
	static constexpr int H   = N / 2;
	static constexpr int NT  = H * H;               // threads that own cells
	static constexpr int TPB = NT < 64 ? 64 : NT;   // 256 for N = 32
	static constexpr int NP  = N * N / 2;           // pairs per plane
	static constexpr int LW  = Tile2<N>::LW;
	static constexpr int LSZ = Tile2<N>::LSZ;
	__host__ __device__ static constexpr int row(int r) { return Tile2<N>::row(r); }
	static_assert(4 * N <= TPB, "one halo entry per thread");
};

// MODE_APPLY : out = A u                     (SchurHelper.h:360-376 + StarPatchOp.h:28-184)
// MODE_RESID : out = f - A u                 (+ Cycle.h:60-61)
// MODE_JACOBI: out = u + omega (f - A u)/diag(A)
// MODE_RESID_RESTRICT: coarse f = AvgRstr(f - A u) without ever storing r (Cycle.h:59-65 fused); the
//   eight fine residuals of a coarse cell are added in AvgRstr's own order (x, y, z), so the result
//   is bit-identical to MODE_RESID followed by k_restrict3d.
// grid: 8*ceil(P*ZS/8) blocks of Tile3<N>::TPB threads; ZS z-slabs per patch.
// The steady-state loop is branch-free: every load of an iteration is issued unconditionally from a
// (pointer, sign) pair chosen with scalar selects, so the whole next plane stays in flight behind the
// LDS barrier.
// RED (MODE_APPLY, MODE_RESID): sums over the result that the caller needs next, formed while the values are in registers
// instead of in a pass of their own over a stored vector (Vector.h:284-321 dot / twoNorm; BiCGStab.h:73-87 calls them right
// after the operator application): RED_OUT_A: sum out*a; RED_OUT_A_OUT: sum out*a and sum out*out; RED_OUT_OUT: sum out*out
// (the residual norm, Cycle.h:60-61 + Vector.h:294). Per thread the products are added in plane order, then wave shuffles ->
// LDS -> one pair per workgroup in red.partial[2 (red.base + work item)]: a fixed order whatever the launch geometry; the
// caller's k_reduce_final2 adds the pairs in index order.
enum StencilRed : int { RED_NONE = 0, RED_OUT_A = 1, RED_OUT_A_OUT = 2, RED_OUT_OUT = 3 };
struct RedSrc {
	const double *a;       // second operand of the dot product (same level and layout as out), or null
	double       *partial; // [2 * work items of the level]
	int           base;    // work items of the launches that precede this one on the level (interior before boundary)
};
template <int N, int MODE, int ZS, int RED = RED_NONE>
__global__ __launch_bounds__(Tile3<N>::TPB) void k_stencil3d(LevelDev L, const double *__restrict__ u,
                                                             const double *__restrict__ f,
                                                             double *__restrict__ out, double omega, RestrictDst rd,
                                                             RedSrc red = RedSrc())
{
	static_assert(RED == RED_NONE || MODE == MODE_APPLY || MODE == MODE_RESID, "fused sums exist for apply and residual");
	using T            = Tile3<N>;
	constexpr int TPB  = T::TPB, NP = T::NP, H = T::H;
	constexpr int NN   = N * N, NNN = N * N * N;
	constexpr int ZL   = N / ZS; // planes per slab
	TE_STAMP_DECL;
	TE_STAMP(0, false);
	if (ZS > 1) argsUpFront(L, u, f, out, rd.parent, rd.orth, MODE == MODE_RESID_RESTRICT ? (const void *) rd.remote_off : (const void *) rd.coarse);
	const int nblocks  = L.count * ZS;
	const int work     = xcdRemap(blockIdx.x, nblocks);
	if (work >= nblocks) return;
	const int pid = L.order ? L.order[L.first + work / ZS] : L.first + work / ZS;
	const int z0  = (work % ZS) * ZL;
	const int tid = threadIdx.x;

	__shared__ __attribute__((aligned(16))) double tile[2][T::LSZ];
	__shared__ double idiag[27];

	const Reg6     fk(L.face_kind + (size_t) pid * 6), fs(L.face_src + (size_t) pid * 6);
	const double   rhx = L.rh2[(size_t) pid * 3], rhy = L.rh2[(size_t) pid * 3 + 1], rhz = L.rh2[(size_t) pid * 3 + 2];
	const double  *up  = u + (size_t) pid * NNN;
	const double2 *up2 = reinterpret_cast<const double2 *>(up);
	const double2 *fp2 = reinterpret_cast<const double2 *>((MODE != MODE_APPLY ? f : u) + (size_t) pid * NNN);
	double2       *op2 = reinterpret_cast<double2 *>(out + (size_t) pid * NNN);

	const bool act = (T::NT == TPB) || tid < T::NT;
	const int  X = act ? tid % H : 0, Yp = act ? tid / H : 0;
	int        q[2], lds[2];
	const int  ldo[2] = {T::row(2 * Yp) + 2 * X + 2, T::row(2 * Yp + 3) + 2 * X + 2}; // the rows below / above the pair
#pragma unroll
	for (int k = 0; k < 2; k++) {
		q[k]   = (2 * Yp + k) * H + X;
		lds[k] = T::row(2 * Yp + k + 1) + 2 * X + 2;
	}

	// fused restriction target
	double *rdst = nullptr; // this thread's coarse cell of plane pair 0 (octant of the parent, or a remote block)
	int     rsz = 0, rorth = -1;
	double  racc = 0.0;
	if (MODE == MODE_RESID_RESTRICT) {
		const int pa = rd.parent[pid];
		rorth        = rd.orth[pid];
		if (rorth < 0) { // copy-through: r lands unchanged in the coarse patch / remote block
			op2 = reinterpret_cast<double2 *>(pa >= 0 ? rd.coarse + (size_t) pa * NNN : rd.remote + rd.remote_off[-(pa + 2)]);
		} else if (pa >= 0) {
			rdst = rd.coarse + (size_t) pa * NNN + ((rorth & 1) ? H : 0) + N * ((rorth & 2) ? H : 0) + NN * ((rorth & 4) ? H : 0) + X + N * Yp;
			rsz  = NN;
		} else {
			rdst = rd.remote + rd.remote_off[-(pa + 2)] + X + H * Yp;
			rsz  = H * H;
		}
	}

	int dix[2][2] = {{0, 0}, {0, 0}}; // cx + 3 cy of (row k, cell)
	if (MODE == MODE_JACOBI) {
		const double *kp = L.face_kadj + (size_t) pid * 6;
		double        ka[6]; // (all six together: one wait instead of three)
#pragma unroll
		for (int s6 = 0; s6 < 6; s6++) ka[s6] = kp[s6];
		if (tid < 27) {
			int    cx = tid % 3, cy = (tid / 3) % 3, cz = tid / 9;
			double kx = 2.0 + (cx == 0 ? ka[0] : 0.0) + (cx == 2 ? ka[1] : 0.0);
			double ky = 2.0 + (cy == 0 ? ka[2] : 0.0) + (cy == 2 ? ka[3] : 0.0);
			double kz = 2.0 + (cz == 0 ? ka[4] : 0.0) + (cz == 2 ? ka[5] : 0.0);
			idiag[tid] = -1.0 / (kx * rhx + ky * rhy + kz * rhz);
		}
		const int cy0 = (Yp == 0) ? 0 : 1, cy1 = (Yp == H - 1) ? 2 : 1;
		const int cx0 = (X == 0) ? 0 : 1, cx1 = (X == H - 1) ? 2 : 1;
		dix[0][0] = cx0 + 3 * cy0, dix[0][1] = cx1 + 3 * cy0, dix[1][0] = cx0 + 3 * cy1, dix[1][1] = cx1 + 3 * cy1;
	}

	const HaloSrc  hs  = haloSrc<N>(tid, fk, fs, u, up, L.ghost, -1.0, 1.0, L.xf);
	const PlaneSrc bot = zPlaneSrc<N>(fk[4], fs[4], false, u, up, L.ghost, -1.0, 1.0);
	const PlaneSrc top = zPlaneSrc<N>(fk[5], fs[5], true, u, up, L.ghost, -1.0, 1.0);
	TE_STAMP(1, true);

	// ---- register pipeline over z ------------------------------------------------------------
	// um, uc, un = planes z-1, z, z+1 of u (ghost planes scaled). Planes z+2, z+3 of u, z+1, z+2 of the right-hand side (and of
	// the fused sums' second operand) are in flight in two-slot rings, plane p in slot p & 1, raw: a slot is taken (takeRegs,
	// see k_rbgs_zero_resid3d) by the step that first needs its plane and requested again in place, two steps before its next
	// use; nothing a step requests is touched by the same step (a load consumed in the step that issues it makes the step wait
	// for it and, loads returning in order, for everything requested before it). The halo value of the next plane is requested
	// first in every step: it is what the next step waits for.
	constexpr bool HAS_F = MODE != MODE_APPLY, HAS_A = RED == RED_OUT_A || RED == RED_OUT_A_OUT;
	const double2 *ap2 = HAS_A ? reinterpret_cast<const double2 *>(red.a + (size_t) pid * NNN) : nullptr;
	double2        um[2], uc[2], un[2], fc[2], ac[2];
	double2        ur[2][2], fr[2][2], ar[2][2];
	double         acc0 = 0.0, acc1 = 0.0;
	auto uPlane = [&](int p) { return (p < 0) ? bot.p : (p < N ? up2 + p * NP : top.p); };   // p = z0 - 1 .. N
	auto uScale = [&](int p) { return (p < 0) ? bot.s : (p < N ? 1.0 : top.s); };
	auto clampP = [&](int p) { return p < N ? p : N - 1; };
	// state before step 0 shifts: uc = plane z0-1 (-> um), un = plane z0 (-> uc); the rings hold planes z0+1, z0+2 of u and
	// z0, z0+1 of the right-hand side
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const double2 a = uPlane(z0 - 1)[q[k]];
		const double  sm = uScale(z0 - 1);
		uc[k] = double2{sm * a.x, sm * a.y};
		un[k] = up2[z0 * NP + q[k]];
	}
	double hraw = hs.p[z0 * hs.stride];
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int i = 0; i < 2; i++) { // oldest first: the loop's waits assume the order of its own requests
#pragma unroll
		for (int k = 0; k < 2; k++) {
			ur[(i + 1) & 1][k] = uPlane(z0 + 1 + i)[q[k]];
			if (HAS_F) fr[i][k] = fp2[clampP(z0 + i) * NP + q[k]];
			if (HAS_A) ar[i][k] = ap2[clampP(z0 + i) * NP + q[k]];
		}
		__builtin_amdgcn_sched_barrier(0);
	}

	// REFILL = false: the last two steps of a slab, which request nothing (their planes would never be used)
	auto step = [&](auto par, auto refill, int zz) {
		constexpr int  PAR = decltype(par)::value; // zz & 1
		constexpr bool REFILL = decltype(refill)::value;
		const int      z = z0 + zz;
		const double   hv = hs.s * takeReg(hraw);
		if (REFILL || PAR == 0) hraw = hs.p[clampP(z + 1) * hs.stride]; // (the last step has no successor)
		__builtin_amdgcn_sched_barrier(0);
		const double sn = uScale(z + 1);
#pragma unroll
		for (int k = 0; k < 2; k++) {
			um[k] = uc[k];
			uc[k] = un[k];
			const double2 a = takeRegs(ur[1 - PAR][k]); // plane z+1
			un[k]           = double2{sn * a.x, sn * a.y};
			if (HAS_F) fc[k] = takeRegs(fr[PAR][k]);
			if (HAS_A) ac[k] = takeRegs(ar[PAR][k]);
			if (REFILL) {
				ur[1 - PAR][k] = uPlane(z + 3)[q[k]];
				if (HAS_F) fr[PAR][k] = fp2[clampP(z + 2) * NP + q[k]];
				if (HAS_A) ar[PAR][k] = ap2[clampP(z + 2) * NP + q[k]];
			}
		}

		double *tl = tile[PAR];
		if (act) {
			ldsStore2(tl + lds[0], uc[0]);
			ldsStore2(tl + lds[1], uc[1]);
		}
		if (hs.lds >= 0) tl[hs.lds] = hv;
		ldsBarrier();

		// outer y-neighbours from LDS, inner ones are the other row's registers
		const double2 ylo = ldsLoad2(tl + ldo[0]);
		const double2 yhi = ldsLoad2(tl + ldo[1]);
		double2       r[2];
#pragma unroll
		for (int k = 0; k < 2; k++) {
			const double *t0 = tl + lds[k];
			const double2 c  = uc[k];
			const double2 ym = (k == 0) ? ylo : uc[0];
			const double2 yp = (k == 0) ? uc[1] : yhi;
			const double  xl = t0[-1], xr = t0[2];
			double2       lap;
			lap.x = (xl - 2 * c.x + c.y) * rhx;
			lap.y = (c.x - 2 * c.y + xr) * rhx;
			lap.x += (ym.x - 2 * c.x + yp.x) * rhy;
			lap.y += (ym.y - 2 * c.y + yp.y) * rhy;
			lap.x += (um[k].x - 2 * c.x + un[k].x) * rhz;
			lap.y += (um[k].y - 2 * c.y + un[k].y) * rhz;
			if (MODE == MODE_APPLY) {
				r[k] = lap;
			} else if (MODE == MODE_RESID || MODE == MODE_RESID_RESTRICT) {
				r[k].x = fc[k].x - lap.x;
				r[k].y = fc[k].y - lap.y;
			} else {
				const int cz = (z == 0) ? 0 : (z == N - 1 ? 18 : 9);
				r[k].x       = c.x + omega * (fc[k].x - lap.x) * idiag[dix[k][0] + cz];
				r[k].y       = c.y + omega * (fc[k].y - lap.y) * idiag[dix[k][1] + cz];
			}
		}
		if (MODE == MODE_RESID_RESTRICT && rorth >= 0) {
			double a = PAR ? racc : 0.0; // (z0 is even) AvgRstr.h:95-102 order: x, then y, then z; each /2^D first
			a += r[0].x / 8;
			a += r[0].y / 8;
			a += r[1].x / 8;
			a += r[1].y / 8;
			racc = a;
			if (PAR && act) rdst[rsz * (z >> 1)] = a;
		} else if (act) {
			op2[z * NP + q[0]] = r[0];
			op2[z * NP + q[1]] = r[1];
		}
		if (RED != RED_NONE && act) { // (products added one by one, as k_reduce does: no contraction across the two sums)
#pragma unroll
			for (int k = 0; k < 2; k++) {
				if (RED == RED_OUT_A || RED == RED_OUT_A_OUT) {
					acc0 += r[k].x * ac[k].x;
					acc0 += r[k].y * ac[k].y;
				}
				if (RED == RED_OUT_A_OUT) {
					acc1 += r[k].x * r[k].x;
					acc1 += r[k].y * r[k].y;
				}
				if (RED == RED_OUT_OUT) {
					acc0 += r[k].x * r[k].x;
					acc0 += r[k].y * r[k].y;
				}
			}
		}
	};
	using B0 = std::integral_constant<int, 0>;
	using B1 = std::integral_constant<int, 1>;
	static_assert(ZL % 2 == 0 && ZL >= 4, "the march is unrolled over the two ring slots and ends with two steps of its own");
	TE_STAMP(2, true);
#pragma unroll 1
	for (int zz = 0; zz < ZL - 2; zz += 2) {
		step(B0{}, std::true_type{}, zz);
		step(B1{}, std::true_type{}, zz + 1);
	}
	TE_STAMP(4, false);
	step(B0{}, std::false_type{}, ZL - 2);
	step(B1{}, std::false_type{}, ZL - 1);
	TE_STAMP(5, false);
	TE_STAMP(6, true);
	TE_STAMP_FLUSH(L.stamp_dst, blockIdx.x);
	if (RED != RED_NONE) {
		__syncthreads(); // (the LDS of blockReduce2 is its own, but every wave must have left the plane loop's barriers)
		blockReduce2(acc0, acc1);
		if (tid == 0) {
			red.partial[2 * ((size_t) red.base + work)]     = acc0;
			red.partial[2 * ((size_t) red.base + work) + 1] = acc1;
		}
	}
}

// Fused prolongation (DrctIntp.h:99-106) for the first post-smoothing sweep: every value of u the
// sweep reads is taken as u + coarse[parent][(c + orthant offset)/2], so the corrected iterate is
// never written out and read back. Only used on levels where every patch is an octant child of a local
// parent and no coarse/fine ghost slot exists (uniform refinement); the host falls back otherwise. Faces
// whose neighbour lives on another rank arrive already corrected: the sender packs u + P(coarse)
// (k_pack_faces_prolong3d), so ghost slots are read as they are.
struct ProlongSrc {
	const int32_t *parent, *orth;
	const double  *coarse;
	// k_rbgs_resweep_prolong3d on a level whose coarser level lives on every rank (a replicated level): the parent of a
	// neighbour on ANOTHER rank is a local coarse patch too, so the correction of a ghost slot's values is formed here
	// as for a local neighbour -- the slot still holds the neighbour's face layer of v from the pre-sweep's exchange, and
	// no second exchange (nor its pack kernel) runs. gparent / gorth [ghost slot]: that coarse patch and orthant; null
	// when the slots hold v + P e already (sent that way: k_pack_faces6_3d)
	const int32_t *gparent = nullptr, *gorth = nullptr;
	// [patch][7]: offset inside `coarse` of the octant base (coarseOctant) of the patch itself and of its W, E, S, N, B, T
	// neighbours (-1: no neighbour patch on this rank behind that face), or null. The z-slab sweeps of the small levels take
	// their seven bases from here in one round trip; followed through face_src -> orth / parent, face by face under the
	// face-kind tests, they were ten dependent ones (profiles/r06_tail_stamps.txt: 2.4 us before the first plane is requested)
	const int64_t *cbase = nullptr;
};
// coarseOctant for the patch behind ghost slot `slot` (see ProlongSrc::gparent)
template <int N> __device__ __forceinline__ const double *coarseOctantSlot(const ProlongSrc &ps, int slot)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	const int     o = ps.gorth[slot];
	return ps.coarse + (size_t) ps.gparent[slot] * NNN + ((o & 1) ? H : 0) + N * ((o & 2) ? H : 0) + NN * ((o & 4) ? H : 0);
}
// base of the coarse octant that fine patch p maps onto: coarse cell of fine (x,y,z) = base[x/2 + N (y/2) + N^2 (z/2)]
template <int N> __device__ __forceinline__ const double *coarseOctant(const ProlongSrc &ps, int p)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	const int     o = ps.orth[p];
	return ps.coarse + (size_t) ps.parent[p] * NNN + ((o & 1) ? H : 0) + N * ((o & 2) ? H : 0) + NN * ((o & 4) ? H : 0);
}

// P(coarse) at one cell of patch p: the octant of the parent, or (orth < 0, the patch copies through) the same cell
template <int N> __device__ __forceinline__ double coarseAtCell(const ProlongSrc &ps, int p, int cell)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	const int     o = ps.orth[p];
	const double *c = ps.coarse + (size_t) ps.parent[p] * NNN;
	if (o < 0) return c[cell];
	const int x = cell % N, y = (cell / N) % N, z = cell / NN;
	return c[((o & 1) ? H : 0) + (x >> 1) + N * (((o & 2) ? H : 0) + (y >> 1)) + NN * (((o & 4) ? H : 0) + (z >> 1))];
}

// k_cf_ghost3d for the iterate u + P(coarse u) that is never stored: every value of u it reads gets its patch's
// correction added first (remote raw layers arrive corrected: the sender packs u + P e). Same weights
// (TriLinInterp.cpp:85-170), same order of operations on the corrected values.
template <int N>
__global__ void k_cf_ghost_prolong3d(const int32_t *__restrict__ desc, const int32_t *__restrict__ slots,
                                     const double *__restrict__ u, ProlongSrc ps, double *__restrict__ ghost)
{
	constexpr int  NN = N * N, NNN = N * N * N;
	const int32_t *d  = desc + (size_t) blockIdx.x * 8;
	const int      p = d[0], s = d[1], kind = d[2], q = d[3];
	const int      ax   = s >> 1;
	const int      sa   = (ax == 0) ? N : 1;
	const int      sb   = (ax == 2) ? N : NN;
	const int      sn   = (ax == 0) ? 1 : (ax == 1 ? N : NN);
	const int      mine = (s & 1) ? (N - 1) * sn : 0;
	const int      oth  = (s & 1) ? 0 : (N - 1) * sn;
	double        *g    = ghost + (size_t) slots[blockIdx.x] * NN;
	auto           U    = [&](int patch, int cell) { return u[(size_t) patch * NNN + cell] + coarseAtCell<N>(ps, patch, cell); };
	for (int i = threadIdx.x; i < NN; i += blockDim.x) {
		const int a = i % N, b = i / N;
		double    m = U(p, mine + a * sa + b * sb);
		double    gamma;
		if (kind == 2) {
			const int a0 = a & ~1, b0 = b & ~1;
			double    sum = 0;
			for (int bb = 0; bb < 2; bb++)
				for (int aa = 0; aa < 2; aa++)
					if (a0 + aa != a || b0 + bb != b) sum += U(p, mine + (a0 + aa) * sa + (b0 + bb) * sb);
			const int ca = (a + ((q & 1) ? N : 0)) / 2, cb = (b + ((q & 2) ? N : 0)) / 2;
			double    C  = d[4] >= 0 ? U(d[4], oth + ca * sa + cb * sb) : ghost[(size_t) (-(d[4] + 2)) * NN + ca + N * cb];
			gamma        = (11 * m - sum) / 12.0 + 4.0 * C / 12.0;
		} else {
			const int qa = (a >= N / 2), qb = (b >= N / 2);
			const int nbq = d[4 + qa + 2 * qb];
			const int fa = 2 * (a - qa * (N / 2)), fb = 2 * (b - qb * (N / 2));
			double    sum = 0;
			for (int bb = 0; bb < 2; bb++)
				for (int aa = 0; aa < 2; aa++)
					sum += 1.0 / 6.0 * (nbq >= 0 ? U(nbq, oth + (fa + aa) * sa + (fb + bb) * sb)
					                             : ghost[(size_t) (-(nbq + 2)) * NN + (fa + aa) + N * (fb + bb)]);
			gamma = 2.0 / 6.0 * m + sum;
		}
		g[i] = 2 * gamma - m;
	}
}

// k_cf_ghost3d / k_cf_ghost_prolong3d for an iterate that exists only as its six face layers (opts.fuse = 3 on a
// refined level): every value the interface weights touch is a face-layer value. f6: [patch][side][a + N b].
// PROLONG: the iterate is that + P(coarse), formed value by value as k_cf_ghost_prolong3d does.
template <int N, bool PROLONG>
__global__ void k_cf_ghost6_3d(const int32_t *__restrict__ desc, const int32_t *__restrict__ slots, const double *__restrict__ f6,
                               ProlongSrc ps, double *__restrict__ ghost, const int32_t *__restrict__ f6off)
{
	constexpr int  NN = N * N;
	const int32_t *d  = desc + (size_t) blockIdx.x * 8;
	const int      p = d[0], s = d[1], kind = d[2], q = d[3];
	const int      ax   = s >> 1;
	const int      sa   = (ax == 0) ? N : 1;
	const int      sb   = (ax == 2) ? N : NN;
	const int      sn   = (ax == 0) ? 1 : (ax == 1 ? N : NN);
	const int      mine = (s & 1) ? (N - 1) * sn : 0;
	const int      oth  = (s & 1) ? 0 : (N - 1) * sn;
	double        *g    = ghost + (size_t) slots[blockIdx.x] * NN;
	// value at face cell (a, b) of `patch` on its side `side` (cell index base + a sa + b sb inside the patch)
	auto U = [&](int patch, int side, int base, int a, int b) {
		double v = f6[f6Face<N>(f6off, patch, side) + a + N * b];
		if (PROLONG) v += coarseAtCell<N>(ps, patch, base + a * sa + b * sb);
		return v;
	};
	for (int i = threadIdx.x; i < NN; i += blockDim.x) {
		const int a = i % N, b = i / N;
		double    m = U(p, s, mine, a, b);
		double    gamma;
		if (kind == 2) {
			const int a0 = a & ~1, b0 = b & ~1;
			double    sum = 0;
			for (int bb = 0; bb < 2; bb++)
				for (int aa = 0; aa < 2; aa++)
					if (a0 + aa != a || b0 + bb != b) sum += U(p, s, mine, a0 + aa, b0 + bb);
			const int ca = (a + ((q & 1) ? N : 0)) / 2, cb = (b + ((q & 2) ? N : 0)) / 2;
			double    C  = d[4] >= 0 ? U(d[4], s ^ 1, oth, ca, cb) : ghost[(size_t) (-(d[4] + 2)) * NN + ca + N * cb];
			gamma        = (11 * m - sum) / 12.0 + 4.0 * C / 12.0;
		} else {
			const int qa = (a >= N / 2), qb = (b >= N / 2);
			const int nbq = d[4 + qa + 2 * qb];
			const int fa = 2 * (a - qa * (N / 2)), fb = 2 * (b - qb * (N / 2));
			double    sum = 0;
			for (int bb = 0; bb < 2; bb++)
				for (int aa = 0; aa < 2; aa++)
					sum += 1.0 / 6.0 * (nbq >= 0 ? U(nbq, s ^ 1, oth, fa + aa, fb + bb) : ghost[(size_t) (-(nbq + 2)) * NN + (fa + aa) + N * (fb + bb)]);
			gamma = 2.0 / 6.0 * m + sum;
		}
		g[i] = 2 * gamma - m;
	}
}

// k_pack_faces3d for the iterate u + P(coarse u) that is never stored (see ProlongSrc): the face layers other
// ranks need, with this rank's coarse correction added on the way out.
template <int N>
__global__ void k_pack_faces_prolong3d(const int32_t *__restrict__ faces, const double *__restrict__ u, ProlongSrc ps,
                                       double *__restrict__ sendbuf, PackPush pp = PackPush())
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	const int     p = faces[2 * blockIdx.x], s = faces[2 * blockIdx.x + 1];
	const int     ax = s >> 1;
	const int     sa = (ax == 0) ? N : 1, sb = (ax == 2) ? N : NN, sn = (ax == 0) ? 1 : (ax == 1 ? N : NN);
	const int     face = (s & 1) ? (N - 1) * sn : 0;
	const double *up   = u + (size_t) p * NNN + face;
	double       *o    = pp.dst ? pp.dst[blockIdx.x] : sendbuf + (size_t) blockIdx.x * NN;
	if (pp.dst && *pp.err) {
		// (a wait has given up: nothing is stored any more)
	} else if (ps.orth[p] < 0) { // copy-through patch: the correction is the same-size coarse patch
		for (int i = threadIdx.x; i < NN; i += blockDim.x) {
			const int cell = (i % N) * sa + (i / N) * sb;
			o[i]           = up[cell] + coarseAtCell<N>(ps, p, face + cell);
		}
	} else {
		const double *cp = coarseOctant<N>(ps, p) + ((s & 1) ? (H - 1) * sn : 0);
		for (int i = threadIdx.x; i < NN; i += blockDim.x) {
			const int a = i % N, b = i / N;
			o[i] = up[a * sa + b * sb] + cp[(a / 2) * sa + (b / 2) * sb];
		}
	}
	packPushTail(pp);
}

// Relax cell CB (0: even x, 1: odd x) of row k of the plane held in `cen` (LDS copy in tl):
// v = (sum of off-diagonal neighbours / h^2 - f) / diag. Everything about the cell's position is static.
template <int N, int K, int CB, bool ZERO_NBRS, bool STORE = true>
__device__ __forceinline__ void relaxCell(double *tl, const double *idiag, int cz9, const int (&lds)[2], const int (&ldo)[2], const int (&dix)[2][2],
                                          bool act, double rhx, double rhy, double rhz, double2 (&cen)[2],
                                          const double2 (&below)[2], const double2 (&above)[2], const double2 (&rhs)[2])
{
	// No implicit contraction in here: a neighbour value that was itself just produced as (o - rh) * idiag would be
	// fused into the sum below as one FMA whenever both updates land in the same basic block, which depends
	// on the instantiation (ZS, PROLONG) -- the variants must stay bit-identical to each other. The two
	// intended FMAs are explicit.
#pragma clang fp contract(off)
	const double  rh = CB ? rhs[K].y : rhs[K].x;
	double        o  = 0.0;
	if (!ZERO_NBRS) { // ZERO_NBRS: every neighbour is known to be 0 (first red half-sweep from a zero guess)
		const double side  = CB ? tl[lds[K] + 2] : tl[lds[K] - 1]; // the x-neighbour outside the pair
		const double mate  = CB ? cen[K].x : cen[K].y;
		const double inner = CB ? cen[1 - K].y : cen[1 - K].x;                            // other row of the pair: a register
		const double outer = (K == 0) ? tl[ldo[0] + CB] : tl[ldo[1] + CB];                    // row y-1 / y+2: LDS
		const double ym = (K == 0) ? outer : inner, yp = (K == 0) ? inner : outer;
		const double zb = CB ? below[K].y : below[K].x, za = CB ? above[K].y : above[K].x;
		o = __builtin_fma(zb + za, rhz, __builtin_fma(ym + yp, rhy, (side + mate) * rhx));
	}
	const double v = (o - rh) * idiag[dix[K][CB] + cz9];
	if (CB)
		cen[K].y = v;
	else
		cen[K].x = v;
	if (STORE && act) tl[lds[K] + CB] = v; // (STORE = false: nobody reads this plane's LDS copy again)
}

// Patch-local red-black Gauss-Seidel sweep with neighbour ghosts frozen at the old iterate
// (hybrid GS: Gauss-Seidel inside the patch, Jacobi across patch faces), out-of-place:
// out = S(u, f). Red = (x+y+z) even. Plane z gets its red update from old black values;
// plane z-1 then gets its black update from new red values, so output lags one plane and the sweep
// costs one pass over u and f. Physical faces are folded into the diagonal (k = 3 Dirichlet,
// 1 Neumann), so their ghost contributes 0 to the off-diagonal sum.
// ZERO: the sweep starts from u == 0 (first pre-smoothing sweep of a cycle, Cycle.h:118 / :63): u is
// never read (neither the patch nor any ghost); bit-identical to the general kernel fed with zeros.
// PROLONG: the sweep runs on u + P(coarse) (see ProlongSrc).
// ZS > 1 (levels with few patches): a patch is split into ZS z-slabs, one workgroup each. A slab [z0, z1) needs
// the new red values of planes z0-1 and z1, which depend on old values only: they are recomputed (never
// stored), so the result is bit-identical to the whole-patch sweep at 2/(N/ZS) extra arithmetic and reads.
// CFP (with PROLONG, ZS = 1): the level is refined -- some patches copy through to the coarser level (orth < 0: their
// correction is the same-size coarse patch, cell by cell) and coarse/fine ghost slots exist (filled from u + P e
// by k_cf_ghost_prolong3d).
template <int N, bool ZERO, bool PROLONG, int ZS = 1, bool CFP = false>
__global__ __launch_bounds__(Tile3<N>::TPB) void k_rbgs3d(LevelDev L, const double *__restrict__ u,
                                                          const double *__restrict__ f,
                                                          double *__restrict__ out, ProlongSrc ps)
{
	using T           = Tile3<N>;
	constexpr int TPB = T::TPB, NP = T::NP, H = T::H;
	constexpr int NN  = N * N, NNN = N * N * N;
	constexpr int ZL  = N / ZS; // planes per slab (even)
	static_assert(ZL % 2 == 0 && ZL >= 2, "slabs start on even planes");
	TE_STAMP_DECL;
	TE_STAMP(0, false);
	if (ZS > 1) argsUpFront(L, u, f, out, PROLONG ? (const void *) ps.cbase : (const void *) ps.parent, ps.orth, ps.coarse);
	const int     nwork = L.count * ZS;
	const int     work  = xcdRemap(blockIdx.x, nwork);
	if (work >= nwork) return;
	const int slot = work / ZS;
	const int z0 = (work % ZS) * ZL, z1 = z0 + ZL; // planes this workgroup stores
	const int zs = (z0 > 0) ? z0 - 1 : 0;          // first plane it relaxes (red only when zs < z0)
	const int pid = L.order ? L.order[L.first + slot] : L.first + slot;
	const int tid = threadIdx.x;

	__shared__ __attribute__((aligned(16))) double tile[3][T::LSZ]; // planes z-1, z, z+1 rotate
	__shared__ double idiag[27]; // 1/diag per (x,y,z) position class

	const Reg6     fk(L.face_kind + (size_t) pid * 6), fs(L.face_src + (size_t) pid * 6);
	int64_t        cb[7] = {0, -1, -1, -1, -1, -1, -1}; // ProlongSrc::cbase (the host passes it with every launch of this variant)
	if (PROLONG && !CFP) {
#pragma unroll
		for (int i = 0; i < 7; i++) cb[i] = ps.cbase[(size_t) pid * 7 + i];
	}
	const double   rhx = L.rh2[(size_t) pid * 3], rhy = L.rh2[(size_t) pid * 3 + 1], rhz = L.rh2[(size_t) pid * 3 + 2];
	const double  *up  = u + (size_t) pid * NNN;
	const double2 *up2 = reinterpret_cast<const double2 *>(up);
	const double2 *fp2 = reinterpret_cast<const double2 *>(f + (size_t) pid * NNN);
	double2       *op2 = reinterpret_cast<double2 *>(out + (size_t) pid * NNN);

	FaceKinds kinds(fk);
	idiagTable(idiag, tid, kinds, rhx, rhy, rhz);

	HaloSrc  hs;
	PlaneSrc bot, top;
	if (!ZERO) {
		hs  = haloSrc<N>(tid, fk, fs, u, up, L.ghost, 0.0, 0.0, L.xf);
		bot = zPlaneSrc<N>(fk[4], fs[4], false, u, up, L.ghost, 0.0, 0.0);
		top = zPlaneSrc<N>(fk[5], fs[5], true, u, up, L.ghost, 0.0, 0.0);
	} else { // halo ring stays zero for the whole sweep
		for (int i = tid; i < 3 * T::LSZ; i += TPB) (&tile[0][0])[i] = 0.0;
	}

	const bool act = (T::NT == TPB) || tid < T::NT;
	const int  X = act ? tid % H : 0, Yp = act ? tid / H : 0;
	int        q[2], lds[2], dix[2][2];
	const int  ldo[2] = {T::row(2 * Yp) + 2 * X + 2, T::row(2 * Yp + 3) + 2 * X + 2}; // the rows below / above the pair
#pragma unroll
	for (int k = 0; k < 2; k++) {
		q[k]   = (2 * Yp + k) * H + X;
		lds[k] = T::row(2 * Yp + k + 1) + 2 * X + 2;
	}
	{
		const int cy0 = (Yp == 0) ? 0 : 1, cy1 = (Yp == H - 1) ? 2 : 1;
		const int cx0 = (X == 0) ? 0 : 1, cx1 = (X == H - 1) ? 2 : 1;
		dix[0][0] = cx0 + 3 * cy0, dix[0][1] = cx1 + 3 * cy0, dix[1][0] = cx0 + 3 * cy1, dix[1][1] = cx1 + 3 * cy1;
	}

	// coarse-correction sources matching the own planes / hs / bot / top (PROLONG only)
	const double *cown = nullptr, *chalo = nullptr, *cbot = nullptr, *ctop = nullptr;
	double        shalo = 0.0, sbot = 0.0, stop = 0.0;
	const int     cq = X + N * Yp; // in-plane coarse offset of this thread's cells (both rows share it)
	// CFP: copy-through patches (own / bottom / top neighbour) are addressed cell by cell; hsh = z shift of the halo's
	// coarse index (0 for a copy-through neighbour)
	bool          cpo = false, cpb = false, cpt = false;
	const double *yown = nullptr, *ybot = nullptr, *ytop = nullptr;
	int           hsh = 1;
	static_assert(!CFP || (PROLONG && ZS == 1), "CFP is a variant of the fused-prolongation sweep without z-slabs");
	if (PROLONG && !CFP) { // (the seven bases were requested with the face tables; the same addresses as below)
		cown  = ps.coarse + cb[0];
		chalo = cbot = ctop = cown;
		if (tid < 4 * N) {
			const int     side = tid / N, t = tid % N;
			const int64_t b    = side == 0 ? cb[1] : (side == 1 ? cb[2] : (side == 2 ? cb[3] : cb[4]));
			if (b >= 0) {
				const int cx = (side == 0) ? H - 1 : (side == 1 ? 0 : t / 2);
				const int cy = (side == 2) ? H - 1 : (side == 3 ? 0 : t / 2);
				chalo        = ps.coarse + b + cx + N * cy;
				shalo        = 1.0;
			}
		}
		if (cb[5] >= 0) cbot = ps.coarse + cb[5] + NN * (H - 1), sbot = 1.0;
		if (cb[6] >= 0) ctop = ps.coarse + cb[6], stop = 1.0;
	} else if (PROLONG) {
		if (CFP && ps.orth[pid] < 0) {
			cpo  = true;
			yown = ps.coarse + (size_t) ps.parent[pid] * NNN;
			cown = yown;
		} else {
			cown = coarseOctant<N>(ps, pid);
		}
		chalo = cbot = ctop = cown; // harmless valid address where no correction applies (scale 0)
		if (tid < 4 * N) {
			const int side = tid / N, t = tid % N;
			if (fk[side] == FACE_LOCAL) {
				if (CFP && ps.orth[fs[side]] < 0) { // the neighbour's facing cell in its same-size coarse patch
					const int nbr = (side == 0) ? t * N + (N - 1) : (side == 1 ? t * N : (side == 2 ? (N - 1) * N + t : t));
					chalo = ps.coarse + (size_t) ps.parent[fs[side]] * NNN + nbr;
					hsh   = 0;
				} else {
					const double *cn = coarseOctant<N>(ps, fs[side]);
					// the neighbour's facing cell: west (N-1,t) east (0,t) south (t,N-1) north (t,0)
					const int cx = (side == 0) ? H - 1 : (side == 1 ? 0 : t / 2);
					const int cy = (side == 2) ? H - 1 : (side == 3 ? 0 : t / 2);
					chalo = cn + cx + N * cy;
				}
				shalo = 1.0;
			}
		}
		if (fk[4] == FACE_LOCAL) {
			if (CFP && ps.orth[fs[4]] < 0)
				cpb = true, ybot = ps.coarse + (size_t) ps.parent[fs[4]] * NNN + NN * (N - 1);
			else
				cbot = coarseOctant<N>(ps, fs[4]) + NN * (H - 1);
			sbot = 1.0;
		}
		if (fk[5] == FACE_LOCAL) {
			if (CFP && ps.orth[fs[5]] < 0)
				cpt = true, ytop = ps.coarse + (size_t) ps.parent[fs[5]] * NNN;
			else
				ctop = coarseOctant<N>(ps, fs[5]);
			stop = 1.0;
		}
	}
	// the correction of this thread's two cells of row k: plane z of the own patch / the bottom / top neighbour's
	// facing plane (CFP only; the uniform case uses the scalar forms below)
	auto ownC = [&](int k, int z) {
		if (cpo) return *reinterpret_cast<const double2 *>(yown + (size_t) z * NN + (2 * Yp + k) * N + 2 * X);
		const double c = cown[NN * (z >> 1) + cq];
		return double2{c, c};
	};
	auto botC = [&](int k) {
		if (cpb) return *reinterpret_cast<const double2 *>(ybot + (2 * Yp + k) * N + 2 * X);
		const double c = cbot[cq];
		return double2{c, c};
	};
	auto topC = [&](int k) {
		if (cpt) return *reinterpret_cast<const double2 *>(ytop + (2 * Yp + k) * N + 2 * X);
		const double c = ctop[cq];
		return double2{c, c};
	};

	TE_STAMP(1, true);
	// planes: umm = z-2, um = z-1, uc = z, un = z+1, un2 = z+2 (values are updated in place); start at z = zs
	double2 umm[2], um[2], uc[2], un[2], un2[2], fm[2], fc[2], fn[2];
#pragma unroll
	for (int k = 0; k < 2; k++) {
		if (!ZERO) {
			const double2 *pm = (zs > 0) ? up2 + (zs - 1) * NP : bot.p; // zs + 1 < N always (a slab has >= 2 planes)
			const double   sm = (zs > 0) ? 1.0 : bot.s;
			uc[k]     = up2[zs * NP + q[k]];
			double2 a = pm[q[k]];
			um[k]     = double2{sm * a.x, sm * a.y};
			un[k]     = up2[(zs + 1) * NP + q[k]];
			if (PROLONG && !CFP) {
				const double c0 = cown[NN * (zs >> 1) + cq], c1 = cown[NN * ((zs + 1) >> 1) + cq];
				// (one load from a selected address, not a load under a branch: the compiler waits for EVERYTHING in flight where such a
				// branch rejoins -- a whole memory latency in the middle of the prologue's requests; 1.0 * x is x)
				const double *cbp = (zs > 0) ? cown + NN * ((zs - 1) >> 1) : cbot;
				const double  cb  = ((zs > 0) ? 1.0 : sbot) * cbp[cq];
				uc[k].x += c0, uc[k].y += c0, un[k].x += c1, un[k].y += c1;
				um[k].x += sm * cb, um[k].y += sm * cb;
			}
			if (PROLONG && CFP) { // zs = 0
				const double2 c0 = ownC(k, 0), c1 = ownC(k, 1), cb = botC(k);
				uc[k].x += c0.x, uc[k].y += c0.y, un[k].x += c1.x, un[k].y += c1.y;
				um[k].x += sm * (sbot * cb.x), um[k].y += sm * (sbot * cb.y);
			}
		} else {
			uc[k] = um[k] = un[k] = un2[k] = double2{0.0, 0.0};
		}
		fc[k]  = fp2[zs * NP + q[k]];
		umm[k] = double2{0.0, 0.0};
		fm[k]  = double2{0.0, 0.0};
	}
	double hv = ZERO ? 0.0 : hs.s * (hs.p[zs * hs.stride] + (PROLONG ? shalo * chalo[NN * (zs >> hsh)] : 0.0));
	TE_STAMP(2, true);
	__syncthreads(); // idiag (and the zeroed tiles)

	int bz = 0; // z % 3
	// one plane step; ZPAR = z & 1 is a compile-time constant so that every cell's colour is static
	auto step = [&](auto zpar, int z) {
		constexpr int ZPAR = decltype(zpar)::value;
		// issue the next iteration's loads (clamped / redirected on the last iterations)
		const int zc  = (z + 1 < N) ? z + 1 : N - 1;
		double    hvn = 0.0;
		if (!ZERO) {
			const double2 *pn = (z + 2 < N) ? up2 + (z + 2) * NP : top.p;
			const double   sn = (z + 2 < N) ? 1.0 : top.s;
#pragma unroll
			for (int k = 0; k < 2; k++) {
				double2 a = pn[q[k]];
				if (PROLONG && !CFP) { // plane z+2 of the patch, or the top neighbour's plane 0
					const double *cp = (z + 2 < N) ? cown + NN * ((z + 2) >> 1) : ctop;
					const double  cs = (z + 2 < N) ? 1.0 : stop;
					const double  c  = cs * cp[cq];
					a.x += c, a.y += c;
				}
				if (PROLONG && CFP) {
					const double2 c  = (z + 2 < N) ? ownC(k, z + 2) : topC(k);
					const double  cs = (z + 2 < N) ? 1.0 : stop;
					a.x += cs * c.x, a.y += cs * c.y;
				}
				un2[k] = double2{sn * a.x, sn * a.y};
			}
			hvn = hs.s * (hs.p[zc * hs.stride] + (PROLONG ? shalo * chalo[NN * (zc >> hsh)] : 0.0));
		}
#pragma unroll
		for (int k = 0; k < 2; k++) fn[k] = fp2[zc * NP + q[k]];
		double *tz = tile[bz];                    // plane z
		double *tm = tile[bz == 0 ? 2 : bz - 1];  // plane z-1
		if (!ZERO && z < N) {
			if (act) {
				ldsStore2(tz + lds[0], uc[0]);
				ldsStore2(tz + lds[1], uc[1]);
			}
			if (hs.lds >= 0) tz[hs.lds] = hv;
		}
		// one barrier per plane: buffer z%3 was last read two iterations ago (black of plane z-3)
		ldsBarrier();
		if (z < N) {
			// red cells of plane z from old black values: cell parity = (0 + k + z) & 1
			const int cz9 = (z == 0) ? 0 : (z == N - 1 ? 18 : 9);
			relaxCell<N, 0, (0 + ZPAR) & 1, ZERO>(tz, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, uc, um, un, fc);
			relaxCell<N, 1, (1 + ZPAR) & 1, ZERO>(tz, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, uc, um, un, fc);
		}
		if (z > z0) { // (the plane below a slab only lends its red values: no black update, nothing stored)
			// black cells of plane z-1: x/y neighbours = new red (LDS / registers); z neighbours = umm (new
			// red, or the frozen bottom ghost) and uc (new red, or the frozen top ghost when z == N).
			// plane z-1 has parity 1-ZPAR; black: (x + y + z - 1) odd -> cell parity = (1 + k + (1-ZPAR)) & 1
			const int cz9 = (z - 1 == 0) ? 0 : (z - 1 == N - 1 ? 18 : 9);
			relaxCell<N, 0, (1 + 0 + 1 - ZPAR) & 1, false>(tm, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, um, umm, uc, fm);
			relaxCell<N, 1, (1 + 1 + 1 - ZPAR) & 1, false>(tm, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, um, umm, uc, fm);
			if (act) {
				op2[(z - 1) * NP + q[0]] = um[0];
				op2[(z - 1) * NP + q[1]] = um[1];
				if (L.xf_out) { // export this plane's x-face columns of the new iterate (rows 2Yp, 2Yp+1 are adjacent)
					double *xo = L.xf_out + (size_t) pid * 2 * NN + N * (z - 1) + 2 * Yp;
					if (X == 0) *reinterpret_cast<double2 *>(xo) = double2{um[0].x, um[1].x};
					if (X == H - 1) *reinterpret_cast<double2 *>(xo + NN) = double2{um[0].y, um[1].y};
				}
			}
		}
#pragma unroll
		for (int k = 0; k < 2; k++) {
			umm[k] = um[k];
			um[k]  = uc[k];
			uc[k]  = un[k];
			un[k]  = un2[k];
			fm[k]  = fc[k];
			fc[k]  = fn[k];
		}
		hv = hvn;
		bz = (bz == 2) ? 0 : bz + 1;
	};
	if (ZS > 1 && z0 > 0) step(std::integral_constant<int, 1>{}, z0 - 1); // red values of the plane below the slab
	TE_STAMP(3, false);
#pragma unroll 1
	for (int z = z0; z < z1; z += 2) {
		step(std::integral_constant<int, 0>{}, z);
		step(std::integral_constant<int, 1>{}, z + 1);
	}
	TE_STAMP(4, false);
	step(std::integral_constant<int, 0>{}, z1); // black update and store of the last plane (and, inside a patch, the red values above it)
	TE_STAMP(5, false);
	TE_STAMP(6, true);
	TE_STAMP_FLUSH(L.stamp_dst, blockIdx.x);
}

// ---- ghost terms of the restricted residual without a pass of their own (opts.fuse = 3, two fused levels in a row) ---
// k_restrict_fixup3d adds, for every face with a neighbour, -(1/h^2)/8 * (the four neighbour values behind a 2x2 block of
// face cells) to the coarse cell behind the block. For x faces those coarse cells sit one per 64-byte sector: over a whole
// level the fix-up reads and rewrites every sector of the coarse vector, 0.10 ms of a 1.09 ms cycle at 512^3. Instead, the
// patch that OWNS the face values forms the finished terms (it holds the values in registers when its plane is final),
// k_fcorr_gather3d adds the y and z terms to the coarse right-hand side in place (contiguous rows and planes) and sorts
// the x terms into a compact side array of the coarse level, `fcorr` [coarse patch][4][N*N]: planes x = 0, H-1, H, N-1
// (the x faces of the eight child octants) at (y + N z), pure stores, one writer per entry; the kernels that read the
// coarse right-hand side (k_rbgs_zero_resid3d / k_rbgs_resweep_prolong3d with FCORR) add them while loading:
// f = ((f + y term) + z term) + x term -- the order in which k_restrict_fixup3d visits the faces (S,N,B,T,W,E), and the
// same (w * g) / 8 summed in the same order, so the result is bit-identical to the fix-up pass. Entries of physical
// faces are +0.
template <int N> __device__ __forceinline__ int octPlane(int c) // 0, H-1, H, N-1 -> 0..3; other coordinates -> -1
{
	constexpr int H = N / 2;
	return c == 0 ? 0 : (c == H - 1 ? 1 : (c == H ? 2 : (c == N - 1 ? 3 : -1)));
}
// what a reader of the coarse right-hand side adds to its two rows (k) of plane z: the x terms (the y and z terms were added
// in place by k_fcorr_gather3d). fcorr: [patch][4][N*N], plane j at (y + N z).
template <int N> struct FCorrSrc {
	// N >= 8: at most one of a thread's two cells lies on an octant face (one pointer, one pair of values); N = 4: both may
	static constexpr int NX = (N >= 8) ? 1 : 2;
	const double        *x[NX]; // that cell's x plane at row 2Yp (a harmless valid address where has[c] is false)
	bool                 has[NX];
	int                  xc;    // N >= 8: which cell of the pair it is
	__device__ __forceinline__ void init(const double *fcorr, int pid, int X, int Yp)
	{
		constexpr int NN = N * N;
		const double *b  = fcorr + (size_t) pid * 4 * NN;
		if (NX == 1) {
			const int j0 = octPlane<N>(2 * X), j1 = octPlane<N>(2 * X + 1), j = j0 >= 0 ? j0 : j1;
			xc     = j0 >= 0 ? 0 : 1;
			has[0] = j >= 0;
			x[0]   = b + (size_t) (j >= 0 ? j : 0) * NN + 2 * Yp;
		} else {
			xc = 0;
#pragma unroll
			for (int c = 0; c < NX; c++) {
				const int j = octPlane<N>(2 * X + c);
				has[c]      = j >= 0;
				x[c]        = b + (size_t) (j >= 0 ? j : 0) * NN + 2 * Yp;
			}
		}
	}
	// issue the loads for plane z: cx[c] = {row 0, row 1} of that cell. Unconditional: a load under a per-lane condition is
	// merged with its default right behind it, and that merge waits for the load in the step that issued it.
	__device__ __forceinline__ void load(int z, double2 (&cx)[NX]) const
	{
#pragma unroll
		for (int c = 0; c < NX; c++) cx[c] = *reinterpret_cast<const double2 *>(x[c] + N * z);
	}
	__device__ __forceinline__ void apply(double2 (&f)[2], const double2 (&cx)[NX]) const
	{
#pragma clang fp contract(off)
		if (NX == 1) { // (the other cell, and both where no cell lies on an octant face, take + 0)
			const bool   h0 = has[0] && xc == 0, h1 = has[0] && xc != 0;
			const double a0 = h0 ? cx[0].x : 0.0, a1 = h1 ? cx[0].x : 0.0;
			const double b0 = h0 ? cx[0].y : 0.0, b1 = h1 ? cx[0].y : 0.0;
			f[0].x = f[0].x + a0, f[0].y = f[0].y + a1;
			f[1].x = f[1].x + b0, f[1].y = f[1].y + b1;
		} else {
			f[0].x = f[0].x + (has[0] ? cx[0].x : 0.0);
			f[0].y = f[0].y + (has[NX - 1] ? cx[NX - 1].x : 0.0);
			f[1].x = f[1].x + (has[0] ? cx[0].y : 0.0);
			f[1].y = f[1].y + (has[NX - 1] ? cx[NX - 1].y : 0.0);
		}
	}
};
// fcorr of the coarse level from the finished 2x2 sums the fine patches left in their own `rs6` [fine patch][6][H*H]
// (k_rbgs_zero_resid3d EXPORT writes them next to its face layers: small stores into the patch's own region -- written
// straight into six far-away coarse planes they stalled the kernel's in-order memory pipeline and doubled its time).
// One workgroup per coarse patch and plane: entry (a, b) of plane j of axis ax belongs to child octant o; the term
// comes from the patch across that child's face: rs6 of a local neighbour, or (neighbour on another rank) the sums of
// k_restrict_fixup3d formed here from the ghost slot. Physical faces: +0. A pure permutation copy of finished values.
// rs6 == null (a refined fine level: copy-through patches, coarse/fine ghost slots): the sums are formed here from the
// fine level's face layers f6; a coarse patch that is a copy of a fine patch takes w g cell by cell on its own two faces
// of each axis, as the copy-through branch of k_restrict_fixup3d adds them.
// gtab [coarse patch][axis][plane j][quadrant oa + 2 ob] (built once per level by k_gather_table3d): where in rs6 the 16x16
// block of finished terms of that quarter plane starts; -1: the terms are zero (physical face, no such child); -2: take the
// general path (neighbour on another rank, copy-through patch). One cached table read instead of the dependent chain
// child -> face kind / source -> value (the kernel is all latency: 90 % of its wave cycles wait). Tried and dropped: a thread's
// four entries in stages (indices together, then values together from a harmless address where there is nothing to read, the
// general chain only where the table has no answer): 36 -> 52-59 us on the same box.
template <int N>
__global__ void k_gather_table3d(LevelDev L, const int32_t *__restrict__ child, const int32_t *__restrict__ copy, int32_t *__restrict__ gtab)
{
	constexpr int HH = (N / 2) * (N / 2);
	const int     pc = blockIdx.x, e = threadIdx.x;
	if (e >= 48) return;
	const int ax = e / 16, j = (e / 4) % 4, oa = e & 1, ob = (e >> 1) & 1;
	int32_t   v;
	if (copy && copy[pc]) {
		v = -2;
	} else {
		const int a0 = (ax == 0) ? 1 : 0, a1 = (ax == 2) ? 1 : 2;
		const int s = 2 * ax + (j & 1), hi = j >> 1;
		const int p = child[(size_t) pc * 8 + ((hi << ax) | (oa << a0) | (ob << a1))];
		if (p < 0) {
			v = -1;
		} else {
			const int kind = L.face_kind[(size_t) p * 6 + s], src = L.face_src[(size_t) p * 6 + s];
			v              = kind < FACE_LOCAL ? -1 : (kind == FACE_LOCAL ? (int32_t) (((size_t) src * 6 + (s ^ 1)) * HH) : -2);
		}
	}
	gtab[(size_t) pc * 48 + e] = v;
}
template <int N>
__global__ __launch_bounds__(256) void k_fcorr_gather3d(LevelDev L, const int32_t *__restrict__ child, const int32_t *__restrict__ copy,
                                                        const double *__restrict__ rs6, const double *__restrict__ f6,
                                                        double *__restrict__ coarse, double *__restrict__ fcorr,
                                                        const int32_t *__restrict__ gtab)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2, HH = H * H;
	// one workgroup per plane; a thread's QN entries are independent chains of dependent loads (tables, then the value): all
	// terms first, then the stores, so that the chains overlap
	constexpr int QN = (NN + 255) / 256;
	const int     pc = blockIdx.x / 12, blk = blockIdx.x % 12; // 0..3: the x planes; 4..7: the y planes; 8..11: the z planes
	const bool    cp = copy && copy[pc];
	// the term of entry (a, b) of plane j of axis ax of this coarse patch
	auto term = [&](int ax, int j, int a, int b) -> double {
		if (rs6 && gtab) { // (a uniformly refined fine level: the finished sums of the neighbours)
			const int oa = a >= H, ob = b >= H;
			const int t  = gtab[(size_t) pc * 48 + ax * 16 + j * 4 + oa + 2 * ob];
			if (t >= 0) return rs6[(size_t) t + (a - oa * H) + H * (b - ob * H)];
			if (t == -1) return 0.0;
		}
		const int a0 = (ax == 0) ? 1 : 0, a1 = (ax == 2) ? 1 : 2; // the two other axes in order
		if (cp) { // the coarse patch IS a fine patch that does not coarsen: its own two faces of the axis, cell by cell (w g)
			if (j == 1 || j == 2) return 0.0;
			const int p = child[(size_t) pc * 8], sp = 2 * ax + (j == 3);
			if (p < 0) return 0.0;
			const int kind = L.face_kind[(size_t) p * 6 + sp], src = L.face_src[(size_t) p * 6 + sp];
			if (kind < FACE_LOCAL) return 0.0;
			const double g = kind == FACE_GHOST ? L.ghost[(size_t) src * NN + a + N * b] : f6[f6Face<N>(L.f6off, src, sp ^ 1) + a + N * b];
			return -L.rh2[(size_t) p * 3 + ax] * g;
		}
		const int s = 2 * ax + (j & 1), hi = j >> 1; // j = 0: low face of the low child, 1: its high face, 2, 3: the high child's
		const int oa = a >= H, ob = b >= H, ha = a - oa * H, hb = b - ob * H;
		const int p = child[(size_t) pc * 8 + ((hi << ax) | (oa << a0) | (ob << a1))];
		if (p < 0) return 0.0;
		const int kind = L.face_kind[(size_t) p * 6 + s], src = L.face_src[(size_t) p * 6 + s];
		if (kind < FACE_LOCAL) return 0.0;
		if (kind == FACE_LOCAL && rs6) return rs6[((size_t) src * 6 + (s ^ 1)) * HH + ha + H * hb];
		// the sum of k_restrict_fixup3d over the 2x2 block, first face coordinate fastest
		const double  w  = -L.rh2[(size_t) p * 3 + ax];
		const double *gp = kind == FACE_GHOST ? L.ghost + (size_t) src * NN : f6 + f6Face<N>(L.f6off, src, s ^ 1);
		double        v  = 0.0;
#pragma unroll
		for (int db = 0; db < 2; db++)
#pragma unroll
			for (int da = 0; da < 2; da++) v += (w * gp[(2 * ha + da) + N * (2 * hb + db)]) / 8;
		return v;
	};
	constexpr int coord[4] = {0, H - 1, H, N - 1};
	double       *cv       = coarse + (size_t) pc * NNN;
	const int     j        = blk & 3;
	double        v[QN], vy[QN], f[QN];
	if (blk < 4) { // an x plane: into the side array
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int i = threadIdx.x + 256 * k;
			v[k]        = i < NN ? term(0, j, i % N, i / N) : 0.0;
		}
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int i = threadIdx.x + 256 * k;
			if (i < NN) fcorr[((size_t) pc * 4 + j) * NN + i] = v[k];
		}
	} else if (blk < 8) { // a y plane: entry (x, z) -> cell (x, coord[j], z); the cells that also lie on a z plane are that plane's
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int i = threadIdx.x + 256 * k, a = i % N, b = i / N;
			v[k]        = (i < NN && octPlane<N>(b) < 0) ? term(1, j, a, b) : 0.0;
		}
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int i = threadIdx.x + 256 * k, a = i % N, b = i / N;
			if (v[k] != 0.0) cv[a + N * coord[j] + NN * b] += v[k];
		}
	} else { // a z plane: entry (x, y) -> cell (x, y, coord[j]); a cell that lies on a y plane too takes its y term first
#pragma clang fp contract(off)
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int i = threadIdx.x + 256 * k, a = i % N, b = i / N, jy = octPlane<N>(b);
			const bool in = i < NN;
			f[k]          = in ? cv[a + N * b + NN * coord[j]] : 0.0;
			vy[k]         = (in && jy >= 0) ? term(1, jy, a, coord[j]) : 0.0;
			v[k]          = in ? term(2, j, a, b) : 0.0;
		}
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int i = threadIdx.x + 256 * k, a = i % N, b = i / N;
			double    t = f[k];
			if (vy[k] != 0.0) t = t + vy[k];
			if (v[k] != 0.0) t = t + v[k];
			if (vy[k] != 0.0 || v[k] != 0.0) cv[a + N * b + NN * coord[j]] = t;
		}
	}
}

// k_fcorr_gather3d, round 6. The kernel above walks child -> face kind / source -> f6 offset -> values for EVERY entry in EVERY thread:
// on a refined fine level (no rs6, no table) each entry costs five dependent loads and ~9 load instructions of which 4 fetch data --
// 150 us per launch at 0.97 TB/s on `2refine --divide 3` (profiles/r05_bench_c4_2refine_div3_n1.json) -- and on a uniform one a table
// word per entry. But the walk depends on the (coarse patch, axis, plane, quadrant) only: 48 answers per coarse patch. k_gather_desc3d
// writes them down once per level as descriptors -- what to read (finished sums of a neighbour / a 2x2 sum over a face layer or a
// ghost slot / a copy-through patch's own face, cell by cell), where, and the weight -- and the gather keeps the (at most 12)
// descriptors of its plane in LDS: a thread issues nothing but 16-byte data loads, two adjacent entries at a time. The values,
// the order of the additions and the stores are those of k_fcorr_gather3d (bit-identical; TE_NO_GTAB runs the kernel above).
struct GatherDesc {
	int32_t mode, pad;
	int64_t off; // doubles, inside rs6 / f6 / the ghost planes
	double  w;   // -1/h^2 of the axis
};
enum GatherMode : int32_t { GD_ZERO = 0, GD_RS6 = 1, GD_SUM_F6 = 2, GD_SUM_GHOST = 3, GD_COPY_F6 = 4, GD_COPY_GHOST = 5 };
template <int N>
__global__ void k_gather_desc3d(LevelDev L, const int32_t *__restrict__ child, const int32_t *__restrict__ copy, int has_rs6,
                                GatherDesc *__restrict__ gd)
{
	constexpr int NN = N * N, HH = (N / 2) * (N / 2);
	const int     pc = blockIdx.x, e = threadIdx.x;
	if (e >= 48) return;
	const int  ax = e / 16, j = (e / 4) % 4, oa = e & 1, ob = (e >> 1) & 1;
	GatherDesc d;
	d.mode = GD_ZERO, d.pad = 0, d.off = 0, d.w = 0.0;
	if (copy && copy[pc]) { // the coarse patch IS a fine patch that does not coarsen: its own two faces of the axis, cell by cell (w g)
		if (j == 0 || j == 3) {
			const int p = child[(size_t) pc * 8], sp = 2 * ax + (j == 3);
			if (p >= 0) {
				const int kind = L.face_kind[(size_t) p * 6 + sp], src = L.face_src[(size_t) p * 6 + sp];
				if (kind >= FACE_LOCAL) {
					d.w    = -L.rh2[(size_t) p * 3 + ax];
					d.mode = kind == FACE_GHOST ? GD_COPY_GHOST : GD_COPY_F6;
					d.off  = kind == FACE_GHOST ? (int64_t) src * NN : (int64_t) f6Face<N>(L.f6off, src, sp ^ 1);
				}
			}
		}
	} else {
		const int a0 = (ax == 0) ? 1 : 0, a1 = (ax == 2) ? 1 : 2;
		const int s = 2 * ax + (j & 1), hi = j >> 1;
		const int p = child[(size_t) pc * 8 + ((hi << ax) | (oa << a0) | (ob << a1))];
		if (p >= 0) {
			const int kind = L.face_kind[(size_t) p * 6 + s], src = L.face_src[(size_t) p * 6 + s];
			if (kind == FACE_LOCAL && has_rs6) {
				d.mode = GD_RS6;
				d.off  = ((int64_t) src * 6 + (s ^ 1)) * HH;
			} else if (kind >= FACE_LOCAL) {
				d.w    = -L.rh2[(size_t) p * 3 + ax];
				d.mode = kind == FACE_GHOST ? GD_SUM_GHOST : GD_SUM_F6;
				d.off  = kind == FACE_GHOST ? (int64_t) src * NN : (int64_t) f6Face<N>(L.f6off, src, s ^ 1);
			}
		}
	}
	gd[(size_t) pc * 48 + e] = d;
}
// the terms of the two adjacent entries (a, b), (a + 1, b) of a plane, a even (they share a quadrant)
template <int N>
__device__ __forceinline__ double2 gatherPair(const GatherDesc &d, int a, int b, const double *__restrict__ rs6, const double *__restrict__ f6,
                                              const double *__restrict__ ghost)
{
	constexpr int H = N / 2;
	const int     oa = a >= H, ob = b >= H, ha = a - oa * H, hb = b - ob * H;
	if (d.mode == GD_ZERO) return double2{0.0, 0.0};
	if (d.mode == GD_RS6) return *reinterpret_cast<const double2 *>(rs6 + d.off + ha + H * hb);
	const double *gp = ((d.mode == GD_SUM_GHOST || d.mode == GD_COPY_GHOST) ? ghost : f6) + d.off;
	if (d.mode == GD_COPY_F6 || d.mode == GD_COPY_GHOST) {
		const double2 g = *reinterpret_cast<const double2 *>(gp + a + N * b);
		return double2{d.w * g.x, d.w * g.y};
	}
	// the sum of k_restrict_fixup3d over the 2x2 block, first face coordinate fastest
	const double2 r00 = *reinterpret_cast<const double2 *>(gp + 2 * ha + N * (2 * hb)), r10 = *reinterpret_cast<const double2 *>(gp + 2 * ha + 2 + N * (2 * hb));
	const double2 r01 = *reinterpret_cast<const double2 *>(gp + 2 * ha + N * (2 * hb + 1)), r11 = *reinterpret_cast<const double2 *>(gp + 2 * ha + 2 + N * (2 * hb + 1));
	double2       v   = double2{0.0, 0.0};
	v.x += (d.w * r00.x) / 8, v.x += (d.w * r00.y) / 8, v.x += (d.w * r01.x) / 8, v.x += (d.w * r01.y) / 8;
	v.y += (d.w * r10.x) / 8, v.y += (d.w * r10.y) / 8, v.y += (d.w * r11.x) / 8, v.y += (d.w * r11.y) / 8;
	return v;
}
template <int N>
__global__ __launch_bounds__(256) void k_fcorr_gather3d_v2(LevelDev L, const GatherDesc *__restrict__ gd, const double *__restrict__ rs6,
                                                           const double *__restrict__ f6, double *__restrict__ coarse, double *__restrict__ fcorr)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	constexpr int NPAIR = NN / 2, QN = (NPAIR + 255) / 256; // pairs of adjacent entries per plane / per thread
	constexpr int coord[4] = {0, H - 1, H, N - 1};
	const int     pc = blockIdx.x / 12, blk = blockIdx.x % 12, j = blk & 3; // 0..3: the x planes; 4..7: the y planes; 8..11: the z planes
	__shared__ GatherDesc sd[12];
	if (threadIdx.x < 12) {
		const int q = threadIdx.x;
		if (q < 4)
			sd[q] = gd[(size_t) pc * 48 + (blk >> 2) * 16 + j * 4 + q];
		else if (blk >= 8) // the y terms of the rows of this z plane that lie on a y plane too: plane jy, quadrant (x half, this plane's z half)
			sd[q] = gd[(size_t) pc * 48 + 16 + ((q - 4) >> 1) * 4 + ((q - 4) & 1) + 2 * (j >> 1)];
	}
	__syncthreads();
	// a plane none of whose descriptors has anything to add (the inner octant planes of a copy-through patch, faces on the physical
	// boundary): nothing to read, and nothing to write -- the side array is zeroed when the descriptors are built, and only this
	// kernel writes it afterwards, the same entries every time
	bool any = false;
#pragma unroll
	for (int q = 0; q < 12; q++) any = any || (q < (blk >= 8 ? 12 : 4) && sd[q].mode != GD_ZERO);
	if (!any) return;
	double  *cv = coarse + (size_t) pc * NNN;
	double2 v[QN], vy[QN], f[QN];
	if (blk < 4) { // an x plane: into the side array
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int i = 2 * (threadIdx.x + 256 * k), a = i % N, b = i / N;
			v[k]        = i < NN ? gatherPair<N>(sd[(a >= H) + 2 * (b >= H)], a, b, rs6, f6, L.ghost) : double2{0.0, 0.0};
		}
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int i = 2 * (threadIdx.x + 256 * k);
			if (i < NN) *reinterpret_cast<double2 *>(fcorr + ((size_t) pc * 4 + j) * NN + i) = v[k];
		}
	} else if (blk < 8) { // a y plane: entry (x, z) -> cell (x, coord[j], z); the cells that also lie on a z plane are that plane's
#pragma clang fp contract(off)
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int  i = 2 * (threadIdx.x + 256 * k), a = i % N, b = i / N;
			const bool in = i < NN && octPlane<N>(b) < 0;
			v[k]          = in ? gatherPair<N>(sd[(a >= H) + 2 * (b >= H)], a, b, rs6, f6, L.ghost) : double2{0.0, 0.0};
			f[k]          = in ? *reinterpret_cast<const double2 *>(cv + a + N * coord[j] + NN * b) : double2{0.0, 0.0};
		}
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int i = 2 * (threadIdx.x + 256 * k), a = i % N, b = i / N;
			double2   t = f[k];
			if (v[k].x != 0.0) t.x = t.x + v[k].x;
			if (v[k].y != 0.0) t.y = t.y + v[k].y;
			if (v[k].x != 0.0 || v[k].y != 0.0) *reinterpret_cast<double2 *>(cv + a + N * coord[j] + NN * b) = t;
		}
	} else { // a z plane: entry (x, y) -> cell (x, y, coord[j]); a cell that lies on a y plane too takes its y term first
#pragma clang fp contract(off)
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int  i = 2 * (threadIdx.x + 256 * k), a = i % N, b = i / N, jy = octPlane<N>(b);
			const bool in = i < NN;
			f[k]          = in ? *reinterpret_cast<const double2 *>(cv + a + N * b + NN * coord[j]) : double2{0.0, 0.0};
			vy[k]         = (in && jy >= 0) ? gatherPair<N>(sd[4 + 2 * jy + (a >= H)], a, coord[j], rs6, f6, L.ghost) : double2{0.0, 0.0};
			v[k]          = in ? gatherPair<N>(sd[(a >= H) + 2 * (b >= H)], a, b, rs6, f6, L.ghost) : double2{0.0, 0.0};
		}
#pragma unroll
		for (int k = 0; k < QN; k++) {
			const int i = 2 * (threadIdx.x + 256 * k), a = i % N, b = i / N;
			double2   t = f[k];
			if (vy[k].x != 0.0) t.x = t.x + vy[k].x;
			if (vy[k].y != 0.0) t.y = t.y + vy[k].y;
			if (v[k].x != 0.0) t.x = t.x + v[k].x;
			if (v[k].y != 0.0) t.y = t.y + v[k].y;
			if (vy[k].x != 0.0 || vy[k].y != 0.0 || v[k].x != 0.0 || v[k].y != 0.0) *reinterpret_cast<double2 *>(cv + a + N * b + NN * coord[j]) = t;
		}
	}
}

// ---- pre-smoothing sweep from a zero iterate + residual + restriction in one pass (opts.fuse = 2) --------------
// Cycle.h:57-65 for the first sweep of a cycle: u = S(0, f); coarse f = AvgRstr(f - A u). The sweep of a patch
// needs no neighbour data at all (every ghost of a zero iterate is zero), and the residual of every cell that
// is not on a patch face with a neighbour needs only this patch's new values: k_rbgs_zero_resid3d therefore
// writes u (8 B/site) and the restricted residual (1 B/site) from one read of f (8 B/site) -- 17 B/site
// instead of 16 + 17 for the sweep and the residual+restrict kernel. On faces with a neighbour the residual
// is formed with a zero ghost; k_restrict_fixup3d then adds the missing term -g/h^2 (g = the neighbour's new
// face value) to the coarse cells along the patch faces, from the face layers alone (x faces from the compact
// columns the sweep exported). That is the one fusion that is NOT bit-identical to the unfused sequence: the
// ghost term enters the coarse value by a separate addition (differences of a few ulp in coarse f along
// patch faces; sharded runs still equal single-rank runs bit for bit). Levels of octant children with local
// parents and no coarse/fine faces only; z-slabs do not apply (the residual would need two more planes).
//
// Pipeline per step z: red update of plane z (f only), black update of plane z-1 (new red values), residual of
// plane z-2 (its z-neighbours z-3 and z-1 are final), restriction over plane pairs. Four LDS planes rotate:
// plane z-2 is still being read (residual) while the fastest waves already write plane z+1.
// STORE_U = false (opts.fuse = 3): the new iterate is not stored at all, only its six face layers (L.f6_out): the
// post-smoothing kernel k_rbgs_resweep_prolong3d recomputes it from f, everything else reads faces.
// EXPORT: the ghost terms of the restricted residual that this patch's face values contribute to its neighbours' coarse
// cells go to rd.fcorr (see above) -- no k_restrict_fixup3d pass; FCORR: this level's own right-hand side carries such
// terms in L.fcorr (it was produced that way by the finer level).
// AH: how many planes ahead of the red update the right-hand side is requested (1: the plane of the next step only -- then
// every step waits for a load issued one step earlier, and a step is shorter than an HBM miss under load; 3: the default).
// FS (te_bicgstab, level 0): the right-hand side of the cycle is itself a vector statement of the Krylov loop that nobody has
// executed yet -- FS = 1: s = resid + ap * (-alpha) (BiCGStab.h:79-80); FS = 2: p = beta (p + ap * (-omega)) + resid (:99-100).
// This kernel, the first reader, forms it from its operands (the expressions of k_bicg_s / k_bicg_p, explicit FMAs in both
// places: bit-identical) and stores it once for the readers that follow (the post-sweep, the dot products): 16 / 24 B/site
// in this pass instead of 24 / 32 in a pass of their own plus the 8 this kernel reads anyway. The operands of a plane are
// requested at the usual distance and combined one step later (not in the step that requests them: that would wait for
// loads just issued), so they add one raw stage to the ring of planes in flight.
struct FSrc {
	const double *a, *b, *c; // FS = 1: resid, ap, -; FS = 2: p, ap, resid
	double       *out;       // s, or p (in place: every thread rewrites exactly the cells it has read)
	double        s1, s2;    // -alpha, - ; -omega, beta
};
template <int FS> __device__ __forceinline__ double2 fsrcCombine(const FSrc &fs, double2 a, double2 b, double2 c)
{
	if (FS == 1) return double2{__builtin_fma(b.x, fs.s1, a.x), __builtin_fma(b.y, fs.s1, a.y)};
	const double tx = __builtin_fma(b.x, fs.s1, a.x), ty = __builtin_fma(b.y, fs.s1, a.y);
	return double2{__builtin_fma(fs.s2, tx, c.x), __builtin_fma(fs.s2, ty, c.y)};
}
template <int N, bool STORE_U, bool EXPORT = false, bool FCORR = false, int AH = 4, int FS = 0>
__global__ __launch_bounds__(Tile3<N>::TPB, (TE_ZR_WIDE && EXPORT && (FS == 2 || FCORR)) ? 2 : 3) void k_rbgs_zero_resid3d(LevelDev L, const double *__restrict__ f,
                                                                     double *__restrict__ out, RestrictDst rd, FSrc fs = FSrc())
{
	static_assert(FS == 0 || (!FCORR && !STORE_U && AH >= 2), "the fused right-hand sides exist for the level-0 path of te_bicgstab");
	using T           = Tile3<N>;
	constexpr int TPB = T::TPB, NP = T::NP, H = T::H;
	constexpr int NN  = N * N, NNN = N * N * N;
	const int     slot = xcdRemap(blockIdx.x, L.count);
	if (slot >= L.count) return;
	const int pid = L.order ? L.order[L.first + slot] : L.first + slot;
	const int tid = threadIdx.x;

	__shared__ __attribute__((aligned(16))) double tile[4][T::LSZ];
	__shared__ double idiag[27];

	const int32_t *fk  = L.face_kind + (size_t) pid * 6;
	const double   rhx = L.rh2[(size_t) pid * 3], rhy = L.rh2[(size_t) pid * 3 + 1], rhz = L.rh2[(size_t) pid * 3 + 2];
	const double2 *fp2 = reinterpret_cast<const double2 *>(f + (size_t) pid * NNN);
	double2       *op2 = reinterpret_cast<double2 *>(out + (size_t) pid * NNN);

	FaceKinds kinds(fk);
	idiagTable(idiag, tid, kinds, rhx, rhy, rhz);
	for (int i = tid; i < 4 * T::LSZ; i += TPB) (&tile[0][0])[i] = 0.0; // halo ring stays zero: ghosts of a zero iterate
	// EXPORT: the 2x2 sums of this patch's six face layers, [6][H*H] next to the patch's other face data
	double *const rs = EXPORT ? rd.rs6 + (size_t) pid * 6 * (H * H) : nullptr;
	double        eX = 0.0, eY = 0.0; // sums in progress on this thread's x face / y face (the z pair spans two steps)

	const bool act = (T::NT == TPB) || tid < T::NT;
	const int  X = act ? tid % H : 0, Yp = act ? tid / H : 0;
	int        q[2], lds[2], dix[2][2];
	const int  ldo[2] = {T::row(2 * Yp) + 2 * X + 2, T::row(2 * Yp + 3) + 2 * X + 2}; // the rows below / above the pair
#pragma unroll
	for (int k = 0; k < 2; k++) {
		q[k]   = (2 * Yp + k) * H + X;
		lds[k] = T::row(2 * Yp + k + 1) + 2 * X + 2;
	}
	{
		const int cy0 = (Yp == 0) ? 0 : 1, cy1 = (Yp == H - 1) ? 2 : 1;
		const int cx0 = (X == 0) ? 0 : 1, cx1 = (X == H - 1) ? 2 : 1;
		dix[0][0] = cx0 + 3 * cy0, dix[0][1] = cx1 + 3 * cy0, dix[1][0] = cx0 + 3 * cy1, dix[1][1] = cx1 + 3 * cy1;
	}
	// which x face (0 W, 1 E) and which y face (0 S, 1 N) this thread's cells lie on, or -1 (at most one of each: H >= 2),
	// and where its entries of plane 0 go in the face layers / in the 2x2 sums
	const int     xs  = !act ? -1 : (X == 0 ? 0 : (X == H - 1 ? 1 : -1)), ys = !act ? -1 : (Yp == 0 ? 0 : (Yp == H - 1 ? 1 : -1));
	// (the face layers' places come from a table: the ones other ranks need sit in send order, LevelDev.f6off; the two z
	// layers' places are fetched here, not in the step that stores them -- a load under a branch inside the march makes every
	// step wait where the branch rejoins)
	double *const xfo = STORE_U ? nullptr : L.f6_out + f6Face<N>(L.f6off, pid, xs > 0 ? 1 : 0) + 2 * Yp;
	double *const yfo = STORE_U ? nullptr : L.f6_out + f6Face<N>(L.f6off, pid, 2 + (ys > 0 ? 1 : 0)) + 2 * X;
	double *const zfo[2] = {STORE_U ? nullptr : L.f6_out + f6Face<N>(L.f6off, pid, 4) + 2 * X + N * (2 * Yp),
	                        STORE_U ? nullptr : L.f6_out + f6Face<N>(L.f6off, pid, 5) + 2 * X + N * (2 * Yp)};
	double *const xrs = EXPORT ? rs + (xs > 0 ? H * H : 0) + Yp : nullptr;
	double *const yrs = EXPORT ? rs + (2 + (ys > 0 ? 1 : 0)) * (H * H) + X : nullptr;
	// ghost of the residual's stencil on each side, as a multiple of the cell just inside: -1 Dirichlet, +1 Neumann
	// (StarPatchOp.h:39-65); 0 on faces with a neighbour (that term is k_restrict_fixup3d's)
	const double gW = (X == 0) ? kinds.phys(0) : 0.0, gE = (X == H - 1) ? kinds.phys(1) : 0.0;
	const double gS = (Yp == 0) ? kinds.phys(2) : 0.0, gN = (Yp == H - 1) ? kinds.phys(3) : 0.0;
	const double gB = kinds.phys(4), gT = kinds.phys(5);

	// fused restriction target (see k_stencil3d<MODE_RESID_RESTRICT>): the parent's octant, or the block this rank
	// ships to the parent's rank; no copy-through patches on these levels
	const int pa = rd.parent[pid], rorth = rd.orth[pid];
	double   *rdst = nullptr;
	double2  *rcpy = nullptr; // a patch that copies through (orth < 0): the residual itself goes to the same-size coarse patch
	int       rsz  = 0;
	if (rorth < 0) {
		rcpy = reinterpret_cast<double2 *>(pa >= 0 ? rd.coarse + (size_t) pa * NNN : rd.remote + rd.remote_off[-(pa + 2)]);
	} else if (pa >= 0) {
		rdst = rd.coarse + (size_t) pa * NNN + ((rorth & 1) ? H : 0) + N * ((rorth & 2) ? H : 0) + NN * ((rorth & 4) ? H : 0) + X + N * Yp;
		rsz  = NN;
	} else {
		rdst = rd.remote + rd.remote_off[-(pa + 2)] + X + H * Yp;
		rsz  = H * H;
	}
	double racc = 0.0;

	// new iterate: u3 = plane z-3, u2 = z-2, u1 = z-1, u0 = z (all start from zero); right-hand sides alongside
	static_assert(AH == 1 || AH == 2 || AH == 4, "planes of f in flight: the loop is unrolled over the ring's slots");
	double2 u3[2], u2[2], u1[2], u0[2], f2[2], f1[2], f0[2];
	// planes z .. z+AH-1 at the start of step z, plane p in slot p % AH. A slot is read (into f0) by the step that consumes its
	// plane and re-requested by the same step; NO register of a plane in flight is ever copied: a ring that rotates through
	// register moves makes every step wait for its newest load (the move reads it), whatever distance the source asks for.
	double2 fa[AH][2];
	const double2 zero2 = double2{0.0, 0.0};
	FCorrSrc<N>   fc;
	double2       cca[AH][FCorrSrc<N>::NX]; // FCORR: the x terms of the planes in fa (added when a plane becomes f0)
	if (FCORR) fc.init(L.fcorr, pid, X, Yp);
	// FS: the operands of the plane requested last (plane z + AH - 1 at the start of step z), and where the combined plane goes
	const double2 *sa2 = FS ? reinterpret_cast<const double2 *>(fs.a + (size_t) pid * NNN) : nullptr;
	const double2 *sb2 = FS ? reinterpret_cast<const double2 *>(fs.b + (size_t) pid * NNN) : nullptr;
	const double2 *sc2 = FS == 2 ? reinterpret_cast<const double2 *>(fs.c + (size_t) pid * NNN) : nullptr;
	double2       *so2 = FS ? reinterpret_cast<double2 *>(fs.out + (size_t) pid * NNN) : nullptr;
	double2        ra[2], rb[2], rc[2];
	auto           rawLoad = [&](int z) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            ra[k] = sa2[z * NP + q[k]];
            rb[k] = sb2[z * NP + q[k]];
            rc[k] = FS == 2 ? sc2[z * NP + q[k]] : zero2;
        }
	};
	auto rawCombineStore = [&](int z, double2(&dst)[2]) { // plane z of the right-hand side from the raw operands; stored once
#pragma unroll
		for (int k = 0; k < 2; k++) {
			dst[k] = fsrcCombine<FS>(fs, ra[k], rb[k], rc[k]);
			if (act && z < N) so2[z * NP + q[k]] = dst[k];
		}
	};
#pragma unroll
	for (int k = 0; k < 2; k++) {
		u3[k] = u2[k] = u1[k] = u0[k] = zero2;
		f2[k] = f1[k] = f0[k] = zero2;
	}
	if constexpr (FS != 0) { // planes 0 .. AH-2 combined here (a one-time wait), plane AH-1 left raw for step 0
#pragma unroll
		for (int a = 0; a + 1 < AH; a++) {
			rawLoad(a < N ? a : N - 1);
			rawCombineStore(a, fa[a]);
		}
		rawLoad(AH - 1 < N ? AH - 1 : N - 1);
	} else {
#pragma unroll
		for (int a = 0; a < AH; a++) { // planes 0 .. AH-1 (step z requests plane z+AH)
			const int za = (a < N) ? a : N - 1;
#pragma unroll
			for (int k = 0; k < 2; k++) fa[a][k] = ldStream<TE_ZR_NT != 0>(fp2 + za * NP + q[k]);
			if (FCORR) fc.load(za, cca[a]);
			// in THIS order: the waits of the loop are derived from the oldest state that reaches its head, and a scheduler that
			// issues slot 0 last here makes every iteration wait for all but the last loads
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	__syncthreads(); // idiag and the zeroed tiles

	int bz = 0; // z % 4
	auto step = [&](auto zpar, auto slot_, int z) {
		constexpr int ZPAR = decltype(zpar)::value, SLOT = decltype(slot_)::value; // z % 2, z % AH
		const int     zc   = (z + AH < N) ? z + AH : N - 1;
#pragma unroll
		for (int k = 0; k < 2; k++) { // plane z leaves the ring (requested AH steps ago) ...
			f2[k] = f1[k];
			f1[k] = f0[k];
			f0[k] = takeRegs(fa[SLOT][k]);
		}
		if (FCORR) fc.apply(f0, cca[SLOT]);
		if constexpr (FS != 0) { // ... the plane requested in the previous step takes its slot, the next one is requested
			rawCombineStore(z + AH - 1, fa[(SLOT + AH - 1) % AH]);
			rawLoad(zc);
		} else { // ... and its slot is requested again
#pragma unroll
			for (int k = 0; k < 2; k++) fa[SLOT][k] = ldStream<TE_ZR_NT != 0>(fp2 + zc * NP + q[k]);
		}
		if (FCORR) fc.load(zc, cca[SLOT]);
		double *tz = tile[bz];            // plane z
		double *t1 = tile[(bz + 3) & 3];  // plane z-1
		double *t2 = tile[(bz + 2) & 3];  // plane z-2
		ldsBarrier(); // plane z's buffer was last read (residual of plane z-4) two steps ago
		if (z < N) { // red cells of plane z: every neighbour is zero
			const int cz9 = (z == 0) ? 0 : (z == N - 1 ? 18 : 9);
			relaxCell<N, 0, (0 + ZPAR) & 1, true>(tz, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, u0, u1, u1, f0);
			relaxCell<N, 1, (1 + ZPAR) & 1, true>(tz, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, u0, u1, u1, f0);
			// (the buffer still holds an older plane at the black cells: nobody reads them before step z+1 rewrites them)
		}
		if (z > 0 && z <= N) { // black cells of plane z-1 from the new red values
			const int cz9 = (z - 1 == 0) ? 0 : (z - 1 == N - 1 ? 18 : 9);
			relaxCell<N, 0, (1 + 0 + 1 - ZPAR) & 1, false>(t1, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, u1, u2, u0, f1);
			relaxCell<N, 1, (1 + 1 + 1 - ZPAR) & 1, false>(t1, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, u1, u2, u0, f1);
			if (act && STORE_U) {
				op2[(z - 1) * NP + q[0]] = u1[0];
				op2[(z - 1) * NP + q[1]] = u1[1];
				if (L.xf_out) {
					double *xo = L.xf_out + (size_t) pid * 2 * NN + N * (z - 1) + 2 * Yp;
					if (X == 0) *reinterpret_cast<double2 *>(xo) = double2{u1[0].x, u1[1].x};
					if (X == H - 1) *reinterpret_cast<double2 *>(xo + NN) = double2{u1[0].y, u1[1].y};
				}
			}
			if (!STORE_U || EXPORT) {
				// The six face layers (W,E at y + N z; S,N at x + N z; B,T at x + N y) and, EXPORT, this plane's face values as
				// ghost terms of the neighbours' restricted residuals (k_restrict_fixup3d's sum: (w g)/8 over the 2x2 block,
				// first face coordinate fastest). A thread lies on at most one x face and one y face: one predicated region
				// per axis serves both sides (value and address chosen per lane) and both jobs -- the kernel is bound by
				// instruction issue, and eight separately predicated regions per plane were a sixth of its instructions.
				const int  zz = z - 1, zh = zz >> 1;
				constexpr bool second = !ZPAR; // zz & 1
				auto       pair = [&](double &e, double a, double b, double w) { // two cells of this plane join the block's sum
                    double t = second ? e : 0.0;
                    t += (w * a) / 8;
                    t += (w * b) / 8;
                    e = t;
				};
				constexpr int HH = H * H;
				if (xs >= 0) { // x faces: block = rows (2Yp, 2Yp+1) x planes (zz, zz+1); entry (y, z)
					const double2 xv = xs == 1 ? double2{u1[0].y, u1[1].y} : double2{u1[0].x, u1[1].x};
					if (!STORE_U) *reinterpret_cast<double2 *>(xfo + N * zz) = xv;
					if (EXPORT) {
						pair(eX, xv.x, xv.y, -rhx);
						if (second) xrs[H * zh] = eX;
					}
				}
				if (ys >= 0) { // y faces: block = cells (2X, 2X+1) x planes; entry (x, z)
					const double2 yv = ys == 1 ? u1[1] : u1[0];
					if (!STORE_U) *reinterpret_cast<double2 *>(yfo + N * zz) = yv;
					if (EXPORT) {
						pair(eY, yv.x, yv.y, -rhy);
						if (second) yrs[H * zh] = eY;
					}
				}
				if (act && (zz == 0 || zz == N - 1)) { // z faces: the whole 2x2 block is in this thread; entry (x, y)
					if (!STORE_U) {
						double *zo = zfo[zz == 0 ? 0 : 1];
						*reinterpret_cast<double2 *>(zo)     = u1[0];
						*reinterpret_cast<double2 *>(zo + N) = u1[1];
					}
					if (EXPORT) {
						double t = 0.0;
						t += (-rhz * u1[0].x) / 8, t += (-rhz * u1[0].y) / 8, t += (-rhz * u1[1].x) / 8, t += (-rhz * u1[1].y) / 8;
						rs[(zz == 0 ? 4 : 5) * HH + X + H * Yp] = t;
					}
				}
			}
		}
		if (z > 1) { // residual of plane zr = z-2 (needs the black cells of this plane, written one step ago by all
			// waves, and of plane z-1, just written by this thread's own registers) and its restriction.
			// RED cells only: a black cell was relaxed last, from exactly the values its residual is formed with (the red
			// neighbours of the finished planes; zero ghosts on faces with a neighbour, whose terms come later; physical
			// faces folded into its diagonal the same way), so its residual is f - (f + diag v - diag v) = 0 up to the rounding
			// of that one update -- it enters the restriction as exactly 0. Half the stencil work of this block, and red
			// cells read only their own column of the rows above and below.
			const int zr = z - 2;
			// the black values of plane z-2 other waves wrote in the previous step are visible (barrier above)
			double a = ZPAR ? racc : 0.0; // zr and z have the same parity
			auto redResid = [&](auto kk, auto cr) {
				constexpr int K = decltype(kk)::value, CR = decltype(cr)::value; // row of the pair; which of its cells is red in plane zr
				const double *t0 = t2 + lds[K];
				const double2 c  = u2[K];
				const double  cc = CR ? c.y : c.x;
				const double  outer = t2[ldo[K] + CR];                                    // row y-1 (K = 0) / y+2 (K = 1): LDS
				const double  inner = CR ? u2[1 - K].y : u2[1 - K].x;                     // the other row of the pair: a register
				double        ym = (K == 0) ? outer : inner, yp = (K == 0) ? inner : outer;
				if (K == 0) ym += gS * cc;
				if (K == 1) yp += gN * cc;
				const double below = (zr == 0) ? gB * cc : (CR ? u3[K].y : u3[K].x);
				const double above = (zr == N - 1) ? gT * cc : (CR ? u1[K].y : u1[K].x);
				double       lap;
				if (CR == 0) {
					const double xl = t0[-1] + gW * c.x;
					lap             = (xl - 2 * c.x + c.y) * rhx;
				} else {
					const double xr = t0[2] + gE * c.y;
					lap             = (c.x - 2 * c.y + xr) * rhx;
				}
				lap += (ym - 2 * cc + yp) * rhy;
				lap += (below - 2 * cc + above) * rhz;
				const double r = (CR ? f2[K].y : f2[K].x) - lap;
				if (rcpy && act) rcpy[zr * NP + q[K]] = CR ? double2{0.0, r} : double2{r, 0.0}; // (wave-uniform) AvgRstr.h:103-107
				a += r / 8; // AvgRstr.h:95-102 order: x, then y, then z; each /2^D first (the black cells add exactly 0)
			};
			redResid(std::integral_constant<int, 0>{}, std::integral_constant<int, (0 + ZPAR) & 1>{});
			redResid(std::integral_constant<int, 1>{}, std::integral_constant<int, (1 + ZPAR) & 1>{});
			racc = a;
			if (rdst && ZPAR && act) rdst[rsz * (zr >> 1)] = a;
		}
#pragma unroll
		for (int k = 0; k < 2; k++) {
			u3[k] = u2[k];
			u2[k] = u1[k];
			u1[k] = u0[k];
			u0[k] = zero2;
		}
		bz = (bz + 1) & 3;
	};
	using S0 = std::integral_constant<int, 0>;
	using S1 = std::integral_constant<int, 1>;
	using A1 = std::integral_constant<int, 1 % AH>;
	using A2 = std::integral_constant<int, 2 % AH>;
	using A3 = std::integral_constant<int, 3 % AH>;
	static_assert(N % 4 == 0, "the march is unrolled over up to four ring slots");
#pragma unroll 1
	for (int z = 0; z < N; z += (AH == 4 ? 4 : 2)) {
		step(S0{}, S0{}, z);
		step(S1{}, A1{}, z + 1);
		if constexpr (AH == 4) {
			step(S0{}, A2{}, z + 2);
			step(S1{}, A3{}, z + 3);
		}
	}
	step(S0{}, S0{}, N);     // black of plane N-1, residual of plane N-2
	step(S1{}, A1{}, N + 1); // residual of plane N-1
}

// Second half of the fusion above: the ghost terms of the residual along patch faces with a neighbour. One
// workgroup per fine patch; per face, one thread per coarse face cell adds -(1/h^2)/8 times the four
// neighbour values behind its 2x2 fine cells to the coarse cell (a coarse cell belongs to exactly one fine
// patch; faces in the fixed order W,E,S,N,B,T with a barrier in between: deterministic and independent of the
// partition). u = the new iterate (its ghost slots current); L.xf = its compact x faces, or L.f6 = all six face
// layers when u itself was never stored (either may be null).
// OWN: the residual that is being completed was formed with the PATCH operator (exact patch solves: interface
// faces closed as homogeneous Dirichlet, ghost = -m, StarPatchOp.h:204-319) rather than with a zero ghost, so
// the missing term is -(g + m)/h^2 = -2 gamma/h^2 with m = this patch's own face value.
template <int N, bool OWN>
__global__ __launch_bounds__(256) void k_restrict_fixup3d(LevelDev L, const double *__restrict__ u, RestrictDst rd)
{
	constexpr int NN = N * N, NNN = N * N * N, H = N / 2;
	const int     p  = L.first + blockIdx.x; // (all patches: this kernel runs after the exchange)
	const int     pa = rd.parent[p], o = rd.orth[p];
	// the octant of the parent, or the H^3 block that travels to the parent's rank: base and strides
	double   *cb;
	int       c1 = 1, c2, c3;
	if (pa >= 0) {
		cb = rd.coarse + (size_t) pa * NNN + ((o & 1) ? H : 0) + N * ((o & 2) ? H : 0) + NN * ((o & 4) ? H : 0);
		c2 = N, c3 = NN;
	} else {
		cb = rd.remote + rd.remote_off[-(pa + 2)];
		c2 = H, c3 = H * H;
	}
	const int cs[3] = {c1, c2, c3};
	if (o < 0) { // the patch copies through: every face cell's term goes to the same cell of the same-size coarse patch
		double *cc = pa >= 0 ? rd.coarse + (size_t) pa * NNN : rd.remote + rd.remote_off[-(pa + 2)];
		for (int so = 0; so < 6; so++) {
			const int s = (so + 2) % 6; // faces in the order S, N, B, T, W, E (see FCorrSrc)
			const int kind = L.face_kind[(size_t) p * 6 + s], src = L.face_src[(size_t) p * 6 + s];
			if (kind >= FACE_LOCAL) {
				const int    ax = s >> 1;
				const int    sa = (ax == 0) ? N : 1, sb = (ax == 2) ? N : NN, sn = (ax == 0) ? 1 : (ax == 1 ? N : NN);
				const int    mine = (s & 1) ? (N - 1) * sn : 0, oth = (s & 1) ? 0 : (N - 1) * sn;
				const double w = -L.rh2[(size_t) p * 3 + ax];
				for (int i = threadIdx.x; i < NN; i += blockDim.x) {
					const int cell = (i % N) * sa + (i / N) * sb;
					double    g;
					if (kind == FACE_GHOST)
						g = L.ghost[(size_t) src * NN + i];
					else if (L.f6) // the iterate exists only as its face layers
						g = L.f6[f6Face<N>(L.f6off, src, s ^ 1) + i];
					else if (ax == 0 && L.xf) // compact x-face columns instead of a stride-N gather
						g = L.xf[((size_t) src * 2 + ((s & 1) ^ 1)) * NN + i];
					else
						g = u[(size_t) src * NNN + oth + cell];
					if (OWN)
						g += L.f6 ? L.f6[f6Face<N>(L.f6off, p, s) + i]
						          : ((ax == 0 && L.xf) ? L.xf[((size_t) p * 2 + (s & 1)) * NN + i] : u[(size_t) p * NNN + mine + cell]);
					cc[mine + cell] += w * g;
				}
			}
			__syncthreads();
		}
		return;
	}
	for (int so = 0; so < 6; so++) {
		const int s = (so + 2) % 6; // S, N, B, T, W, E
		const int kind = L.face_kind[(size_t) p * 6 + s], src = L.face_src[(size_t) p * 6 + s];
		if (kind >= FACE_LOCAL) {
			const int    ax = s >> 1;
			const int    sa = (ax == 0) ? N : 1, sb = (ax == 2) ? N : NN, sn = (ax == 0) ? 1 : (ax == 1 ? N : NN);
			const int    oth = (s & 1) ? 0 : (N - 1) * sn;             // the neighbour's facing layer
			const double w   = -L.rh2[(size_t) p * 3 + ax];
			// (round 6) the neighbour's FINISHED sum where the pre-sweep exported it (rd.rs6: the same products added in the same order
			// by the patch that owns the values, k_rbgs_zero_resid3d EXPORT): a quarter of the bytes of the face layer
			const double *fin = (!OWN && rd.rs6 && kind == FACE_LOCAL) ? rd.rs6 + ((size_t) src * 6 + (s ^ 1)) * (H * H) : nullptr;
			for (int i = threadIdx.x; i < H * H; i += blockDim.x) {
				const int ha = i % H, hb = i / H;
				double    acc = 0.0;
				if (fin) acc = fin[i];
#pragma unroll
				for (int db = 0; db < 2 && !fin; db++)
#pragma unroll
					for (int da = 0; da < 2; da++) {
						const int a = 2 * ha + da, b = 2 * hb + db;
						double    g;
						if (kind == FACE_GHOST)
							g = L.ghost[(size_t) src * NN + a + N * b];
						else if (L.f6)
							g = L.f6[f6Face<N>(L.f6off, src, s ^ 1) + a + N * b];
						else if (ax == 0 && L.xf)
							g = L.xf[((size_t) src * 2 + ((s & 1) ^ 1)) * NN + a + N * b];
						else
							g = u[(size_t) src * NNN + oth + a * sa + b * sb];
						if (OWN) {
							if (L.f6)
								g += L.f6[f6Face<N>(L.f6off, p, s) + a + N * b];
							else if (ax == 0 && L.xf)
								g += L.xf[((size_t) p * 2 + (s & 1)) * NN + a + N * b];
							else
								g += u[(size_t) p * NNN + ((s & 1) ? (N - 1) * sn : 0) + a * sa + b * sb];
						}
						acc += (w * g) / 8;
					}
				// coarse cell (ha, hb) of the face: axes in order, the normal index 0 or H-1
				const int ca = (ax == 0) ? 1 : 0, cbx = (ax == 2) ? 1 : 2;
				cb[ha * cs[ca] + hb * cs[cbx] + ((s & 1) ? (H - 1) * cs[ax] : 0)] += acc;
			}
		}
		__syncthreads();
	}
}

// ---- opts.fuse = 3: the post-smoothing sweep recomputes the pre-smoothed iterate instead of reading it ---------
// With one pre- and one post-sweep the iterate between them, v = S(0, f), is a cheap function of f alone (red
// cells: -f/diag; black cells: their six red neighbours), and the post sweep reads f anyway. So
// k_rbgs_zero_resid3d<N, false> stores only v's six face layers (for the neighbours' ghosts, the fix-up and the
// exchange), and this kernel recomputes v plane by plane two planes ahead of the sweep, adds P(coarse) and
// relaxes: read f, 1/8 coarse and the neighbours' face layers, write u -- 18.5 B/site instead of 25, and the
// pre-sweep kernel writes 2.5 instead of 9.5 B/site. v is formed by the same instructions as in the pre-sweep
// kernel and v + P e as in k_rbgs3d<N, false, true>, so the result is bit-identical to opts.fuse = 2.
//
// Step z (one barrier, as in k_rbgs3d): before it, red values of plane z+3 (registers only), the red values
// of plane z+2 into their LDS plane, the iterate of plane z (w = v + P e) into its LDS plane; after it, black
// values of plane z+2 (its z-neighbours z+1 and z+3 are registers of the same lane) -> w(z+2), then the sweep's
// red update of plane z and black update of plane z-1 exactly as in k_rbgs3d.
template <int N>
__global__ void k_pack_faces6_3d(const int32_t *__restrict__ faces, const double *__restrict__ f6, ProlongSrc ps,
                                 double *__restrict__ sendbuf, const int32_t *__restrict__ f6off, PackPush pp = PackPush())
{
	constexpr int NN = N * N, H = N / 2;
	const int     p = faces[2 * blockIdx.x], s = faces[2 * blockIdx.x + 1];
	const int     ax = s >> 1;
	const int     sa = (ax == 0) ? N : 1, sb = (ax == 2) ? N : NN, sn = (ax == 0) ? 1 : (ax == 1 ? N : NN);
	const double *fp = f6 + f6Face<N>(f6off, p, s);
	double       *o  = pp.dst ? pp.dst[blockIdx.x] : sendbuf + (size_t) blockIdx.x * NN;
	if (pp.dst && *pp.err) {
		// (a wait has given up: nothing is stored any more)
	} else if (ps.coarse && ps.orth[p] < 0) { // a patch that copies through (refined level): its correction is the same-size coarse
		// patch, cell by cell -- as k_pack_faces_prolong3d and k_cf_ghost6_3d<N, true> form it for local readers
		const int face = (s & 1) ? (N - 1) * sn : 0;
		for (int i = threadIdx.x; i < NN; i += blockDim.x) o[i] = fp[i] + coarseAtCell<N>(ps, p, face + (i % N) * sa + (i / N) * sb);
	} else {
		const double *cp = ps.coarse ? coarseOctant<N>(ps, p) + ((s & 1) ? (H - 1) * sn : 0) : nullptr;
		for (int i = threadIdx.x; i < NN; i += blockDim.x) {
			const int a = i % N, b = i / N;
			o[i] = cp ? fp[i] + cp[(a / 2) * sa + (b / 2) * sb] : fp[i];
		}
	}
	packPushTail(pp);
}

// V (tuning variants, all bit-identical): bit 0: the black values of the recomputed plane and of the sweep's plane z-1 are
// not written back to LDS (nobody reads them: their x/y neighbours are red); bit 1: the top neighbour's plane is loaded on
// the last steps only; bit 2: three ring slots for the planes of f in flight instead of two (a request is two steps ahead
// of its first use instead of one; costs 8 registers: spills at three workgroups per CU); bit 3 / bit 4: non-temporal loads
// of f / stores of u (a level's vectors are not re-read before they have left the caches: +1.4 % at 512^3, +10 % at 256^3);
// bit 5: two workgroups per CU (256 registers) with four ring slots (with bit 2: three). Defaults: see resweepProlongN.
// FCORR: this level's right-hand side carries ghost terms in L.fcorr (see FCorrSrc)
// CFP: a refined level, as k_rbgs3d<..., CFP>: patches that copy through take their correction cell by cell from the
// same-size coarse patch, coarse/fine ghost slots hold u + P e already (k_cf_ghost6_3d<N, true>)
template <int N, int V = 0, bool FCORR = false, bool CFP = false>
__global__ __launch_bounds__(Tile3<N>::TPB, (V & 32) ? 2 : 3) void k_rbgs_resweep_prolong3d(LevelDev L, const double *__restrict__ f,
                                                                          double *__restrict__ out, ProlongSrc ps)
{
	constexpr bool LDS_ALL = !(V & 1), TG_ALWAYS = !(V & 2), DEEP = (V & 4) != 0, NTL = (V & 8) != 0, NTS = (V & 16) != 0;
	using T           = Tile3<N>;
	constexpr int TPB = T::TPB, NP = T::NP, H = T::H;
	constexpr int NN  = N * N, NNN = N * N * N;
	static_assert(N >= 4, "the pipeline looks three planes ahead");
	const int     slot = xcdRemap(blockIdx.x, L.count);
	if (slot >= L.count) return;
	const int pid = L.order ? L.order[L.first + slot] : L.first + slot;
	const int tid = threadIdx.x;

	__shared__ __attribute__((aligned(16))) double tileS[3][T::LSZ]; // the sweep's planes z-1, z, z+1 (as k_rbgs3d)
	__shared__ __attribute__((aligned(16))) double tileV[2][T::LSZ]; // red values of the recomputed plane z+2
	__shared__ double idiag[27];

	const int32_t *fk  = L.face_kind + (size_t) pid * 6;
	const int32_t *fs  = L.face_src + (size_t) pid * 6;
	const double   rhx = L.rh2[(size_t) pid * 3], rhy = L.rh2[(size_t) pid * 3 + 1], rhz = L.rh2[(size_t) pid * 3 + 2];
	const double2 *fp2 = reinterpret_cast<const double2 *>(f + (size_t) pid * NNN);
	double2       *op2 = reinterpret_cast<double2 *>(out + (size_t) pid * NNN);

	FaceKinds kinds(fk);
	idiagTable(idiag, tid, kinds, rhx, rhy, rhz);
	for (int i = tid; i < 2 * T::LSZ; i += TPB) (&tileV[0][0])[i] = 0.0; // its halo ring stays zero: ghosts of a zero iterate

	const HaloSrc  hs  = haloSrc6<N>(tid, fk, fs, L.f6, L.ghost, L.f6off);
	const PlaneSrc bot = zPlaneSrc6<N>(fk[4], fs[4], false, L.f6, L.ghost, L.f6off);
	const PlaneSrc top = zPlaneSrc6<N>(fk[5], fs[5], true, L.f6, L.ghost, L.f6off);

	const bool act = (T::NT == TPB) || tid < T::NT;
	const int  X = act ? tid % H : 0, Yp = act ? tid / H : 0;
	int        q[2], lds[2], dix[2][2];
	const int  ldo[2] = {T::row(2 * Yp) + 2 * X + 2, T::row(2 * Yp + 3) + 2 * X + 2}; // the rows below / above the pair
#pragma unroll
	for (int k = 0; k < 2; k++) {
		q[k]   = (2 * Yp + k) * H + X;
		lds[k] = T::row(2 * Yp + k + 1) + 2 * X + 2;
	}
	{
		const int cy0 = (Yp == 0) ? 0 : 1, cy1 = (Yp == H - 1) ? 2 : 1;
		const int cx0 = (X == 0) ? 0 : 1, cx1 = (X == H - 1) ? 2 : 1;
		dix[0][0] = cx0 + 3 * cy0, dix[0][1] = cx1 + 3 * cy0, dix[1][0] = cx0 + 3 * cy1, dix[1][1] = cx1 + 3 * cy1;
	}

	// coarse-correction sources (as k_rbgs3d<N, false, true, 1, CFP>)
	static_assert(!CFP || !FCORR, "the exported ghost terms are for uniformly refined levels");
	const bool    cpo  = CFP && ps.orth[pid] < 0; // this patch copies through: its correction is the same-size coarse patch
	const double *yown = cpo ? ps.coarse + (size_t) ps.parent[pid] * NNN : nullptr;
	const double *cown = cpo ? yown : coarseOctant<N>(ps, pid), *chalo = cown, *cbot = cown, *ctop = cown;
	double        shalo = 0.0, sbot = 0.0, stop = 0.0;
	const int     cq = X + N * Yp;
	bool          cpb = false, cpt = false;
	const double *ybot = nullptr, *ytop = nullptr;
	int           hsh = 1; // z shift of the halo's coarse index (0 for a copy-through neighbour)
	if (tid < 4 * N) {
		const int side = tid / N, t = tid % N;
		if (fk[side] == FACE_LOCAL) {
			if (CFP && ps.orth[fs[side]] < 0) {
				const int nbr = (side == 0) ? t * N + (N - 1) : (side == 1 ? t * N : (side == 2 ? (N - 1) * N + t : t));
				chalo = ps.coarse + (size_t) ps.parent[fs[side]] * NNN + nbr;
				hsh   = 0;
			} else {
				const double *cn = coarseOctant<N>(ps, fs[side]);
				const int     cx = (side == 0) ? H - 1 : (side == 1 ? 0 : t / 2);
				const int     cy = (side == 2) ? H - 1 : (side == 3 ? 0 : t / 2);
				chalo = cn + cx + N * cy;
			}
			shalo = 1.0;
		} else if (!CFP && fk[side] == FACE_GHOST && ps.gparent) { // a neighbour on another rank whose parent is local
			const double *cn = coarseOctantSlot<N>(ps, fs[side]);
			const int     cx = (side == 0) ? H - 1 : (side == 1 ? 0 : t / 2);
			const int     cy = (side == 2) ? H - 1 : (side == 3 ? 0 : t / 2);
			chalo = cn + cx + N * cy;
			shalo = 1.0;
		}
	}
	if (fk[4] == FACE_LOCAL) {
		if (CFP && ps.orth[fs[4]] < 0)
			cpb = true, ybot = ps.coarse + (size_t) ps.parent[fs[4]] * NNN + NN * (N - 1);
		else
			cbot = coarseOctant<N>(ps, fs[4]) + NN * (H - 1);
		sbot = 1.0;
	} else if (!CFP && fk[4] == FACE_GHOST && ps.gparent) {
		cbot = coarseOctantSlot<N>(ps, fs[4]) + NN * (H - 1);
		sbot = 1.0;
	}
	if (fk[5] == FACE_LOCAL) {
		if (CFP && ps.orth[fs[5]] < 0)
			cpt = true, ytop = ps.coarse + (size_t) ps.parent[fs[5]] * NNN;
		else
			ctop = coarseOctant<N>(ps, fs[5]);
		stop = 1.0;
	} else if (!CFP && fk[5] == FACE_GHOST && ps.gparent) {
		ctop = coarseOctantSlot<N>(ps, fs[5]);
		stop = 1.0;
	}
	// the correction of this thread's two cells of row k: plane z of the own patch / the bottom / top neighbour's facing plane
	auto ownC = [&](int k, int z) {
		if (CFP && cpo) return *reinterpret_cast<const double2 *>(yown + (size_t) z * NN + (2 * Yp + k) * N + 2 * X);
		const double c = cown[NN * (z >> 1) + cq];
		return double2{c, c};
	};
	auto botC = [&](int k) {
		if (CFP && cpb) return *reinterpret_cast<const double2 *>(ybot + (2 * Yp + k) * N + 2 * X);
		const double c = cbot[cq];
		return double2{c, c};
	};
	auto topC = [&](int k) {
		if (CFP && cpt) return *reinterpret_cast<const double2 *>(ytop + (2 * Yp + k) * N + 2 * X);
		const double c = ctop[cq];
		return double2{c, c};
	};

	const double2 zero2 = double2{0.0, 0.0};
	auto cz9of = [](int z) { return (z == 0) ? 0 : (z == N - 1 ? 18 : 9); };
	// red values of a plane from its right-hand side alone: (0 - f) / diag at the red cells, 0 elsewhere
	// (the arithmetic of relaxCell<..., ZERO_NBRS = true>)
	auto redOf = [&](auto zpar, int z, const double2(&rhs)[2], double2(&dst)[2]) {
		constexpr int ZP = decltype(zpar)::value;
		const int     c9 = cz9of(z);
		dst[0] = dst[1] = zero2;
		if ((0 + ZP) & 1)
			dst[0].y = (0.0 - rhs[0].y) * idiag[dix[0][1] + c9];
		else
			dst[0].x = (0.0 - rhs[0].x) * idiag[dix[0][0] + c9];
		if ((1 + ZP) & 1)
			dst[1].y = (0.0 - rhs[1].y) * idiag[dix[1][1] + c9];
		else
			dst[1].x = (0.0 - rhs[1].x) * idiag[dix[1][0] + c9];
	};
	auto putRed = [&](auto zpar, double *tv, const double2(&src)[2]) { // the red cells of a plane into its LDS plane
		constexpr int ZP = decltype(zpar)::value;
		if (!act) return;
		tv[lds[0] + ((0 + ZP) & 1)] = ((0 + ZP) & 1) ? src[0].y : src[0].x;
		tv[lds[1] + ((1 + ZP) & 1)] = ((1 + ZP) & 1) ? src[1].y : src[1].x;
	};
	auto fillBlack = [&](auto zpar, int z, double *tv, double2(&cen)[2], const double2(&below)[2], const double2(&above)[2],
	                     const double2(&rhs)[2]) {
		constexpr int ZP = decltype(zpar)::value;
		relaxCell<N, 0, (1 + 0 + ZP) & 1, false, LDS_ALL>(tv, idiag, cz9of(z), lds, ldo, dix, act, rhx, rhy, rhz, cen, below, above, rhs);
		relaxCell<N, 1, (1 + 1 + ZP) & 1, false, LDS_ALL>(tv, idiag, cz9of(z), lds, ldo, dix, act, rhx, rhy, rhz, cen, below, above, rhs);
	};
	using P0 = std::integral_constant<int, 0>;
	using P1 = std::integral_constant<int, 1>;

	// right-hand sides: fm, f0, f1, f2 = planes z-1 .. z+2 in registers; planes z+3 .. z+2+RING in the ring fr, plane p in slot
	// p % RING from its request (step p-2-RING) over its first use (the red values of step p-3, read from the slot itself) to
	// the top of step p-2, which takes it into f2 (takeRegs: see k_rbgs_zero_resid3d) and requests plane p+RING into the slot:
	// RING - 1 steps between a request and its first use, and no register of a plane in flight is moved.
	// Recomputed planes r1 = v(z+1) (red entries are what is read), r2 = red(z+2) -> v(z+2), r3 = red(z+3); the sweep's planes
	// as in k_rbgs3d
	constexpr int RING = (V & 32) ? (DEEP ? 3 : 4) : (DEEP ? 3 : 2);
	double2 fm[2], f0[2], f1[2], f2[2], fr[RING][2], r1[2], r2[2], r3[2];
	double2 umm[2], um[2], uc[2], un[2], un2[2];
	FCorrSrc<N> fc;
	double2     ccr[RING][FCorrSrc<N>::NX]; // FCORR: the x terms of the planes in the ring (added at a plane's first use)
	if (FCORR) fc.init(L.fcorr, pid, X, Yp);
	// state "before step 0 shifts": f0 = plane -1 (zero), f1 = plane 0, f2 = plane 1, the ring holds planes 2 .. 1+RING
#pragma unroll
	for (int k = 0; k < 2; k++) {
		fm[k] = f0[k] = zero2;
		f1[k] = fp2[0 * NP + q[k]];
		f2[k] = fp2[1 * NP + q[k]];
		umm[k] = zero2;
	}
	if (FCORR) {
		double2 c0[FCorrSrc<N>::NX], c1[FCorrSrc<N>::NX];
		fc.load(0, c0);
		fc.load(1, c1);
		fc.apply(f1, c0);
		fc.apply(f2, c1);
	}
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int a = 0; a < RING; a++) { // planes 2 .. 1+RING, oldest first (the loop's waits assume this order)
		const int p = 2 + a, zp = p < N ? p : N - 1;
#pragma unroll
		for (int k = 0; k < 2; k++) fr[p % RING][k] = ldStream<NTL>(fp2 + zp * NP + q[k]);
		if (FCORR) fc.load(zp, ccr[p % RING]);
		__builtin_amdgcn_sched_barrier(0);
	}
	if (FCORR) fc.apply(fr[2 % RING], ccr[2 % RING]); // plane 2 is read below
	__syncthreads(); // idiag, zeroed tileV
	{ // prologue: v(0), v(1)
		double2 v0[2], v1[2];
		redOf(P0{}, 0, f1, v0);
		redOf(P1{}, 1, f2, v1);
		redOf(P0{}, 2, fr[2 % RING], r2);
		putRed(P0{}, tileV[0], v0);
		putRed(P1{}, tileV[1], v1);
		ldsBarrier();
		const double2 none[2] = {zero2, zero2};
		fillBlack(P0{}, 0, tileV[0], v0, none, v1, f1);
		fillBlack(P1{}, 1, tileV[1], v1, v0, r2, f2);
#pragma unroll
		for (int k = 0; k < 2; k++) {
			if (!CFP) {
				const double  c0 = cown[cq], cb = sbot * cbot[cq]; // planes 0 and 1 share coarse plane 0
				const double2 a  = bot.p[q[k]];
				uc[k] = double2{v0[k].x + c0, v0[k].y + c0};
				un[k] = double2{v1[k].x + c0, v1[k].y + c0};
				um[k] = double2{bot.s * a.x, bot.s * a.y};
				um[k].x += bot.s * cb, um[k].y += bot.s * cb;
			} else { // (the arithmetic of k_rbgs3d<..., CFP>'s first planes)
				const double2 c0 = ownC(k, 0), c1 = ownC(k, 1), cb = botC(k);
				const double2 a  = bot.p[q[k]];
				uc[k] = double2{v0[k].x + c0.x, v0[k].y + c0.y};
				un[k] = double2{v1[k].x + c1.x, v1[k].y + c1.y};
				um[k] = double2{bot.s * a.x, bot.s * a.y};
				um[k].x += bot.s * (sbot * cb.x), um[k].y += bot.s * (sbot * cb.y);
			}
			r1[k] = v1[k];
		}
		ldsBarrier(); // both red planes have been read: step 0 rewrites the first one
	}
	// Nothing a step loads is consumed by the same step: loads return in order, so one load that is waited for in the step that
	// issues it drags every older request (the ring's) to completion with it. The halo value of plane z+1 is requested as its two
	// raw operands and combined at the top of step z+1; the coarse value of planes z+4, z+5 is requested at even z, two steps
	// before its first use (one load per coarse plane instead of one per fine plane).
	double hA = hs.p[0], hB = chalo[0];
	double cnx = CFP ? 0.0 : cown[NN * 1 + cq], ccur = 0.0; // N >= 4: planes 2 and 3 share coarse plane 1
	// CFP: the correction of the own patch's plane p, rows k = 0, 1, WITHOUT a branch in the march (a load under a condition makes
	// the compiler wait for every request in flight where the branch rejoins -- on every step) and one step ahead of its use, like
	// the halo operands: one 16-byte load per row from (base, plane stride, offset) chosen once per patch -- a patch that copies
	// through reads its two cells of the same-size coarse patch; an octant child the aligned pair that holds its coarse cell and
	// takes the half it needs (two 32-bit selects per value: not fp64 work).
#ifndef TE_CFP_AHEAD
#define TE_CFP_AHEAD 0 // 1: the pairs are requested a whole step ahead (8 more registers across the step: spills at 168)
#endif
	const double *cfb  = cpo ? yown : cown;
	const int     cfo0 = cpo ? (2 * Yp + 0) * N + 2 * X : (cq & ~1), cfd = cpo ? N : 0; // (row 1: cfo0 + cfd)
	const bool    cfy  = !cpo && (cq & 1);
	auto cfLoad = [&](int p, double2(&dst)[2]) { // (p clamped by the caller)
		const double *b = cfb + (size_t) NN * (cpo ? p : (p >> 1)) + cfo0;
		dst[0]          = *reinterpret_cast<const double2 *>(b);
		dst[1]          = *reinterpret_cast<const double2 *>(b + cfd);
	};
	auto cfPick = [&](const double2 &d) { return cpo ? d : (cfy ? double2{d.y, d.y} : double2{d.x, d.x}); };
	double2 cfn[2]; // the raw pairs of the plane the NEXT step adds its correction to
	if (CFP && TE_CFP_AHEAD) cfLoad(2, cfn);

	int bz = 0; // z % 3
	// LAST: the three steps z = N-2, N-1, N, whose plane z+2 is the top neighbour's: code of their own, so that the loads they
	// alone issue (that plane, used at once) exist nowhere in the loop -- a load inside a branch makes the compiler wait for
	// everything where the branch rejoins, on every step (it cannot know the branch was not taken).
	auto step = [&](auto zpar, auto slot_, auto last_, int z) {
		constexpr int  ZPAR = decltype(zpar)::value, SLOT = decltype(slot_)::value; // z % 2, z % RING
		constexpr bool LAST = decltype(last_)::value;                               // z + 2 >= N
		constexpr int S2 = (SLOT + 2) % RING, S3 = (SLOT + 3) % RING;               // the slots of planes z+2 and z+3
		using ZQ          = std::integral_constant<int, 1 - ZPAR>;
		double2 c2v[2]; // CFP: the correction of plane z+2 (or of the top neighbour's plane), per row
		// Order of the requests: the operands that are consumed ONE step later (halo, coarse value) first, then the ring's slot,
		// whose first use is RING - 1 steps away -- loads return in order, so waiting for a load completes everything requested
		// before it, and nothing more.
		const int zf = (z + 2 + RING < N) ? z + 2 + RING : N - 1, zc = (z + 1 < N) ? z + 1 : N - 1;
		const double hv = hs.s * (takeReg(hA) + shalo * takeReg(hB)); // plane z's
		hA              = hs.p[zc * hs.stride];
		hB              = chalo[NN * (zc >> hsh)];
		if constexpr (ZPAR == 0 && !CFP) {
			ccur = takeReg(cnx);
			cnx  = (z + 4 < N) ? cown[NN * ((z + 4) >> 1) + cq] : ctop[cq];
		}
		if constexpr (CFP && !LAST) {
			if constexpr (TE_CFP_AHEAD) { // plane z+2's correction arrived a step ago; plane z+3's is requested now
#pragma unroll
				for (int k = 0; k < 2; k++) c2v[k] = cfPick(takeRegs(cfn[k]));
				cfLoad(z + 3 < N ? z + 3 : N - 1, cfn);
			} else { // requested FIRST in this step, used behind its barrier: the wait leaves the ring's refill (below) in flight
				cfLoad(z + 2, cfn);
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		// the right-hand sides move down one plane; plane z+2 leaves the ring and its slot is requested again
#pragma unroll
		for (int k = 0; k < 2; k++) {
			fm[k] = f0[k];
			f0[k] = f1[k];
			f1[k] = f2[k];
			f2[k] = takeRegs(fr[S2][k]);
			fr[S2][k] = ldStream<NTL>(fp2 + zf * NP + q[k]);
		}
		if (FCORR) {
			fc.load(zf, ccr[S2]);
			fc.apply(fr[S3], ccr[S3]); // plane z+3: first read below
		}
		const double c2 = CFP ? 0.0 : (!LAST ? ccur : stop * ccur);
		if constexpr (CFP && LAST) { // (the top neighbour's plane: three steps of 34, loaded where it is used)
#pragma unroll
			for (int k = 0; k < 2; k++) {
				const double2 c = topC(k);
				c2v[k]          = double2{stop * c.x, stop * c.y};
			}
		}
		double2      tg[2];
		if constexpr (TG_ALWAYS || LAST) { // the top neighbour's plane (used on the last steps only)
#pragma unroll
			for (int k = 0; k < 2; k++) tg[k] = top.p[q[k]];
		}
		// before the barrier: red values two and three planes ahead, this plane of the iterate into LDS
		double *tvz = tileV[z & 1];
		if (z + 3 < N)
			redOf(ZQ{}, z + 3, fr[S3], r3);
		else
			r3[0] = r3[1] = zero2;
		if constexpr (!LAST) putRed(zpar, tvz, r2);
		double *tz = tileS[bz];
		double *tm = tileS[bz == 0 ? 2 : bz - 1];
		if (z < N) {
			if (act) {
				ldsStore2(tz + lds[0], uc[0]);
				ldsStore2(tz + lds[1], uc[1]);
			}
			if (hs.lds >= 0) tz[hs.lds] = hv;
		}
		ldsBarrier();
		// the iterate two planes ahead: black values of the recomputed plane, plus the coarse correction
		if constexpr (!LAST) {
			fillBlack(zpar, z + 2, tvz, r2, r1, r3, f2);
#pragma unroll
			for (int k = 0; k < 2; k++) {
				if constexpr (CFP && !TE_CFP_AHEAD) c2v[k] = cfPick(cfn[k]);
				un2[k] = CFP ? double2{r2[k].x + c2v[k].x, r2[k].y + c2v[k].y} : double2{r2[k].x + c2, r2[k].y + c2};
			}
		} else {
#pragma unroll
			for (int k = 0; k < 2; k++)
				un2[k] = CFP ? double2{top.s * (tg[k].x + c2v[k].x), top.s * (tg[k].y + c2v[k].y)}
				             : double2{top.s * (tg[k].x + c2), top.s * (tg[k].y + c2)};
		}
		// the sweep itself (k_rbgs3d)
		if (z < N) {
			const int cz9 = cz9of(z);
			relaxCell<N, 0, (0 + ZPAR) & 1, false>(tz, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, uc, um, un, f0);
			relaxCell<N, 1, (1 + ZPAR) & 1, false>(tz, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, uc, um, un, f0);
		}
		if (z > 0) {
			const int cz9 = cz9of(z - 1);
			relaxCell<N, 0, (1 + 0 + 1 - ZPAR) & 1, false, LDS_ALL>(tm, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, um, umm, uc, fm);
			relaxCell<N, 1, (1 + 1 + 1 - ZPAR) & 1, false, LDS_ALL>(tm, idiag, cz9, lds, ldo, dix, act, rhx, rhy, rhz, um, umm, uc, fm);
			if (act) {
				stStream<NTS>(op2 + (z - 1) * NP + q[0], um[0]);
				stStream<NTS>(op2 + (z - 1) * NP + q[1], um[1]);
				if (L.xf_out) {
					double *xo = L.xf_out + (size_t) pid * 2 * NN + N * (z - 1) + 2 * Yp;
					if (X == 0) *reinterpret_cast<double2 *>(xo) = double2{um[0].x, um[1].x};
					if (X == H - 1) *reinterpret_cast<double2 *>(xo + NN) = double2{um[0].y, um[1].y};
				}
			}
		}
#pragma unroll
		for (int k = 0; k < 2; k++) {
			umm[k] = um[k];
			um[k]  = uc[k];
			uc[k]  = un[k];
			un[k]  = un2[k];
			r1[k]  = r2[k];
			r2[k]  = r3[k];
		}
		bz = (bz == 2) ? 0 : bz + 1;
	};
	// unrolled over parity and ring slots: U = lcm(2, RING) steps per iteration over z < N-2, then the (N-2) % U steps left, then
	// the three LAST steps
	constexpr int U = (RING == 3) ? 6 : (RING == 4 ? 4 : 2);
	using NotLast = std::integral_constant<bool, false>;
	using Last    = std::integral_constant<bool, true>;
	auto run = [&](auto ic, int z) {
		constexpr int I = decltype(ic)::value;
		step(std::integral_constant<int, I & 1>{}, std::integral_constant<int, I % RING>{}, NotLast{}, z);
	};
	// every request of the prologue has arrived before the loop starts (planes 0..2 were waited for above anyway): the waits
	// inside the loop are derived from the worst state that reaches its head, and the prologue's request order is not the loop's
	__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
	int z = 0;
#pragma unroll 1
	for (; z + U <= N - 2; z += U) {
		run(std::integral_constant<int, 0>{}, z);
		run(std::integral_constant<int, 1>{}, z + 1);
		if constexpr (U >= 4) {
			run(std::integral_constant<int, 2>{}, z + 2);
			run(std::integral_constant<int, 3>{}, z + 3);
		}
		if constexpr (U == 6) {
			run(std::integral_constant<int, 4>{}, z + 4);
			run(std::integral_constant<int, 5>{}, z + 5);
		}
	}
	constexpr int REM = (N - 2) % U; // (z is a multiple of U here)
	if constexpr (REM > 0) run(std::integral_constant<int, 0>{}, z);
	if constexpr (REM > 1) run(std::integral_constant<int, 1>{}, z + 1);
	if constexpr (REM > 2) run(std::integral_constant<int, 2>{}, z + 2);
	if constexpr (REM > 3) run(std::integral_constant<int, 3>{}, z + 3);
	if constexpr (REM > 4) run(std::integral_constant<int, 4>{}, z + 4);
	step(std::integral_constant<int, 0>{}, std::integral_constant<int, (N - 2) % RING>{}, Last{}, N - 2);
	step(std::integral_constant<int, 1>{}, std::integral_constant<int, (N - 1) % RING>{}, Last{}, N - 1);
	step(std::integral_constant<int, 0>{}, std::integral_constant<int, N % RING>{}, Last{}, N);
}
} // namespace te
