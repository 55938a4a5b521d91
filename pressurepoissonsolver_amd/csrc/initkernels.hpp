// Init::initDirichlet / Init::initNeumann (apps/shared/Init.cpp:57-245 in 3D, :246-361 in 2D) for the drivers' canned
// problems, as device kernels that fill a vector in place -- no host loop, no PCIe. One thread per cell: the
// right-hand side and the exact solution at the cell centre (getXYZ, Init.cpp:25-50), then the physical-boundary data
// folded into the cells along physical faces in the reference's order (west, east, south, north, bottom, top):
// Dirichlet  f -= 2 g(face point) / h^2  (Init.cpp:186-240),  Neumann  f += n(face point) / h on a low side, -= on a
// high side (Init.cpp:89-146). Problems (apps/3d/steady.cpp:221-292, apps/2d/steady.cpp:296-318): TRIG, GAUSS; RANDOM =
// the timing inputs of BASELINE.md, f ~ U(-1,1) from splitmix64(seed + tree node id), exact = 0.
// The host-callback form of the same (arbitrary std::function problems) is thunderegg/HipInit.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace te
{
enum InitProblem : int { PROBLEM_TRIG = 0, PROBLEM_GAUSS = 1, PROBLEM_RANDOM = 2 };

struct InitGeom {
	int            dim, n, P;
	const double  *starts;  // [P][3]
	const double  *h;       // [P][3]
	const int32_t *face_kind; // [P][2 dim] (FACE_DIRICHLET / FACE_NEUMANN = physical)
	const int32_t *ids;     // [P] tree node ids
};

template <int PROB> struct Prob3 {
	static __device__ double exact(double x, double y, double z)
	{
		if (PROB == PROBLEM_TRIG) { // apps/3d/steady.cpp:258-263
			x += .3, y += .3, z += .3;
			return sin(M_PI * x) * cos(2.0 / 3 * M_PI * y) * sin(5.0 / 6 * M_PI * z);
		}
		return exp(cos(10 * M_PI * x)) - exp(cos(11 * M_PI * y)) + exp(cos(12 * M_PI * z)); // :231-233
	}
	static __device__ double rhs(double x, double y, double z)
	{
		if (PROB == PROBLEM_TRIG) { // :252-257
			x += .3, y += .3, z += .3;
			return -77.0 / 36 * M_PI * M_PI * sin(M_PI * x) * cos(2.0 / 3 * M_PI * y) * sin(5.0 / 6 * M_PI * z);
		}
		// :234-242
		return -M_PI * M_PI
		       * (100 * exp(cos(10 * M_PI * x)) * cos(10 * M_PI * x) - 100 * exp(cos(10 * M_PI * x)) * pow(sin(10 * M_PI * x), 2)
		          - 121 * exp(cos(11 * M_PI * y)) * cos(11 * M_PI * y) + 121 * exp(cos(11 * M_PI * y)) * pow(sin(11 * M_PI * y), 2)
		          + 144 * exp(cos(12 * M_PI * z)) * cos(12 * M_PI * z) - 144 * exp(cos(12 * M_PI * z)) * pow(sin(12 * M_PI * z), 2));
	}
	static __device__ double normal(int ax, double x, double y, double z)
	{
		if (PROB == PROBLEM_TRIG) { // :266-284
			x += .3, y += .3, z += .3;
			if (ax == 0) return M_PI * cos(M_PI * x) * cos(2.0 / 3 * M_PI * y) * sin(5.0 / 6 * M_PI * z);
			if (ax == 1) return -2.0 / 3 * M_PI * sin(M_PI * x) * sin(2.0 / 3 * M_PI * y) * sin(5.0 / 6 * M_PI * z);
			return 5.0 / 6 * M_PI * sin(M_PI * x) * cos(2.0 / 3 * M_PI * y) * cos(5.0 / 6 * M_PI * z);
		}
		// :243-251
		if (ax == 0) return -10 * M_PI * sin(10 * M_PI * x) * exp(cos(10 * M_PI * x));
		if (ax == 1) return 11 * M_PI * sin(11 * M_PI * y) * exp(cos(11 * M_PI * y));
		return -12 * M_PI * sin(12 * M_PI * z) * exp(cos(12 * M_PI * z));
	}
};
template <int PROB> struct Prob2 {
	static __device__ double exact(double x, double y)
	{
		if (PROB == PROBLEM_TRIG) return sin(M_PI * y) * cos(2 * M_PI * x); // apps/2d/steady.cpp:316
		return exp(cos(10 * M_PI * x)) - exp(cos(11 * M_PI * y));          // :297-299
	}
	static __device__ double rhs(double x, double y)
	{
		if (PROB == PROBLEM_TRIG) return -5 * M_PI * M_PI * sin(M_PI * y) * cos(2 * M_PI * x); // :314-315
		return 100 * M_PI * M_PI * (pow(sin(10 * M_PI * x), 2) - cos(10 * M_PI * x)) * exp(cos(10 * M_PI * x))
		       + 121 * M_PI * M_PI * (cos(11 * M_PI * y) - pow(sin(11 * M_PI * y), 2)) * exp(cos(11 * M_PI * y)); // :300-305
	}
	static __device__ double normal(int ax, double x, double y)
	{
		if (PROB == PROBLEM_TRIG) // :317-318
			return ax == 0 ? -2 * M_PI * sin(M_PI * y) * sin(2 * M_PI * x) : M_PI * cos(M_PI * y) * cos(2 * M_PI * x);
		return ax == 0 ? -10 * M_PI * sin(10 * M_PI * x) * exp(cos(10 * M_PI * x)) : 11 * M_PI * sin(11 * M_PI * y) * exp(cos(11 * M_PI * y));
	}
};

// Init.cpp:25-50: index -1 / n = the patch face, else the cell centre
__device__ __forceinline__ double initCoord(double start, double h, int n, int i)
{
	return i == -1 ? start : (i == n ? start + h * n : start + h / 2.0 + h * i);
}

template <int PROB, bool NEUMANN> __global__ __launch_bounds__(256) void k_init3d(InitGeom G, double *__restrict__ f, double *__restrict__ exact)
{
	const int    n  = G.n;
	const size_t nc = (size_t) n * n * n, total = nc * G.P;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t) gridDim.x * 256) {
		const int     p = (int) (i / nc), c = (int) (i % nc);
		const int     ci[3] = {c % n, (c / n) % n, c / (n * n)};
		const double *st = G.starts + (size_t) p * 3, *h = G.h + (size_t) p * 3;
		const double  x = initCoord(st[0], h[0], n, ci[0]), y = initCoord(st[1], h[1], n, ci[1]), z = initCoord(st[2], h[2], n, ci[2]);
		double        v = Prob3<PROB>::rhs(x, y, z);
		if (exact) exact[i] = Prob3<PROB>::exact(x, y, z);
#pragma unroll
		for (int s = 0; s < 6; s++) {
			const int ax = s >> 1, hi = s & 1;
			if (ci[ax] != (hi ? n - 1 : 0) || G.face_kind[(size_t) p * 6 + s] >= 2) continue;
			int o[3] = {ci[0], ci[1], ci[2]};
			o[ax]    = hi ? n : -1;
			const double bx = initCoord(st[0], h[0], n, o[0]), by = initCoord(st[1], h[1], n, o[1]), bz = initCoord(st[2], h[2], n, o[2]);
			if (NEUMANN) {
				const double g = Prob3<PROB>::normal(ax, bx, by, bz) / h[ax];
				v              = hi ? v - g : v + g;
			} else {
				v -= 2 * Prob3<PROB>::exact(bx, by, bz) / (h[ax] * h[ax]);
			}
		}
		f[i] = v;
	}
}
template <int PROB, bool NEUMANN> __global__ __launch_bounds__(256) void k_init2d(InitGeom G, double *__restrict__ f, double *__restrict__ exact)
{
	const int    n  = G.n;
	const size_t nc = (size_t) n * n, total = nc * G.P;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t) gridDim.x * 256) {
		const int     p = (int) (i / nc), c = (int) (i % nc);
		const int     ci[2] = {c % n, c / n};
		const double *st = G.starts + (size_t) p * 3, *h = G.h + (size_t) p * 3;
		const double  x = initCoord(st[0], h[0], n, ci[0]), y = initCoord(st[1], h[1], n, ci[1]);
		double        v = Prob2<PROB>::rhs(x, y);
		if (exact) exact[i] = Prob2<PROB>::exact(x, y);
#pragma unroll
		for (int s = 0; s < 4; s++) {
			const int ax = s >> 1, hi = s & 1;
			if (ci[ax] != (hi ? n - 1 : 0) || G.face_kind[(size_t) p * 4 + s] >= 2) continue;
			int o[2] = {ci[0], ci[1]};
			o[ax]    = hi ? n : -1;
			const double bx = initCoord(st[0], h[0], n, o[0]), by = initCoord(st[1], h[1], n, o[1]);
			if (NEUMANN) {
				const double g = Prob2<PROB>::normal(ax, bx, by) / h[ax];
				v              = hi ? v - g : v + g;
			} else {
				v -= 2 * Prob2<PROB>::exact(bx, by) / (h[ax] * h[ax]);
			}
		}
		f[i] = v;
	}
}
// f ~ U(-1, 1): element k of patch p = splitmix64 output number k+1 of the stream seeded with seed + node id (the same
// integers and the same exact conversion as problems.random_rhs: bit-identical, mesh-order independent)
static __global__ __launch_bounds__(256) void k_init_random(InitGeom G, size_t nc, uint64_t seed, double *__restrict__ f, double *__restrict__ exact)
{
	const size_t total = nc * G.P;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t) gridDim.x * 256) {
		const size_t p = i / nc, k = i % nc;
		uint64_t     x = (seed + (uint64_t) (int64_t) G.ids[p]) + 0x9E3779B97F4A7C15ull * (uint64_t) (k + 1);
		x ^= x >> 30;
		x *= 0xBF58476D1CE4E5B9ull;
		x ^= x >> 27;
		x *= 0x94D049BB133111EBull;
		x ^= x >> 31;
		f[i] = (double) (x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
		if (exact) exact[i] = 0.0;
	}
}
} // namespace te
