// Device stream-bandwidth probes (tooling, not product): what a 2-read + 1-write fp64 stream can
// reach on this MI355X with the access shapes our kernels use. hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NT> __device__ __forceinline__ void st2(double2 *p, double2 v)
{
	if (NT) {
		__builtin_nontemporal_store(v.x, &p->x);
		__builtin_nontemporal_store(v.y, &p->y);
	} else
		*p = v;
}
template <int NT> __device__ __forceinline__ double2 ld2(const double2 *p)
{
	if (NT) {
		double2 v;
		v.x = __builtin_nontemporal_load(&p->x);
		v.y = __builtin_nontemporal_load(&p->y);
		return v;
	}
	return *p;
}
// grid-stride, 16 B per lane
template <int NT> __global__ __launch_bounds__(256) void fill16(size_t n2, double2 *a)
{
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t) gridDim.x * 256) st2<NT>(a + i, double2{1.0, 2.0});
}
template <int NT> __global__ __launch_bounds__(256) void read16(size_t n2, const double2 *a, double *out)
{
	double s = 0;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t) gridDim.x * 256) {
		double2 v = ld2<NT>(a + i);
		s += v.x + v.y;
	}
	if (s == 123.456) out[0] = s;
}
template <int NT> __global__ __launch_bounds__(256) void triad16(size_t n2, double2 *a, const double2 *b, const double2 *c)
{
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t) gridDim.x * 256) {
		double2 x = ld2<NT>(b + i), y = ld2<NT>(c + i);
		st2<NT>(a + i, double2{x.x + 0.5 * y.x, x.y + 0.5 * y.y});
	}
}
__global__ __launch_bounds__(256) void triad8(size_t n, double *a, const double *b, const double *c)
{
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) a[i] = b[i] + 0.5 * c[i];
}
// patch-shaped: one workgroup owns a contiguous 32^3 block and walks it plane by plane (8 KiB steps),
// W = bytes per lane (8 or 16), DEPTH planes in flight
template <int W, int NT> __global__ __launch_bounds__(256) void triad_patch(int P, double *a, const double *b, const double *c, int stagger = 0, int remap = 1)
{
	const int chunk = (P + 7) >> 3;
	const int pid   = remap ? (blockIdx.x & 7) * chunk + (blockIdx.x >> 3) : blockIdx.x;
	if (pid >= P) return;
	const int zoff = (pid * stagger) & 31;
	const size_t base = (size_t) pid * 32768;
	if (W == 8) {
		for (int zz = 0; zz < 32; zz++) {
			const int z = (zz + zoff) & 31;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				size_t i = base + z * 1024 + k * 256 + threadIdx.x;
				a[i]     = b[i] + 0.5 * c[i];
			}
		}
	} else {
		for (int zz = 0; zz < 32; zz++) {
			const int z = (zz + zoff) & 31;
#pragma unroll
			for (int k = 0; k < 2; k++) {
				size_t  i = (base + z * 1024) / 2 + k * 256 + threadIdx.x;
				double2 x = ld2<NT>((const double2 *) b + i), y = ld2<NT>((const double2 *) c + i);
				st2<NT>((double2 *) a + i, double2{x.x + 0.5 * y.x, x.y + 0.5 * y.y});
			}
		}
	}
}

// chunked walk: each workgroup streams CH planes (CH*8 KiB per array) of a contiguous array, chunks in
// address order; CH = 32 is the patch walk, CH = 1 is one plane per workgroup
template <int CH> __global__ __launch_bounds__(256) void triad_chunk(size_t nchunks, double *a, const double *b, const double *c)
{
	const size_t base = (size_t) blockIdx.x * CH * 512; // double2 units
	for (int z = 0; z < CH; z++) {
#pragma unroll
		for (int k = 0; k < 2; k++) {
			size_t  i = base + z * 512 + k * 256 + threadIdx.x;
			double2 x = ((const double2 *) b)[i], y = ((const double2 *) c)[i];
			((double2 *) a)[i] = double2{x.x + 0.5 * y.x, x.y + 0.5 * y.y};
		}
	}
}
// same, but all loads of a chunk's plane are issued one plane ahead of the stores (software pipeline)
template <int CH> __global__ __launch_bounds__(256) void triad_chunk_pipe(size_t nchunks, double *a, const double *b, const double *c)
{
	const size_t base = (size_t) blockIdx.x * CH * 512;
	double2      x[2], y[2], xn[2], yn[2];
#pragma unroll
	for (int k = 0; k < 2; k++) {
		size_t i = base + k * 256 + threadIdx.x;
		x[k]     = ((const double2 *) b)[i];
		y[k]     = ((const double2 *) c)[i];
	}
	for (int z = 0; z < CH; z++) {
		const int zn = (z + 1 < CH) ? z + 1 : z;
#pragma unroll
		for (int k = 0; k < 2; k++) {
			size_t i = base + zn * 512 + k * 256 + threadIdx.x;
			xn[k]    = ((const double2 *) b)[i];
			yn[k]    = ((const double2 *) c)[i];
		}
#pragma unroll
		for (int k = 0; k < 2; k++) {
			size_t i = base + z * 512 + k * 256 + threadIdx.x;
			((double2 *) a)[i] = double2{x[k].x + 0.5 * y[k].x, x[k].y + 0.5 * y[k].y};
			x[k] = xn[k];
			y[k] = yn[k];
		}
	}
}

// patch walk with PL planes per step: loads of PL planes issued together one step ahead, stores of PL planes together
template <int CH, int PL> __global__ __launch_bounds__(256) void triad_walk_multi(size_t nchunks, double *a, const double *b, const double *c)
{
	const size_t base = (size_t) blockIdx.x * CH * 512;
	double2      x[2 * PL], y[2 * PL], xn[2 * PL], yn[2 * PL];
#pragma unroll
	for (int k = 0; k < 2 * PL; k++) {
		size_t i = base + k * 256 + threadIdx.x;
		x[k]     = ((const double2 *) b)[i];
		y[k]     = ((const double2 *) c)[i];
	}
	for (int z = 0; z < CH; z += PL) {
		const int zn = (z + PL < CH) ? z + PL : z;
#pragma unroll
		for (int k = 0; k < 2 * PL; k++) {
			size_t i = base + zn * 512 + k * 256 + threadIdx.x;
			xn[k]    = ((const double2 *) b)[i];
			yn[k]    = ((const double2 *) c)[i];
		}
#pragma unroll
		for (int k = 0; k < 2 * PL; k++) {
			size_t i = base + z * 512 + k * 256 + threadIdx.x;
			((double2 *) a)[i] = double2{x[k].x + 0.5 * y[k].x, x[k].y + 0.5 * y[k].y};
			x[k] = xn[k];
			y[k] = yn[k];
		}
	}
}
// read-one-write-one walk (the sweep kernels' mix), PL planes per step
template <int CH, int PL> __global__ __launch_bounds__(256) void copy_walk_multi(size_t nchunks, double *a, const double *b)
{
	const size_t base = (size_t) blockIdx.x * CH * 512;
	double2      x[2 * PL], xn[2 * PL];
#pragma unroll
	for (int k = 0; k < 2 * PL; k++) x[k] = ((const double2 *) b)[base + k * 256 + threadIdx.x];
	for (int z = 0; z < CH; z += PL) {
		const int zn = (z + PL < CH) ? z + PL : z;
#pragma unroll
		for (int k = 0; k < 2 * PL; k++) xn[k] = ((const double2 *) b)[base + zn * 512 + k * 256 + threadIdx.x];
#pragma unroll
		for (int k = 0; k < 2 * PL; k++) {
			((double2 *) a)[base + z * 512 + k * 256 + threadIdx.x] = double2{x[k].x * 0.5, x[k].y * 0.5};
			x[k] = xn[k];
		}
	}
}
// long-lived workgroups, but at every step the grid covers ONE contiguous window: workgroup i's step z is plane z*G + i
__global__ __launch_bounds__(256) void copy_walk_interleaved(size_t nplanes, double *a, const double *b)
{
	const size_t G = gridDim.x;
	double2      x[2], xn[2];
	size_t       pl = blockIdx.x;
	if (pl >= nplanes) return;
#pragma unroll
	for (int k = 0; k < 2; k++) x[k] = ((const double2 *) b)[pl * 512 + k * 256 + threadIdx.x];
	for (; pl < nplanes; pl += G) {
		const size_t pn = (pl + G < nplanes) ? pl + G : pl;
#pragma unroll
		for (int k = 0; k < 2; k++) xn[k] = ((const double2 *) b)[pn * 512 + k * 256 + threadIdx.x];
#pragma unroll
		for (int k = 0; k < 2; k++) {
			((double2 *) a)[pl * 512 + k * 256 + threadIdx.x] = double2{x[k].x * 0.5, x[k].y * 0.5};
			x[k] = xn[k];
		}
	}
}
// WAVE-SPECIALISED walk: waves 0-3 load (AH planes ahead) and hand every plane to LDS; wave 4 alone stores it to global
// memory. Loads and stores then sit in different waves' vmcnt queues: a wave that waits for a load never waits for a
// store issued after it (one in-order counter per wave covers both on gfx9). READS = 1 (copy) or 2 (triad).
template <int CH, int AH, int READS> __global__ __launch_bounds__(320) void walk_ws(size_t nchunks, double *a, const double *b, const double *c)
{
	extern __shared__ __attribute__((aligned(16))) double2 wsbuf[]; // [2][512] (+ padding that limits residency)
	const size_t base = (size_t) blockIdx.x * CH * 512;
	const int    t    = threadIdx.x;
	if (t < 256) {
		double2 x[AH][2], y[AH][2];
#pragma unroll
		for (int h = 0; h < AH; h++)
#pragma unroll
			for (int k = 0; k < 2; k++) {
				x[h][k] = ((const double2 *) b)[base + h * 512 + k * 256 + t];
				if (READS == 2) y[h][k] = ((const double2 *) c)[base + h * 512 + k * 256 + t];
			}
		for (int z0 = 0; z0 < CH; z0 += AH) {
#pragma unroll
			for (int h = 0; h < AH; h++) {
				const int z = z0 + h;
				if (z >= CH) break;
				double2   v[2];
#pragma unroll
				for (int k = 0; k < 2; k++) {
					v[k] = READS == 2 ? double2{x[h][k].x + 0.5 * y[h][k].x, x[h][k].y + 0.5 * y[h][k].y} : double2{x[h][k].x * 0.5, x[h][k].y * 0.5};
				}
				const int zn = (z + AH < CH) ? z + AH : z;
#pragma unroll
				for (int k = 0; k < 2; k++) {
					x[h][k] = ((const double2 *) b)[base + zn * 512 + k * 256 + t];
					if (READS == 2) y[h][k] = ((const double2 *) c)[base + zn * 512 + k * 256 + t];
				}
				wsbuf[(z & 1) * 512 + t]       = v[0];
				wsbuf[(z & 1) * 512 + 256 + t] = v[1];
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_s_barrier();
			}
		}
	} else {
		const int l = t - 256;
		for (int z = 0; z < CH; z++) {
			__builtin_amdgcn_s_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			double2 v[8];
#pragma unroll
			for (int k = 0; k < 8; k++) v[k] = wsbuf[(z & 1) * 512 + k * 64 + l];
#pragma unroll
			for (int k = 0; k < 8; k++) ((double2 *) a)[base + z * 512 + k * 64 + l] = v[k];
		}
	}
}
// the same walk with the same LDS hop but NO specialisation: every wave loads, passes its values through LDS and stores
template <int CH, int AH, int READS> __global__ __launch_bounds__(256) void walk_lds_nows(size_t nchunks, double *a, const double *b, const double *c)
{
	extern __shared__ __attribute__((aligned(16))) double2 wsbuf[];
	const size_t base = (size_t) blockIdx.x * CH * 512;
	const int    t    = threadIdx.x;
	double2      x[AH][2], y[AH][2];
#pragma unroll
	for (int h = 0; h < AH; h++)
#pragma unroll
		for (int k = 0; k < 2; k++) {
			x[h][k] = ((const double2 *) b)[base + h * 512 + k * 256 + t];
			if (READS == 2) y[h][k] = ((const double2 *) c)[base + h * 512 + k * 256 + t];
		}
	for (int z0 = 0; z0 < CH; z0 += AH) {
#pragma unroll
		for (int h = 0; h < AH; h++) {
			const int z = z0 + h;
			if (z >= CH) break;
			double2   v[2];
#pragma unroll
			for (int k = 0; k < 2; k++)
				v[k] = READS == 2 ? double2{x[h][k].x + 0.5 * y[h][k].x, x[h][k].y + 0.5 * y[h][k].y} : double2{x[h][k].x * 0.5, x[h][k].y * 0.5};
			const int zn = (z + AH < CH) ? z + AH : z;
#pragma unroll
			for (int k = 0; k < 2; k++) {
				x[h][k] = ((const double2 *) b)[base + zn * 512 + k * 256 + t];
				if (READS == 2) y[h][k] = ((const double2 *) c)[base + zn * 512 + k * 256 + t];
			}
			((double2 *) a)[base + z * 512 + t]       = v[0];
			((double2 *) a)[base + z * 512 + 256 + t] = v[1];
		}
	}
}
__global__ __launch_bounds__(256) void copy_flat(size_t n2, double2 *a, const double2 *b)
{
	size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (i < n2) { double2 x = b[i]; a[i] = double2{x.x * 0.5, x.y * 0.5}; }
}

int main(int argc, char **argv)
{
	size_t n = (size_t) 512 * 512 * 512;
	if (argc > 1) n = (size_t) atol(argv[1]);
	double *a, *b, *c;
	CK(hipMalloc(&a, n * 8));
	CK(hipMalloc(&b, n * 8));
	CK(hipMalloc(&c, n * 8));
	CK(hipMemset(a, 0, n * 8));
	CK(hipMemset(b, 0, n * 8));
	CK(hipMemset(c, 0, n * 8));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	auto timeit = [&](const char *name, double bytes, auto launch) {
		for (int i = 0; i < 3; i++) launch();
		CK(hipDeviceSynchronize());
		float best = 1e30f, tot = 0;
		const int reps = 10;
		for (int i = 0; i < reps; i++) {
			CK(hipEventRecord(e0));
			launch();
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			best = ms < best ? ms : best;
			tot += ms;
		}
		printf("%-44s best %8.1f us  %7.1f GB/s   avg %7.1f GB/s\n", name, best * 1e3, bytes / best / 1e6, bytes / (tot / reps) / 1e6);
	};
	const size_t n2 = n / 2;
	const int    P  = (int) (n / 32768);
	char         nm[128];
	timeit("hipMemsetAsync", n * 8.0, [&] { (void) hipMemsetAsync(a, 0, n * 8, 0); });
	for (int grid : {1024, 2048, 4096, 16384, (int) (n2 / 256)}) {
		snprintf(nm, sizeof nm, "fill16 grid=%d", grid);
		timeit(nm, n * 8.0, [&] { hipLaunchKernelGGL(fill16<0>, dim3(grid), dim3(256), 0, 0, n2, (double2 *) a); });
		snprintf(nm, sizeof nm, "fill16 nt grid=%d", grid);
		timeit(nm, n * 8.0, [&] { hipLaunchKernelGGL(fill16<1>, dim3(grid), dim3(256), 0, 0, n2, (double2 *) a); });
	}
	for (int grid : {2048, 4096, (int) (n2 / 256)}) {
		snprintf(nm, sizeof nm, "read16 grid=%d", grid);
		timeit(nm, n * 8.0, [&] { hipLaunchKernelGGL(read16<0>, dim3(grid), dim3(256), 0, 0, n2, (const double2 *) b, a); });
		snprintf(nm, sizeof nm, "triad16 grid=%d", grid);
		timeit(nm, n * 24.0, [&] { hipLaunchKernelGGL(triad16<0>, dim3(grid), dim3(256), 0, 0, n2, (double2 *) a, (const double2 *) b, (const double2 *) c); });
		snprintf(nm, sizeof nm, "triad16 nt grid=%d", grid);
		timeit(nm, n * 24.0, [&] { hipLaunchKernelGGL(triad16<1>, dim3(grid), dim3(256), 0, 0, n2, (double2 *) a, (const double2 *) b, (const double2 *) c); });
		snprintf(nm, sizeof nm, "triad8 grid=%d", grid);
		timeit(nm, n * 24.0, [&] { hipLaunchKernelGGL(triad8, dim3(grid), dim3(256), 0, 0, n, a, b, c); });
	}
	const int pg = 8 * ((P + 7) / 8);
	timeit("triad_patch W=8", n * 24.0, [&] { hipLaunchKernelGGL((triad_patch<8, 0>), dim3(pg), dim3(256), 0, 0, P, a, b, c); });
	timeit("triad_patch W=16", n * 24.0, [&] { hipLaunchKernelGGL((triad_patch<16, 0>), dim3(pg), dim3(256), 0, 0, P, a, b, c); });
	timeit("triad_patch W=16 nt", n * 24.0, [&] { hipLaunchKernelGGL((triad_patch<16, 1>), dim3(pg), dim3(256), 0, 0, P, a, b, c); });
	for (int st : {1, 5, 7, 13})
		for (int rm : {0, 1}) {
			snprintf(nm, sizeof nm, "triad_patch W=16 stagger=%d remap=%d", st, rm);
			timeit(nm, n * 24.0, [&] { hipLaunchKernelGGL((triad_patch<16, 0>), dim3(pg), dim3(256), 0, 0, P, a, b, c, st, rm); });
		}
	timeit("triad_patch W=16 stagger=0 remap=0", n * 24.0, [&] { hipLaunchKernelGGL((triad_patch<16, 0>), dim3(pg), dim3(256), 0, 0, P, a, b, c, 0, 0); });
	timeit("triad_patch W=8 stagger=5 remap=1", n * 24.0, [&] { hipLaunchKernelGGL((triad_patch<8, 0>), dim3(pg), dim3(256), 0, 0, P, a, b, c, 5, 1); });
	timeit("triad_patch W=16 nt stagger=5 remap=1", n * 24.0, [&] { hipLaunchKernelGGL((triad_patch<16, 1>), dim3(pg), dim3(256), 0, 0, P, a, b, c, 5, 1); });
#define CHUNK(CH)                                                                                                        \
	snprintf(nm, sizeof nm, "triad_chunk CH=%d", CH);                                                                    \
	timeit(nm, n * 24.0, [&] { hipLaunchKernelGGL(triad_chunk<CH>, dim3(n / (CH * 1024)), dim3(256), 0, 0, n / (CH * 1024), a, b, c); }); \
	snprintf(nm, sizeof nm, "triad_chunk_pipe CH=%d", CH);                                                               \
	timeit(nm, n * 24.0, [&] { hipLaunchKernelGGL(triad_chunk_pipe<CH>, dim3(n / (CH * 1024)), dim3(256), 0, 0, n / (CH * 1024), a, b, c); });
	CHUNK(1) CHUNK(2) CHUNK(4) CHUNK(8) CHUNK(16) CHUNK(32)
#define WALK(PL)                                                                                                          \
	snprintf(nm, sizeof nm, "triad_walk_multi CH=32 PL=%d", PL);                                                          \
	timeit(nm, n * 24.0, [&] { hipLaunchKernelGGL((triad_walk_multi<32, PL>), dim3(n / (32 * 1024)), dim3(256), 0, 0, n / (32 * 1024), a, b, c); }); \
	snprintf(nm, sizeof nm, "copy_walk_multi CH=32 PL=%d", PL);                                                           \
	timeit(nm, n * 16.0, [&] { hipLaunchKernelGGL((copy_walk_multi<32, PL>), dim3(n / (32 * 1024)), dim3(256), 0, 0, n / (32 * 1024), a, b); });
	WALK(1) WALK(2) WALK(4)
	for (int G : {1024, 2048, 4096}) {
		snprintf(nm, sizeof nm, "copy_walk_interleaved G=%d", G);
		timeit(nm, n * 16.0, [&] { hipLaunchKernelGGL(copy_walk_interleaved, dim3(G), dim3(256), 0, 0, n / 1024, a, b); });
	}
	// wave-specialised walks: LDS bytes per workgroup set how many are resident per CU (48 KiB: three, as the fused kernels)
	for (int lds : {16384, 32768, 49152, 65536})
		for (int rd : {1, 2}) {
			const double bytes = n * (rd == 2 ? 24.0 : 16.0);
			snprintf(nm, sizeof nm, "walk_ws AH=3 reads=%d lds=%d", rd, lds);
			if (rd == 1) timeit(nm, bytes, [&] { hipLaunchKernelGGL((walk_ws<32, 3, 1>), dim3(n / (32 * 1024)), dim3(320), lds, 0, n / (32 * 1024), a, b, c); });
			else timeit(nm, bytes, [&] { hipLaunchKernelGGL((walk_ws<32, 3, 2>), dim3(n / (32 * 1024)), dim3(320), lds, 0, n / (32 * 1024), a, b, c); });
			snprintf(nm, sizeof nm, "walk_nows AH=3 reads=%d lds=%d", rd, lds);
			if (rd == 1) timeit(nm, bytes, [&] { hipLaunchKernelGGL((walk_lds_nows<32, 3, 1>), dim3(n / (32 * 1024)), dim3(256), lds, 0, n / (32 * 1024), a, b, c); });
			else timeit(nm, bytes, [&] { hipLaunchKernelGGL((walk_lds_nows<32, 3, 2>), dim3(n / (32 * 1024)), dim3(256), lds, 0, n / (32 * 1024), a, b, c); });
		}
	for (int lds : {32768, 49152}) {
		snprintf(nm, sizeof nm, "walk_ws AH=1 reads=1 lds=%d", lds);
		timeit(nm, n * 16.0, [&] { hipLaunchKernelGGL((walk_ws<32, 1, 1>), dim3(n / (32 * 1024)), dim3(320), lds, 0, n / (32 * 1024), a, b, c); });
		snprintf(nm, sizeof nm, "walk_ws AH=4 reads=2 lds=%d", lds);
		timeit(nm, n * 24.0, [&] { hipLaunchKernelGGL((walk_ws<32, 4, 2>), dim3(n / (32 * 1024)), dim3(320), lds, 0, n / (32 * 1024), a, b, c); });
	}
	timeit("copy_flat", n * 16.0, [&] { hipLaunchKernelGGL(copy_flat, dim3(n2 / 256), dim3(256), 0, 0, n2, (double2 *) a, (const double2 *) b); });
	return 0;
}
