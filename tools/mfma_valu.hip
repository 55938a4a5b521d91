// Tooling (not product): do fp64 vector instructions and v_mfma_f64_16x16x4_f64 share execution resources on gfx950?
// The dense fp64 matrix peak of MI355X equals its fp64 vector peak; if the matrix instruction runs on the vector unit's DP lanes,
// every v_fma_f64 / v_add_f64 of k_ps_sym (butterflies, eigenvalue divisions) is time taken from its MFMAs, whichever wave issues it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pressurepoissonsolver_amd/csrc tools/mfma_valu.hip -o tools/mfma_valu
// One 512-thread workgroup per CU (two waves per SIMD, as k_ps_sym). Variants:
//   mfma      both waves of a SIMD issue M MFMAs each (4 independent accumulators)
//   valu      both waves issue V dependent-free v_fma_f64 each (8 independent chains)
//   split     wave w < 4 (one per SIMD) issues MFMAs only, its partner w + 4 fp64 FMAs only: separate units -> max(t), shared -> sum
//   mixed k   every wave: k fp64 FMAs behind each MFMA (same wave interleaving, what the kernel's phases look like)
//   rcp       v_rcp_f64 rate; div: the kernel's pssDiv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f64 mfma_f64(double a, double b, v4f64 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <int MODE, int K> __global__ __launch_bounds__(512) void k_mix(int iters, double *out)
{
	const int wave = threadIdx.x >> 6;
	v4f64     acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	double    a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
	double    c[8];
#pragma unroll
	for (int k = 0; k < 8; k++) c[k] = 0.5 + 0.01 * k + threadIdx.x * 1e-5;
	const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
	const bool do_valu = MODE == 1 || (MODE == 2 && wave >= 4);
	if (MODE == 3) {
		for (int i = 0; i < iters; i++) {
#pragma unroll
			for (int k = 0; k < 4; k++) {
				acc[k] = mfma_f64(a, b, acc[k]);
#pragma unroll
				for (int v = 0; v < K; v++) c[(k * K + v) & 7] = __builtin_fma(c[(k * K + v) & 7], 0.999, 1e-3);
			}
		}
	} else if (MODE == 4) { // v_rcp_f64
		for (int i = 0; i < iters; i++) {
#pragma unroll
			for (int k = 0; k < 8; k++) c[k] = __builtin_amdgcn_rcp(c[k]) + 0.5;
		}
	} else if (MODE == 5) { // pssDiv
		for (int i = 0; i < iters; i++) {
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const double n = c[k], d = b + k;
				double       x = __builtin_amdgcn_rcp(d);
				double       e = __builtin_fma(-d, x, 1.0);
				x              = __builtin_fma(x, e, x);
				e              = __builtin_fma(-d, x, 1.0);
				x              = __builtin_fma(x, e, x);
				const double q = n * x;
				c[k]           = __builtin_fma(__builtin_fma(-d, q, n), x, q) + 0.25;
			}
		}
	} else {
		if (do_mfma)
			for (int i = 0; i < iters; i++) {
#pragma unroll
				for (int k = 0; k < 4; k++) acc[k] = mfma_f64(a, b, acc[k]);
			}
		if (do_valu)
			for (int i = 0; i < iters * K; i++) {
#pragma unroll
				for (int k = 0; k < 8; k++) c[k] = __builtin_fma(c[k], 0.999, 1e-3);
			}
	}
	double s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
#pragma unroll
	for (int k = 0; k < 8; k++) s += c[k];
	out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main()
{
	double *out;
	CK(hipMalloc(&out, 256 * 512 * 8));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	auto timeit = [&](auto &&fn) {
		fn();
		CK(hipDeviceSynchronize());
		float best = 1e9;
		for (int r = 0; r < 5; r++) {
			CK(hipEventRecord(e0));
			fn();
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		return best;
	};
	const int iters = 4096;
	const float t_m = timeit([&] { hipLaunchKernelGGL((k_mix<0, 1>), dim3(256), dim3(512), 0, 0, iters, out); });
	printf("mfma   : %.3f ms  = %.1f ns per MFMA per SIMD (2 waves x %d)\n", t_m, t_m * 1e6 / (2.0 * iters * 4), iters * 4);
	const float t_v = timeit([&] { hipLaunchKernelGGL((k_mix<1, 4>), dim3(256), dim3(512), 0, 0, iters, out); });
	printf("valu   : %.3f ms  = %.2f ns per v_fma_f64 per SIMD (2 waves x %d)\n", t_v, t_v * 1e6 / (2.0 * iters * 4 * 8), iters * 32);
	// split: 1 wave per SIMD does iters*4 MFMAs, the other iters*K*8 FMAs
	const float t_m1 = timeit([&] { hipLaunchKernelGGL((k_mix<2, 0>), dim3(256), dim3(512), 0, 0, iters, out); });
	printf("split K=0 (one wave per SIMD: MFMAs only, partner idle): %.3f ms\n", t_m1);
	const float t_s4 = timeit([&] { hipLaunchKernelGGL((k_mix<2, 4>), dim3(256), dim3(512), 0, 0, iters, out); });
	const float t_s8 = timeit([&] { hipLaunchKernelGGL((k_mix<2, 8>), dim3(256), dim3(512), 0, 0, iters, out); });
	const float t_s16 = timeit([&] { hipLaunchKernelGGL((k_mix<2, 16>), dim3(256), dim3(512), 0, 0, iters, out); });
	printf("split K=4 (partner: %d FMAs): %.3f ms;  K=8: %.3f ms;  K=16: %.3f ms   (FMAs alone on one wave: %.3f / %.3f / %.3f ms at the rate above)\n",
	       iters * 32, t_s4, t_s8, t_s16, t_v / 2, t_v, 2 * t_v);
	const float t_x1 = timeit([&] { hipLaunchKernelGGL((k_mix<3, 1>), dim3(256), dim3(512), 0, 0, iters, out); });
	const float t_x2 = timeit([&] { hipLaunchKernelGGL((k_mix<3, 2>), dim3(256), dim3(512), 0, 0, iters, out); });
	const float t_x4 = timeit([&] { hipLaunchKernelGGL((k_mix<3, 4>), dim3(256), dim3(512), 0, 0, iters, out); });
	const float t_x8 = timeit([&] { hipLaunchKernelGGL((k_mix<3, 8>), dim3(256), dim3(512), 0, 0, iters, out); });
	printf("mixed (k FMAs behind every MFMA, both waves): k=1 %.3f  k=2 %.3f  k=4 %.3f  k=8 %.3f ms   (mfma alone %.3f)\n", t_x1, t_x2, t_x4, t_x8, t_m);
	const float t_r = timeit([&] { hipLaunchKernelGGL((k_mix<4, 1>), dim3(256), dim3(512), 0, 0, iters, out); });
	printf("rcp    : %.3f ms  = %.2f ns per v_rcp_f64 (+ add) per SIMD\n", t_r, t_r * 1e6 / (2.0 * iters * 8));
	const float t_d = timeit([&] { hipLaunchKernelGGL((k_mix<5, 1>), dim3(256), dim3(512), 0, 0, iters, out); });
	printf("pssDiv : %.3f ms  = %.2f ns per division per SIMD\n", t_d, t_d * 1e6 / (2.0 * iters * 8));
	return 0;
}
