// Tooling (not product): times k_ps_fused / the three-pass kernels on synthetic data and calibrates the
// f64 MFMA issue rate. hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pressurepoissonsolver_amd/csrc tools/psf_bench.hip
#define PSF_TIMING
#include "patchsolve32_sym.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace te;

__global__ __launch_bounds__(512) void k_mfma_rate(int iters, double *out)
{
	v4f64 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int k = 0; k < 4; k++) acc[k] = mfma_f64(a, b, acc[k]);
	}
	out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

int main(int argc, char **argv)
{
	const int P = argc > 1 ? atoi(argv[1]) : 4096;
	const size_t n = (size_t) P * 32768;
	double *in, *out, *s0, *mats, *lam, *rh2, *corr;
	int32_t *plan, *zm;
	CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&s0, n * 8));
	CK(hipMalloc(&corr, (size_t) P * 6 * 1024 * 8));
	CK(hipMalloc(&mats, 6 * 1024 * 8)); CK(hipMalloc(&lam, 96 * 8)); CK(hipMalloc(&rh2, (size_t) P * 3 * 8));
	CK(hipMalloc(&plan, P * 4)); CK(hipMalloc(&zm, 4));
	std::vector<double> h(n);
	for (size_t i = 0; i < n; i++) h[i] = (double) ((i * 2654435761u) & 0xffff) / 65536.0 - 0.5;
	CK(hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice));
	CK(hipMemcpy(corr, h.data(), (size_t) P * 6 * 1024 * 8, hipMemcpyHostToDevice));
	std::vector<double> m(6 * 1024), lm(96, 1.0), rh((size_t) P * 3, 1.0);
	for (size_t i = 0; i < m.size(); i++) m[i] = ((double) ((i * 40503u) & 0xff) / 256.0 - 0.5) / 8;
	for (int i = 0; i < 96; i++) lm[i] = 1.0 + i * 0.01;
	CK(hipMemcpy(mats, m.data(), m.size() * 8, hipMemcpyHostToDevice));
	CK(hipMemcpy(lam, lm.data(), 96 * 8, hipMemcpyHostToDevice));
	CK(hipMemcpy(rh2, rh.data(), rh.size() * 8, hipMemcpyHostToDevice));
	CK(hipMemset(plan, 0, P * 4)); CK(hipMemset(zm, 0, 4));
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_fused<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PSF_LDS_BYTES));
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_fused<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PSF_LDS_BYTES));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto timeit = [&](const char *name, auto &&fn) {
		fn();
		CK(hipDeviceSynchronize());
		float best = 1e9;
		for (int r = 0; r < 5; r++) {
			CK(hipEventRecord(e0));
			fn();
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		printf("%-28s %8.3f ms\n", name, best);
		return best;
	};
	{
		const int iters = 4096;
		float ms = timeit("mfma_rate", [&] { hipLaunchKernelGGL(k_mfma_rate, dim3(256), dim3(512), 0, 0, iters, out); });
		double per_simd = (double) iters * 4 * 2; // 2 waves per SIMD
		printf("  -> %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz); %.1f TFLOP/s\n", ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4,
		       256.0 * 512 / 64 * iters * 4 * 2048 / (ms * 1e-3) / 1e12);
	}
	const dim3 gf(8 * ((P + 7) / 8)), gs(P < 256 ? P : 256);
	double *fr; CK(hipMalloc(&fr, PSS_FRAG * 8)); CK(hipMemcpy(fr, h.data(), PSS_FRAG * 8, hipMemcpyHostToDevice));
	double *inv; CK(hipMalloc(&inv, PSS_INV * 8)); CK(hipMemcpy(inv, h.data() + 4096, PSS_INV * 8, hipMemcpyHostToDevice));
	int32_t *itab; CK(hipMalloc(&itab, P * 4)); CK(hipMemset(itab, 0, P * 4));
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_sym<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PSS_LDS_BYTES));
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_sym<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PSS_LDS_BYTES));
	auto stamps = [&] {
		long long st[8][12];
		CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(psf_stamp), sizeof st));
		const char *nm[12] = {"start", "A", "B1", "Z0", "B2", "C1", "B4", "Z1", "B5", "C2", "-", "-"};
		for (int w = 0; w < 8; w += 7) {
			printf("  wave %d:", w);
			for (int k = 1; k < 10; k++) printf(" %s=%lld", nm[k], st[w][k] - st[0][0]);
			printf("\n");
		}
	};
	auto sstamps = [&] {
		long long st[8][12];
		CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(pss_stamp), sizeof st));
		const char *nm[12] = {"start", "A", "b1", "Z0", "b2", "C1", "b34", "Z1", "b5", "C2", "-", "-"};
		for (int w = 0; w < 8; w += 7) {
			printf("  wave %d:", w);
			for (int k = 1; k < 10; k++) printf(" %s=%lld", nm[k], st[w][k] - st[0][0]);
			printf("\n");
		}
	};
	for (int skew = 0; skew <= 0; skew++) {
		printf("skew %d\n", skew);
		timeit("sym zero-guess", [&] { hipLaunchKernelGGL(k_ps_sym<false>, gs, dim3(512), PSS_LDS_BYTES, 0, P, plan, fr, inv, itab, in, (const double *) nullptr, out, s0, (const int32_t *) nullptr); });
		sstamps();
		timeit("sym with corr", [&] { hipLaunchKernelGGL(k_ps_sym<true>, gs, dim3(512), PSS_LDS_BYTES, 0, P, plan, fr, inv, itab, in, corr, out, s0, (const int32_t *) nullptr); });
		sstamps();
	}
	timeit("fused zero-guess", [&] { hipLaunchKernelGGL(k_ps_fused<false>, gf, dim3(512), PSF_LDS_BYTES, 0, P, plan, mats, lam, zm, rh2, in, (const double *) nullptr, out, (const int32_t *) nullptr); });
	stamps();
	timeit("fused with corr", [&] { hipLaunchKernelGGL(k_ps_fused<true>, gf, dim3(512), PSF_LDS_BYTES, 0, P, plan, mats, lam, zm, rh2, in, corr, out, (const int32_t *) nullptr); });
	stamps();
	timeit("xy fwd zero", [&] { hipLaunchKernelGGL(k_ps_xy<false>, dim3(P), dim3(256), 0, 0, P, plan, mats, in, (const double *) nullptr, out); });
	timeit("xy fwd corr", [&] { hipLaunchKernelGGL((k_ps_xy<false, true>), dim3(P), dim3(256), 0, 0, P, plan, mats, in, corr, out); });
	timeit("z", [&] { hipLaunchKernelGGL(k_ps_z, dim3(P), dim3(256), 0, 0, P, plan, mats, lam, zm, rh2, out, s0); });
	timeit("xy inv", [&] { hipLaunchKernelGGL(k_ps_xy<true>, dim3(P), dim3(256), 0, 0, P, plan, mats, s0, (const double *) nullptr, out); });
	return 0;
}
