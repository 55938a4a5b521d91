// Tooling: what a hand-rolled barrier among a few co-resident workgroups costs on this chip (the coarse tail of a V-cycle is a
// chain of launches of a few workgroups; one launch with barriers in it only pays if a barrier is much cheaper than a kernel
// boundary). G workgroups of 256 threads run K rounds of: write one line each, barrier (agent-scope release / acquire on a
// counter in global memory), read the line of the neighbour workgroup. Compared with K dependent empty launches of G workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/gridbar.hip -o tools/gridbar && tools/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void gbar(unsigned *cnt, unsigned *gen, unsigned nwg)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
			__hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
		}
	}
	__syncthreads();
}
// stride: only workgroups with blockIdx % stride == 0 take part (stride 8: all on one XCD under round-robin dispatch)
__global__ void k_rounds(int K, int stride, unsigned *cnt, unsigned *gen, double *buf, double *out, int payload)
{
	if (blockIdx.x % stride) return;
	const int w = blockIdx.x / stride, nw = gridDim.x / stride;
	double    acc = 0.0;
	for (int k = 0; k < K; k++) {
		for (int i = threadIdx.x; i < payload; i += blockDim.x) buf[(size_t) w * payload + i] = k + w + i;
		__threadfence();
		gbar(cnt, gen, nw);
		const int o = (w + 1) % nw;
		for (int i = threadIdx.x; i < payload; i += blockDim.x) acc += __hip_atomic_load(&buf[(size_t) o * payload + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		gbar(cnt, gen, nw);
	}
	if (acc == -1.0) out[0] = acc;
	if (threadIdx.x == 0 && w == 0) out[1] = acc;
}
__global__ void k_empty(double *buf, int payload)
{
	for (int i = threadIdx.x; i < payload; i += blockDim.x) buf[(size_t) blockIdx.x * payload + i] += 1.0;
}
int main()
{
	unsigned *ctr;
	double   *buf, *out;
	const int payload = 1024; // 8 KiB per workgroup and round
	CHK(hipMalloc(&ctr, 256));
	CHK(hipMemset(ctr, 0, 256));
	CHK(hipMalloc(&buf, sizeof(double) * payload * 4096));
	CHK(hipMalloc(&out, 64));
	hipEvent_t a, b;
	CHK(hipEventCreate(&a));
	CHK(hipEventCreate(&b));
	const int K = 200;
	for (int stride : {1, 8})
		for (int G : {8, 64, 256, 512}) {
			const int grid = G * stride;
			if (grid > 512) continue; // co-residency: 256 CUs x >= 2 workgroups of 256 threads
			for (int rep = 0; rep < 2; rep++) {
				CHK(hipMemset(ctr, 0, 256));
				CHK(hipEventRecord(a));
				hipLaunchKernelGGL(k_rounds, dim3(grid), dim3(256), 0, 0, K, stride, ctr, ctr + 32, buf, out, payload);
				CHK(hipEventRecord(b));
				CHK(hipEventSynchronize(b));
				float ms;
				CHK(hipEventElapsedTime(&ms, a, b));
				if (rep) printf("barrier rounds: %3d workgroups (grid %3d, stride %d): %.2f us per round of 2 barriers\n", G, grid, stride, ms * 1e3 / K);
			}
		}
	for (int G : {8, 64, 256, 512})
		for (int rep = 0; rep < 2; rep++) {
			CHK(hipEventRecord(a));
			for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_empty, dim3(G), dim3(256), 0, 0, buf, payload);
			CHK(hipEventRecord(b));
			CHK(hipEventSynchronize(b));
			float ms;
			CHK(hipEventElapsedTime(&ms, a, b));
			if (rep) printf("dependent launches: %3d workgroups: %.2f us per launch\n", G, ms * 1e3 / K);
		}
	return 0;
}
