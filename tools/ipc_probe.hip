// Tooling: can two PROCESSES on this machine exchange halo data by direct stores into each other's device memory (hipIpc*),
// with flags in fine-grained memory and bounded spin-wait kernels -- and what does one such exchange cost next to an RCCL
// group (about 20 us per rcclGenericKernel in profiles/r04_mr8_timeline.txt)? Two processes (fork BEFORE any HIP call), both
// on device 0 (the pool's boxes have one GPU; on a node each would take its own). Per iteration each process pushes `bytes`
// into the peer's landing buffer (parity = iteration & 1), fences at system scope, raises the peer's flag to the iteration
// number; a one-workgroup kernel waits for its own flag (bounded), a check kernel compares every value with the iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/ipc_probe.hip -o tools/ipc_probe && tools/ipc_probe [bytes] [iterations]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s: %s\n", me, #x, hipGetErrorString(e_)); _exit(3); } } while (0)
static int me = 0;

__global__ void k_push(const double *src, double *dst, size_t n, unsigned long long *peer_flag, unsigned long long epoch, unsigned *done)
{
	for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) dst[i] = src[i];
	__threadfence_system();
	__syncthreads();
	if (threadIdx.x == 0) {
		if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
			__hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(peer_flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}
__global__ void k_wait(const unsigned long long *flag, unsigned long long epoch, long long budget_ticks, int *err)
{
	if (threadIdx.x != 0) return;
	const long long t0 = wall_clock64();
	while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
		if (wall_clock64() - t0 > budget_ticks) {
			*err = 1;
			return;
		}
		__builtin_amdgcn_s_sleep(8);
	}
	__threadfence_system();
}
__global__ void k_fill(double *p, size_t n, double v)
{
	for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_check(const double *p, size_t n, double v, unsigned *bad)
{
	for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
		if (p[i] != v) atomicAdd(bad, 1u);
}
static void xfer(int wfd, int rfd, const void *mine, void *theirs, size_t n)
{
	if (write(wfd, mine, n) != (ssize_t) n || read(rfd, theirs, n) != (ssize_t) n) {
		fprintf(stderr, "[%d] pipe failed\n", me);
		_exit(4);
	}
}
int main(int argc, char **argv)
{
	const size_t bytes = argc > 1 ? (size_t) atol(argv[1]) : (size_t) 1536 * 1024;
	const int    iters = argc > 2 ? atoi(argv[2]) : 2000;
	int ab[2], ba[2];
	if (pipe(ab) || pipe(ba)) return 2;
	const pid_t child = fork(); // before any HIP call
	me = child == 0 ? 1 : 0;
	const int wfd = me == 0 ? ab[1] : ba[1], rfd = me == 0 ? ba[0] : ab[0];
	CHK(hipSetDevice(0));
	const size_t n = bytes / 8;
	double *land, *src;
	unsigned long long *flags;
	unsigned *done, *bad;
	int *err;
	CHK(hipMalloc(&land, 2 * bytes));
	CHK(hipMalloc(&src, bytes));
	CHK(hipMalloc(&done, 8));
	CHK(hipMalloc(&bad, 8));
	CHK(hipMemset(done, 0, 8));
	CHK(hipMemset(bad, 0, 8));
	CHK(hipHostMalloc((void **) &err, 64, hipHostMallocMapped));
	*err = 0;
	bool fine = true;
	if (hipExtMallocWithFlags((void **) &flags, 4096, hipDeviceMallocFinegrained) != hipSuccess) {
		fine = false;
		(void) hipGetLastError();
		CHK(hipMalloc(&flags, 4096));
	}
	CHK(hipMemset(flags, 0, 4096));
	CHK(hipMemset(land, 0, 2 * bytes));
	CHK(hipDeviceSynchronize());
	hipIpcMemHandle_t hl, hf, pl, pf;
	CHK(hipIpcGetMemHandle(&hl, land));
	hipError_t ef = hipIpcGetMemHandle(&hf, flags);
	if (ef != hipSuccess && fine) { // fine-grained memory cannot be shared: a plain allocation instead
		fprintf(stderr, "[%d] hipIpcGetMemHandle(fine-grained) failed: %s; using hipMalloc for the flags\n", me, hipGetErrorString(ef));
		(void) hipGetLastError();
		fine = false;
		CHK(hipMalloc(&flags, 4096));
		CHK(hipMemset(flags, 0, 4096));
		CHK(hipDeviceSynchronize());
		CHK(hipIpcGetMemHandle(&hf, flags));
	}
	xfer(wfd, rfd, &hl, &pl, sizeof hl);
	xfer(wfd, rfd, &hf, &pf, sizeof hf);
	double *peer_land;
	unsigned long long *peer_flags;
	CHK(hipIpcOpenMemHandle((void **) &peer_land, pl, hipIpcMemLazyEnablePeerAccess));
	CHK(hipIpcOpenMemHandle((void **) &peer_flags, pf, hipIpcMemLazyEnablePeerAccess));
	int go = 1, peer_go = 0;
	xfer(wfd, rfd, &go, &peer_go, sizeof go);
	hipStream_t st;
	CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	const long long budget = 5LL * 100000000LL; // 5 s of the 100 MHz wall clock
	auto run = [&](int k0, int k1) {
		for (int k = k0; k <= k1; k++) {
			hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, src, n, (double) k + me * 0.5);
			hipLaunchKernelGGL(k_push, dim3(64), dim3(256), 0, st, src, peer_land + (k & 1) * n, n, peer_flags, (unsigned long long) k, done);
			hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st, flags, (unsigned long long) k, budget, err);
			hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, st, land + (k & 1) * n, n, (double) k + (1 - me) * 0.5, bad);
		}
	};
	run(1, 50);
	CHK(hipStreamSynchronize(st));
	const auto t0 = std::chrono::steady_clock::now();
	run(51, 50 + iters);
	CHK(hipStreamSynchronize(st));
	const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters * 1e6;
	unsigned hbad = 0;
	CHK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
	printf("[%d] flags in %s memory; %zu bytes each way, %d iterations: %.2f us per (fill + push + wait + check), mismatches %u, timeouts %d\n", me,
	       fine ? "fine-grained" : "plain device", bytes, iters, us, hbad, *err);
	// the same four launches without the peer (a local copy, no wait): what the launches alone cost
	const auto t1 = std::chrono::steady_clock::now();
	for (int k = 0; k < iters; k++) {
		hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, src, n, (double) k);
		hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, land, n, (double) k);
		hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, st, land, n, (double) k, bad);
	}
	CHK(hipStreamSynchronize(st));
	printf("[%d] three local launches of the same size: %.2f us\n", me, std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() / iters * 1e6);
	xfer(wfd, rfd, &go, &peer_go, sizeof go); // nobody unmaps while the other still pushes
	CHK(hipIpcCloseMemHandle(peer_land));
	CHK(hipIpcCloseMemHandle(peer_flags));
	fflush(stdout);
	_exit(hbad || *err ? 1 : 0);
}
