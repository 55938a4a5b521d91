// Direct-store halo exchange (te_gmg_use_push): a rank writes the face layers (or restricted blocks) a peer needs straight
// into that peer's receive buffer -- device memory of another process / another GPU of the node, mapped through hipIpc* -- and
// raises a flag there; the peer's one-workgroup wait kernel holds its solver stream until every expected flag has arrived.
// Two small launches per exchange instead of an RCCL group (one rcclGenericKernel of about 20 us plus its gaps,
// profiles/r04_mr8_timeline.txt). Replaces, like the RCCL path, the PETSc VecScatter of SchurHelper.h:123-150 and
// GMG/InterLevelComm.h:169-189.
//
// Memory model, spelled out because it cannot be tested on one GPU: the data are plain stores into the peer's (coarse-grained)
// buffer; every thread then executes a system-scope fence; the LAST workgroup of the push kernel (agent-scope counter) stores
// the epoch into the peer's flag with system-scope release. Flags live in FINE-GRAINED device memory (hipDeviceMallocFinegrained:
// coherent across agents without a kernel boundary). The waiting kernel polls its own flag with system-scope acquire loads
// and ends; the kernels that read the received data start after it on the same stream (kernel boundary = acquire).
// Receive buffers are double buffered by the parity of the exchange count, which is what lets a rank run at most one exchange
// ahead of a peer without a credit message (gmg_transport.hip pushExchange). A wait is bounded: after `budget` ticks of the 100 MHz wall
// clock it sets *err and returns, and every later push / wait of the solver returns at once -- the host turns that into an error
// (or, inside te_gmg_autotune, into "this transport is not usable here").
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace te
{
struct PushPeer {
	double             *dst;  // where this peer's range starts in ITS receive buffer (of the current parity)
	int64_t             src_off, cnt; // the range in my send buffer, doubles
	unsigned long long *flag; // the flag I raise in the peer's flag table (null: nothing to tell it)
};
constexpr int PUSH_MAX_PEERS = 64;
struct PushPlan {
	PushPeer p[PUSH_MAX_PEERS];
	int      n;
};
// grid (blocks per peer, peers); even counts and 16-byte aligned ranges (face layers and blocks are multiples of 16 doubles)
static __global__ __launch_bounds__(256) void k_push_ranges(const double *__restrict__ src, PushPlan plan, unsigned long long epoch, unsigned *done,
                                                     const int *__restrict__ err)
{
	if (*err) return;
	const PushPeer  pp = plan.p[blockIdx.y];
	const double2  *s2 = reinterpret_cast<const double2 *>(src + pp.src_off);
	double2        *d2 = reinterpret_cast<double2 *>(pp.dst);
	const size_t    n2 = (size_t) pp.cnt / 2;
	for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n2; i += (size_t) gridDim.x * blockDim.x) d2[i] = s2[i];
	// one release fence per workgroup, behind the barrier that orders every thread's stores before it (a fence per thread costs
	// 2-4 x as much: MI355X_MICROARCH.md "inter-workgroup visibility")
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence_system();
		const unsigned total = gridDim.x * gridDim.y;
		if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == total - 1) { // every workgroup's data is out
			__hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__threadfence_system();
			for (int k = 0; k < plan.n; k++)
				if (plan.p[k].flag) __hip_atomic_store(plan.p[k].flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}
struct PushWait {
	const unsigned long long *flag[PUSH_MAX_PEERS]; // my flags the peers of this exchange raise
	int                       n;
};
// err: this solver's error word in device memory (what the kernels test); err_host: the same in pinned host memory, written
// only when a wait gives up (what the host's watchdog reads without touching the device)
static __global__ __launch_bounds__(64) void k_push_wait(PushWait w, unsigned long long epoch, long long budget, int *err, int *err_host)
{
	const int k = threadIdx.x;
	if (k < w.n && !*err) {
		const long long t0 = wall_clock64();
		while (__hip_atomic_load(w.flag[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
			if (wall_clock64() - t0 > budget || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
				__hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				break;
			}
			__builtin_amdgcn_s_sleep(4);
		}
	}
	// (no fence here: the loads above are acquires, and the kernels that read the received data start behind this one on the same
	// stream -- a kernel boundary is a system-scope release / acquire; a fence by all 64 lanes cost 2 us per exchange)
}
} // namespace te
