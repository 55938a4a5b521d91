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
//
// The no-credit argument as CHECKS (PushErr): every exchange of a slot (level, kind) is symmetric -- each rank raises a flag at
// every peer of the plan and waits for every peer's flag, whatever the counts -- so when a rank pushes epoch e it has passed its
// own wait for e - 1, which needed every peer's flag >= e - 1: the peer's push e - 1 sits, in the peer's stream order, behind every
// reader of the buffer (e & 1) that push e is about to overwrite. (i) The wait for epoch e therefore never sees a flag beyond e + 1
// (the peer cannot pass ITS wait for e + 1 before this rank's push e + 1, which comes after this wait): a larger value is
// PUSH_ERR_OVERRUN. (ii) The kernel that raises epoch e first reads the flag the same peer raises HERE: less than e - 1 is
// PUSH_ERR_PEER_BEHIND (the stores of this push may have overwritten data the peer had not read). (iii) It keeps the last epoch
// it raised per peer and slot in its own memory: anything but e - 1 is PUSH_ERR_SEQUENCE (the host skipped or repeated an exchange).
// All three set the solver's error word like a wait that gave up; te_gmg_push_failed returns the code.
// ASSUMPTION the three checks rest on: the exchanges of ONE slot are ordered on the device -- push e + 1 of a slot is issued behind
// the wait for e of the same slot. pushExchange runs on whichever stream the level's overlap form hands it (the communication
// stream under the interior patches, or the solver stream), and the host orders the two with events around every exchange
// (withGhosts: ev_pack in front, ev_recv behind), so the order holds whatever form a level takes and when the form changes between
// cycles; tests/test_gpu_multirank.py::test_direct_store_transport_under_every_overlap_form runs every form and every change of form.
// A caller that issued exchanges of one slot on two streams WITHOUT such events would trip PUSH_ERR_SEQUENCE (by design).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace te
{
enum PushErr : int { PUSH_OK = 0, PUSH_ERR_TIMEOUT = 1, PUSH_ERR_OVERRUN = 2, PUSH_ERR_PEER_BEHIND = 3, PUSH_ERR_SEQUENCE = 4 };
// one flag this rank raises: `flag` in the peer's table; `back` = the flag the same peer raises in MY table for the same slot (its
// progress as I can see it); `sent` = my own record of the last epoch I raised there (device memory of this rank)
struct PushFlag {
	unsigned long long       *flag;
	const unsigned long long *back;
	unsigned long long       *sent;
};
__device__ __forceinline__ void pushFail(int *err, int *err_host, int code)
{
	int expected = 0; // (the first failure names the cause)
	if (__hip_atomic_compare_exchange_strong(err, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
		__hip_atomic_store(err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// one thread, after the system-scope fence behind the last workgroup's data: checks (ii) and (iii) above, then the flag.
// `raise` is `epoch` except under fault injection (TE_PUSH_FAULT).
__device__ __forceinline__ void pushRaise(const PushFlag &f, unsigned long long epoch, unsigned long long raise, int *err, int *err_host)
{
	if (!f.flag) return;
	if (f.back && __hip_atomic_load(f.back, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) + 1 < epoch) pushFail(err, err_host, PUSH_ERR_PEER_BEHIND);
	if (f.sent) {
		if (*f.sent + 1 != epoch) pushFail(err, err_host, PUSH_ERR_SEQUENCE);
		*f.sent = epoch;
	}
	__hip_atomic_store(f.flag, raise, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct PushPeer {
	double  *dst;          // where this peer's range starts in ITS receive buffer (of the current parity)
	int64_t  src_off, cnt; // the range in my send buffer, doubles (cnt may be 0: the flag is raised all the same)
	PushFlag fl;
};
constexpr int PUSH_MAX_PEERS = 64;
struct PushPlan {
	PushPeer p[PUSH_MAX_PEERS];
	int      n;
};
// grid (blocks per peer, peers); even counts and 16-byte aligned ranges (face layers and blocks are multiples of 16 doubles)
static __global__ __launch_bounds__(256) void k_push_ranges(const double *__restrict__ src, PushPlan plan, unsigned long long epoch,
                                                     unsigned long long raise, unsigned *done, int *err, int *err_host)
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
			for (int k = 0; k < plan.n; k++) pushRaise(plan.p[k].fl, epoch, raise, err, err_host);
		}
	}
}
struct PushWait {
	const unsigned long long *flag[PUSH_MAX_PEERS]; // my flags the peers of this exchange raise
	int                       n;
};
// err: this solver's error word in device memory (what the kernels test); err_host: the same in pinned host memory, written
// only on a failure (what the host's watchdog reads without touching the device)
static __global__ __launch_bounds__(64) void k_push_wait(PushWait w, unsigned long long epoch, long long budget, int *err, int *err_host)
{
	const int k = threadIdx.x;
	if (k < w.n && !*err) {
		const long long    t0 = wall_clock64();
		unsigned long long v;
		while ((v = __hip_atomic_load(w.flag[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) < epoch) {
			if (wall_clock64() - t0 > budget || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
				pushFail(err, err_host, PUSH_ERR_TIMEOUT);
				break;
			}
			__builtin_amdgcn_s_sleep(4);
		}
		if (v > epoch + 1) pushFail(err, err_host, PUSH_ERR_OVERRUN); // check (i): the peer is two exchanges ahead of what it can know
	}
	// (no fence here: the loads above are acquires, and the kernels that read the received data start behind this one on the same
	// stream -- a kernel boundary is a system-scope release / acquire; a fence by all 64 lanes cost 2 us per exchange)
}
} // namespace te
