// Exchanges between ranks: RCCL binding, host callback, direct-store transport, scalar reductions, watchdog (see gmg_internal.hpp).
#include "gmg_internal.hpp"

namespace tei
{
// ------------------------------------------------------------------------------ launches
// Watchdog (multi-rank only): a peer that never posts its half of an exchange leaves RCCL (or the host callback)
// waiting for ever, with no error. Every exchange arms a deadline and records an event behind itself on its
// stream; a thread polls the event and ends the PROCESS (exit status 86, message on stderr) when the deadline
// passes first -- the launcher (torchrun, mpirun) then takes the job down instead of hanging the node.
void watchdogRetire(te_gmg::Watchdog &w) // (mutex held) drop completed exchanges from the front
{
	while (w.head < w.tail) {
		auto &sl = w.slot[w.head % te_gmg::Watchdog::RING];
		if (!sl.recorded || hipEventQuery(sl.ev) != hipSuccess) break;
		w.head++;
	}
}

// (mutex held through `lk`) The ring is full -- the host is RING exchanges ahead of the GPU: wait for the oldest outstanding one
// (without the mutex: the polling thread needs it) until a slot is free. Nothing that is being watched is ever given up.
void watchdogMakeRoom(te_gmg::Watchdog &w, std::unique_lock<std::mutex> &lk)
{
	watchdogRetire(w);
	while (w.tail - w.head == te_gmg::Watchdog::RING) {
		auto      &sl = w.slot[w.head % te_gmg::Watchdog::RING];
		hipEvent_t ev = sl.ev;
		const bool rec = sl.recorded;
		lk.unlock();
		if (rec)
			(void) hipEventSynchronize(ev);
		else
			std::this_thread::sleep_for(std::chrono::milliseconds(1)); // (still inside its host call on another thread)
		lk.lock();
		watchdogRetire(w);
	}
}

void watchdogLoop(te_gmg *g)
{
	auto &w = g->wd;
	while (!w.stop.load()) {
		std::this_thread::sleep_for(std::chrono::milliseconds(50));
		// (TE_PUSH_NONFATAL: the caller polls te_gmg_push_failed itself -- bench.py's trial of the transport behind its headline)
		if (g->push.err_host && *g->push.err_host && g->push.fatal.load() && g->cfg.num(O_PUSH_NONFATAL, 0) == 0) { // (TE_PUSH_NONFATAL=0 is "fatal", like the switch unset)
			fprintf(stderr,
			        "te_hip watchdog: rank %d: a direct-store exchange failed (code %d: 1 gave up waiting for a peer's data, 2 a peer's flag two "
			        "exchanges ahead, 3 a peer behind when its buffer was overwritten, 4 epochs out of sequence) -- a peer is missing or issued "
			        "a different exchange sequence; ending the process\n",
			        g->rank, *g->push.err_host);
			fflush(stderr);
			_exit(86);
		}
		std::lock_guard<std::mutex> lk(w.mu);
		watchdogRetire(w);
		if (w.head == w.tail) continue;
		const auto  &sl     = w.slot[w.head % te_gmg::Watchdog::RING];
		const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - sl.since).count();
		if (waited > w.timeout_s) {
			fprintf(stderr,
			        "te_hip watchdog: rank %d: exchange (tag %d, level %d) has not completed after %.0f s -- a peer is "
			        "missing or issued a different exchange sequence; ending the process\n",
			        g->rank, sl.tag, sl.level, waited);
			fflush(stderr);
			_exit(86);
		}
	}
}

void watchdogStart(te_gmg *g)
{
	auto &w = g->wd;
	if (w.th.joinable() || (g->nranks < 2 && !g->cfg.has(O_EXCHANGE_TIMEOUT))) return; // (one rank: only when asked for, te_gmg_watchdog_selftest)
	w.timeout_s = g->cfg.real(O_EXCHANGE_TIMEOUT, w.timeout_s);
	if (w.timeout_s <= 0) return; // TE_EXCHANGE_TIMEOUT=0 disables it
	for (auto &sl : w.slot)
		if (hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) return;
	w.th = std::thread(watchdogLoop, g);
}

void watchdogStop(te_gmg *g)
{
	auto &w = g->wd;
	if (w.th.joinable()) {
		w.stop.store(true);
		w.th.join();
	}
	for (auto &sl : w.slot) {
		if (sl.ev) (void) hipEventDestroy(sl.ev);
		sl.ev = nullptr;
	}
}

int doExchange(te_gmg *g, int tag, const ExPlan &pl, const double *send, double *recv, hipStream_t stream)
{
	if (!stream) stream = g->stream;
	const bool timed = (stream == g->stream);
	if (pl.empty()) return TE_OK;
	if (g->recording) { // te_gmg_verify_schedule: who would talk to whom, in which order; nothing moves
		for (size_t i = 0; i < pl.peers.size(); i++)
			g->record.push_back({tag, g->cur_level, pl.peers[i], pl.send_cnt[i], pl.recv_cnt[i]});
		return TE_OK;
	}
	WatchdogArm arm(g, stream, tag);
	if (g->rccl.comm) {
		// one RCCL group per exchange, enqueued on the solver stream behind the pack kernel: every
		// send/recv of the exchange progresses together over the direct xGMI links, no host round trip
		int64_t total = 0; // (loop-back only)
		if (g->cfg.has(O_RCCL_LOOPBACK)) {
			for (size_t i = 0; i < pl.peers.size(); i++) total += std::max(pl.send_cnt[i], pl.recv_cnt[i]);
			if ((size_t) (2 * total) > g->loopbuf.n) { // (grows to the largest plan of the solver within the first cycle; outside the timed scope)
				HIPCHK(hipStreamSynchronize(g->stream));
				HIPCHK(hipStreamSynchronize(g->comm_stream)); // (overlapped exchanges use it too)
				if (g->loopbuf.p) HIPCHK(hipFree(g->loopbuf.p));
				g->loopbuf.p = nullptr;
				g->loopbuf.n = 0;
				int rc0      = g->loopbuf.alloc((size_t) (2 * total));
				if (rc0) {
					g->loopbuf.p = nullptr;
					g->loopbuf.n = 0;
					return rc0;
				}
			}
		}
		std::unique_ptr<Timed> t(timed ? new Timed(g, KC_EXCHANGE, 0) : nullptr);
		if (g->cfg.has(O_RCCL_LOOPBACK)) {
			// DIAGNOSTIC (tools/mr8_budget.py): one rank of an N-rank hierarchy alone on a GPU, every peer replaced by the rank
			// itself -- the same group of ncclRecv/ncclSend calls with the same message sizes, between scratch buffers. What
			// is measured is real (host enqueue cost, RCCL's launch, this rank's kernels with the GPU to themselves); the
			// exchanged DATA are not: results are meaningless in this mode.
			int     rc  = g->rccl.GroupStart();
			int64_t off = 0;
			for (size_t i = 0; i < pl.peers.size() && rc == 0; i++) {
				const size_t c = (size_t) std::max(pl.send_cnt[i], pl.recv_cnt[i]);
				if (c == 0) continue;
				rc = g->rccl.Recv(g->loopbuf.p + total + off, c, ncclFloat64, 0, g->rccl.comm, stream);
				if (rc == 0) rc = g->rccl.Send(g->loopbuf.p + off, c, ncclFloat64, 0, g->rccl.comm, stream);
				off += (int64_t) c;
			}
			int rc2 = g->rccl.GroupEnd();
			if (rc || rc2) return te::fail(TE_ESTATE, std::string("RCCL loopback exchange failed: ") + g->rccl.GetErrorString(rc ? rc : rc2));
			return TE_OK;
		}
		int rc = g->rccl.GroupStart();
		for (size_t i = 0; i < pl.peers.size() && rc == 0; i++) {
			if (pl.recv_cnt[i] > 0)
				rc = g->rccl.Recv(recv + pl.recv_off[i], (size_t) pl.recv_cnt[i], ncclFloat64, pl.peers[i], g->rccl.comm, stream);
			if (rc == 0 && pl.send_cnt[i] > 0)
				rc = g->rccl.Send(send + pl.send_off[i], (size_t) pl.send_cnt[i], ncclFloat64, pl.peers[i], g->rccl.comm, stream);
		}
		int rc2 = g->rccl.GroupEnd();
		if (rc || rc2) return te::fail(TE_ESTATE, std::string("RCCL exchange failed: ") + g->rccl.GetErrorString(rc ? rc : rc2));
		return TE_OK;
	}
	if (!g->exchange)
		return te::fail(TE_ESTATE, "this level has off-rank neighbours: call te_gmg_set_exchange or te_gmg_use_rccl first");
	std::unique_ptr<Timed> t(timed ? new Timed(g, KC_EXCHANGE, 0) : nullptr);
	int   rc = g->exchange(g->exchange_user, tag, send, recv, (int) pl.peers.size(), pl.peers.data(), pl.send_off.data(),
	                       pl.send_cnt.data(), pl.recv_off.data(), pl.recv_cnt.data(), (void *) stream);
	if (rc) return te::fail(TE_ESTATE, "exchange callback failed with status " + std::to_string(rc));
	return TE_OK;
}

// Sum (op 0) or maximum (op 1) over the ranks of n (<= 4) doubles that the solver stream has left in g->result;
// returns them in g->result_host. One rank: a copy. Replaces the MPI_Allreduce of Vector.h:294,306,319.
int finishReduce(te_gmg *g, int n, int op, bool global)
{
	if (global && g->nranks > 1 && g->rccl.comm) {
		WatchdogArm arm(g, g->stream, 100 + op);
		int rc = g->rccl.AllReduce(g->result.p, g->result.p, (size_t) n, ncclFloat64, op ? ncclMax : ncclSum, g->rccl.comm, g->stream);
		if (rc) return te::fail(TE_ESTATE, std::string("ncclAllReduce failed: ") + g->rccl.GetErrorString(rc));
	}
	HIPCHK(hipMemcpyAsync(g->result_host, g->result.p, n * sizeof(double), hipMemcpyDeviceToHost, g->stream));
	HIPCHK(hipStreamSynchronize(g->stream));
	if (global && g->nranks > 1 && !g->rccl.comm) {
		if (!g->allreduce)
			return te::fail(TE_ESTATE, "a reduction over ranks needs te_gmg_use_rccl or te_gmg_set_allreduce");
		WatchdogArm arm(g, g->stream, 100 + op);
		int rc = g->allreduce(g->allreduce_user, g->result_host, n, op);
		if (rc) return te::fail(TE_ESTATE, "allreduce callback failed with status " + std::to_string(rc));
	}
	return TE_OK;
}

// One exchange through the direct-store transport (pushkernels.hpp): kind 1 = the level's face exchange (send: the layers in
// send order; lands in the peers' ghost slots of this exchange's parity), kind 2 = the in-place exchange of restricted blocks
// (send = the coarse level's right-hand side of this gather's parity; every rank's run lands at the same offsets there).
// Two launches on `stream`: push, wait. The epochs only ever grow, so a flag that is already ahead (a fast peer) passes.
// pushBegin: the exchange's parity and epoch (L.push_par / push_ep) and what it will wait for (L.push_wait); pushFinish: the wait
// kernel, and the switch of the ghost buffer the kernels read. Between the two: k_push_ranges, or a pack kernel that stores
// into the peers' buffers itself (PackPush).
void pushBegin(te_gmg *g, LevelHost &L, int kind)
{
	const ExPlan &pl = kind == 1 ? L.fx : L.tx_direct;
	uint64_t     &ep = kind == 1 ? L.face_epoch : L.blk_epoch;
	L.push_par       = (int) (ep & 1); // this exchange's buffer
	L.push_ep        = ++ep;
	const int slot   = 2 * L.index + (kind - 1);
	L.push_wait.n    = 0;
	// every peer of the plan, whatever the counts: the exchange is symmetric (pushkernels.hpp, the no-credit argument)
	for (size_t i = 0; i < pl.peers.size(); i++) L.push_wait.flag[L.push_wait.n++] = g->push.flags + (size_t) pl.peers[i] * g->push.nslot + slot;
}

int pushFinish(te_gmg *g, LevelHost &L, int kind, hipStream_t stream)
{
	const double    secs   = g->push.trial ? std::min(g->push.timeout_s, g->push.trial_timeout_s) : g->push.timeout_s;
	const long long budget = (long long) (secs * 1e8); // wall_clock64: 100 MHz
	if (L.push_wait.n > 0)
		hipLaunchKernelGGL(k_push_wait, dim3(1), dim3(64), 0, stream, L.push_wait, L.push_ep, budget, g->push.err, g->push.err_host);
	if (kind == 1) L.ghost_par = L.push_par; // what the kernels behind this exchange read
	HIPCHK(hipGetLastError());
	return TE_OK;
}

// the flag this rank raises at peer r for `slot`, with the two words its checks need (pushkernels.hpp PushFlag)
static PushFlag pushFlagFor(te_gmg *g, int r, int slot)
{
	PushFlag f;
	f.flag = g->push.peer_flags[r] + (size_t) g->rank * g->push.nslot + slot;
	f.back = g->push.flags + (size_t) r * g->push.nslot + slot;
	f.sent = g->push.sent + (size_t) r * g->push.nslot + slot;
	return f;
}

int pushExchange(te_gmg *g, LevelHost &L, int kind, const double *send, hipStream_t stream)
{
	if (!stream) stream = g->stream;
	const ExPlan &pl = kind == 1 ? L.fx : L.tx_direct;
	if (pl.empty()) return TE_OK;
	if ((int) pl.peers.size() > PUSH_MAX_PEERS) return te::fail(TE_EUNSUPPORTED, "direct-store exchange: too many peers");
	std::unique_ptr<Timed> t(stream == g->stream ? new Timed(g, KC_EXCHANGE, 0) : nullptr);
	WatchdogArm            arm(g, stream, kind);
	pushBegin(g, L, kind);
	const int par  = L.push_par;
	const int slot = 2 * L.index + (kind - 1);
	PushPlan  pp;
	int64_t   most = 0;
	pp.n           = 0;
	for (size_t i = 0; i < pl.peers.size(); i++) { // (a peer that gets no data still gets its flag)
		PushPeer &q = pp.p[pp.n++];
		q.dst       = (kind == 1 ? L.push_peer_ghost[par][i] : L.push_peer_cf[par][i] + pl.send_off[i]);
		q.src_off   = pl.send_off[i];
		q.cnt       = pl.send_cnt[i];
		q.fl        = pushFlagFor(g, pl.peers[i], slot);
		most        = std::max(most, q.cnt);
	}
	const int bx = (int) std::min<int64_t>(64, std::max<int64_t>(1, most / 2 / 256 / 4));
	hipLaunchKernelGGL(k_push_ranges, dim3(bx, pp.n), dim3(256), 0, stream, send, pp, L.push_ep, pushRaiseValue(g, L.push_ep), L.push_done.p + (kind - 1),
	                   g->push.err, g->push.err_host);
	return pushFinish(g, L, kind, stream);
}

// the level's face exchange: remote slots of the current ghost buffer <- the peers' layers (`send` in send order)
int faceExchange(te_gmg *g, LevelHost &L, const double *send, hipStream_t stream)
{
	if (g->push.on && L.push_faces && !g->recording) return pushExchange(g, L, 1, send, stream);
	return doExchange(g, 1, L.fx, send, L.ghostCur(), stream);
}

} // namespace tei

extern "C" {
int   te_gmg_set_exchange(te_gmg *g, te_exchange_fn fn, void *user)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_set_exchange: null");
		if (fn && g->rccl.comm) { // an explicit callback replaces the native RCCL back-end
			(void) g->rccl.CommDestroy(g->rccl.comm);
			g->rccl.comm = nullptr;
		}
		g->exchange      = fn;
		g->exchange_user = user;
		if (fn) watchdogStart(g);
		return TE_OK;
	});
}

int te_gmg_set_allreduce(te_gmg *g, te_allreduce_fn fn, void *user)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_set_allreduce: null");
		g->allreduce      = fn;
		g->allreduce_user = user;
		return TE_OK;
	});
}

static void *rcclSym(void *lib, const char *name) { return dlsym(lib, name); }

int te_rccl_unique_id(const char *libpath, char *id128)
{
	return guarded([&]() -> int {
		if (!libpath || !id128) return te::fail(TE_EINVAL, "te_rccl_unique_id: null argument");
		void *lib = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
		if (!lib) return te::fail(TE_EIO, std::string("te_rccl_unique_id: dlopen failed: ") + dlerror());
		auto get = (int (*)(void *)) rcclSym(lib, "ncclGetUniqueId");
		if (!get) return te::fail(TE_EIO, "te_rccl_unique_id: ncclGetUniqueId not found");
		int rc = get(id128);
		if (rc) return te::fail(TE_ESTATE, "ncclGetUniqueId failed");
		return TE_OK;
	});
}

int te_gmg_use_rccl(te_gmg *g, const char *libpath, const char *id128, int rank, int nranks)
{
	return guarded([&]() -> int {
		if (!g || !libpath || !id128) return te::fail(TE_EINVAL, "te_gmg_use_rccl: null argument");
		HIPCHK(hipSetDevice(g->device));
		void *lib = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
		if (!lib) return te::fail(TE_EIO, std::string("te_gmg_use_rccl: dlopen failed: ") + dlerror());
		static_assert(sizeof(ncclUniqueId) == 128, "te_rccl_unique_id hands out 128 bytes");
		ncclUniqueId id;
		memcpy(&id, id128, 128);
		auto init = (int (*)(void **, int, ncclUniqueId, int)) rcclSym(lib, "ncclCommInitRank");
		te_gmg::Rccl r;
		r.lib            = lib;
		r.GroupStart     = (int (*)()) rcclSym(lib, "ncclGroupStart");
		r.GroupEnd       = (int (*)()) rcclSym(lib, "ncclGroupEnd");
		r.Send           = (int (*)(const void *, size_t, int, int, void *, hipStream_t)) rcclSym(lib, "ncclSend");
		r.Recv           = (int (*)(void *, size_t, int, int, void *, hipStream_t)) rcclSym(lib, "ncclRecv");
		r.CommDestroy    = (int (*)(void *)) rcclSym(lib, "ncclCommDestroy");
		r.AllReduce      = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)) rcclSym(lib, "ncclAllReduce");
		r.GetErrorString = (const char *(*) (int) ) rcclSym(lib, "ncclGetErrorString");
		r.CommCount      = (int (*)(void *, int *)) rcclSym(lib, "ncclCommCount");
		r.CommUserRank   = (int (*)(void *, int *)) rcclSym(lib, "ncclCommUserRank");
		if (!init || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.CommDestroy || !r.GetErrorString || !r.AllReduce)
			return te::fail(TE_EIO, "te_gmg_use_rccl: RCCL symbols missing in " + std::string(libpath));
		if (nranks > 1 && (rank != g->rank || nranks != g->nranks)) // (before the communicator exists: nothing to leak)
			return te::fail(TE_EINVAL, "te_gmg_use_rccl: rank / nranks differ from the hierarchy's");
		int rc = init(&r.comm, nranks, id, rank);
		if (rc) return te::fail(TE_ESTATE, std::string("ncclCommInitRank failed: ") + r.GetErrorString(rc));
		if (g->rccl.comm && g->rccl.CommDestroy) (void) g->rccl.CommDestroy(g->rccl.comm); // a second call replaces the first communicator
		g->rccl = r;
		watchdogStart(g);
		return TE_OK;
	});
}

// one 64-bit word per process, drawn once: tells ranks that live in THIS process (the tests' virtual ranks: their device pointers are
// valid here as they are) from ranks of other processes that happen to publish the same pid (pids repeat across containers, PID
// namespaces and nodes) -- those are reached through hipIpcOpenMemHandle or not at all
static uint64_t processNonce()
{
	static const uint64_t nonce = [] {
		uint64_t v = 0;
		if (FILE *f = fopen("/dev/urandom", "rb")) {
			if (fread(&v, sizeof v, 1, f) != 1) v = 0;
			fclose(f);
		}
		v ^= (uint64_t) std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull;
		v ^= (uint64_t) getpid() << 32;
		return v ? v : 1;
	}();
	return nonce;
}
static uint32_t hostHash() // (same node? a mapping across nodes cannot work: said by name instead of by a failing hipIpcOpenMemHandle)
{
	char buf[256] = {0};
	if (FILE *f = fopen("/proc/sys/kernel/random/boot_id", "r")) {
		if (!fgets(buf, sizeof buf, f)) buf[0] = 0;
		fclose(f);
	}
	if (!buf[0]) (void) gethostname(buf, sizeof buf - 1);
	uint32_t h = 2166136261u;
	for (const char *c = buf; *c; c++) h = (h ^ (uint8_t) *c) * 16777619u;
	return h ? h : 1;
}

} // extern "C" (resumed below)
void tei::pushTeardown(te_gmg *g, bool final)
{
	auto &P = g->push;
	for (void *m : P.opened) (void) hipIpcCloseMemHandle(m);
	P.opened.clear();
	for (size_t l = 0; l + 1 < g->levels.size(); l++) { // (the coarse vectors own their first buffer, the level its second)
		LevelHost &L = *g->levels[l];
		if (L.cf_buf[0]) g->levels[l + 1]->f->d = L.cf_buf[0];
		L.cf_buf[0] = L.cf_buf[1] = nullptr;
	}
	for (auto &Lp : g->levels) {
		Lp->push_faces = Lp->push_blocks = false;
		Lp->ghost_par                    = 0;
	}
	if (P.flags) (void) hipFree(P.flags);
	if (P.sent) (void) hipFree(P.sent);
	// (the error words outlive a failed set-up: the watchdog thread reads err_host without a lock; they go with the solver)
	if (final) {
		if (P.err) (void) hipFree(P.err);
		if (P.err_host) (void) hipHostFree(P.err_host);
		P.err = P.err_host = nullptr;
	}
	P.flags = P.sent = nullptr;
	P.peer_flags.clear();
	P.ready = P.on = false;
}
extern "C" {

// The direct-store transport (pushkernels.hpp). Collective over the ranks of the hierarchy; needs a working reduction over the
// ranks (te_gmg_use_rccl or te_gmg_set_allreduce) to publish, once, every rank's IPC handles and receive offsets: a directory of
// 32-bit words, one slice per rank, summed over the ranks eight words at a time (each word has one contributor).
// Ranks that live in this very process (the tests' virtual ranks: same pid AND same process nonce) are reached through their raw
// pointers. Whatever fails on one rank -- an allocation, hipIpcGetMemHandle, a mapping -- that rank still takes part in every
// reduction of the set-up and the failure travels with them: all ranks return an error together, nobody waits for a peer that
// has left, and what was allocated or mapped so far is released (pushTeardown), so that a later attempt starts clean.
static int pushSetup(te_gmg *g)
{
	auto &P = g->push;
	if (P.ready) return TE_OK;
	const int R = g->nranks, NL = (int) g->levels.size();
	if (R < 2) return te::fail(TE_ESTATE, "te_gmg_use_push: one rank has nobody to push to");
	if (!g->rccl.comm && !g->allreduce) return te::fail(TE_ESTATE, "te_gmg_use_push: needs te_gmg_use_rccl or te_gmg_set_allreduce first (the handles travel through it)");
	if (R > PUSH_MAX_PEERS) return te::fail(TE_EUNSUPPORTED, "te_gmg_use_push: too many ranks");
	const bool self = g->cfg.has(O_RCCL_LOOPBACK); // diagnostic: every peer is this rank itself (tools/mr8_budget.py)
	P.nslot = 2 * NL;
	P.timeout_s = std::max(0.1, g->cfg.real(O_PUSH_TIMEOUT, g->wd.timeout_s > 0 ? g->wd.timeout_s : 300.0));
	// ---- this rank's allocations (a failure is carried into the directory, not returned: the peers are already inside the reductions)
	auto allocate = [&]() -> int {
		const size_t fbytes = sizeof(unsigned long long) * (size_t) R * P.nslot;
		HIPCHK(hipExtMallocWithFlags((void **) &P.flags, fbytes, hipDeviceMallocFinegrained));
		HIPCHK(hipMemset(P.flags, 0, fbytes));
		HIPCHK(hipMalloc((void **) &P.sent, fbytes));
		HIPCHK(hipMemset(P.sent, 0, fbytes));
		if (!P.err) HIPCHK(hipMalloc((void **) &P.err, 64));
		HIPCHK(hipMemset(P.err, 0, 64));
		if (!P.err_host) HIPCHK(hipHostMalloc((void **) &P.err_host, 64, hipHostMallocMapped));
		*P.err_host = 0;
		int rc;
		for (int l = 0; l < NL; l++) {
			LevelHost &L = *g->levels[l];
			if (!L.push_done.p) {
				if ((rc = L.push_done.alloc(2))) return rc;
			}
			HIPCHK(hipMemset(L.push_done.p, 0, 2 * sizeof(unsigned)));
			L.face_epoch = L.blk_epoch = 0; // (a set-up after a failed one starts the epochs, like the flags, from zero)
			if (L.nremote > 0) {
				if (!L.ghost_alt.p && (rc = L.ghost_alt.alloc(L.ghost.n))) return rc;
				HIPCHK(hipMemset(L.ghost_alt.p, 0, sizeof(double) * L.ghost.n));
			}
			if (L.repl_up && L.repl_direct && l + 1 < NL) {
				te_vec *cf = g->levels[l + 1]->f.get();
				if (!L.cf_alt.p && (rc = L.cf_alt.alloc(std::max<size_t>(cf->n, 2)))) return rc;
				HIPCHK(hipMemset(L.cf_alt.p, 0, sizeof(double) * std::max<size_t>(cf->n, 2)));
				L.cf_buf[0] = cf->d, L.cf_buf[1] = L.cf_alt.p;
			}
		}
		HIPCHK(hipDeviceSynchronize());
		return TE_OK;
	};
	int         local_rc  = allocate();
	// (TE_PUSH_FAULT=setup:<rank>, diagnostic: that rank's set-up "fails" after its allocations -- every rank must come back with an
	// error, nobody may wait for it, and a later attempt must start clean)
	if (!local_rc && g->cfg.has(O_PUSH_FAULT) && !strncmp(g->cfg.str(O_PUSH_FAULT), "setup:", 6) && atoi(g->cfg.str(O_PUSH_FAULT) + 6) == g->rank)
		local_rc = te::fail(TE_ENOMEM, "injected set-up failure (TE_PUSH_FAULT)");
	std::string local_msg = local_rc ? std::string(te_last_error()) : std::string();
	// ---- the directory: per rank [ok, pid, nonce (2), host, flags (handle 16 + pointer 2), per level: 4 x (handle 16 + pointer 2), R x recv offset 2]
	const int      D_OK = 0, D_PID = 1, D_NONCE = 2, D_HOST = 4, D_FLAGS = 5;
	const int      HW = 18, LW = 4 * HW + 2 * R, D_LEVELS = D_FLAGS + HW, W = D_LEVELS + NL * LW;
	std::vector<double> dir((size_t) R * W, 0.0);
	auto put = [&](double *dst, const void *devptr) { // handle + raw pointer of one allocation (null: zeros)
		if (!devptr) return TE_OK;
		hipIpcMemHandle_t h;
		HIPCHK(hipIpcGetMemHandle(&h, const_cast<void *>(devptr)));
		uint32_t w[16];
		static_assert(sizeof h == 64, "hipIpcMemHandle_t is 64 bytes");
		memcpy(w, &h, 64);
		for (int k = 0; k < 16; k++) dst[k] = (double) w[k];
		const uint64_t a = (uint64_t) (uintptr_t) devptr;
		dst[16] = (double) (uint32_t) (a & 0xFFFFFFFFu), dst[17] = (double) (uint32_t) (a >> 32);
		return TE_OK;
	};
	double *mine = &dir[(size_t) g->rank * W];
	// (TE_PUSH_FAULT=nonce, diagnostic: this rank publishes another process's nonce under its own pid -- what two containers with
	// equal pids look like; its in-process peers must then go through hipIpcOpenMemHandle, and fail or work cleanly)
	const uint64_t my_nonce = processNonce() ^ ((g->cfg.has(O_PUSH_FAULT) && !strcmp(g->cfg.str(O_PUSH_FAULT), "nonce")) ? (uint64_t) (g->rank + 1) << 8 : 0);
	auto publish = [&]() -> int {
		int rc;
		mine[D_PID]       = (double) (uint32_t) getpid();
		mine[D_NONCE]     = (double) (uint32_t) (my_nonce & 0xFFFFFFFFu);
		mine[D_NONCE + 1] = (double) (uint32_t) (my_nonce >> 32);
		mine[D_HOST]      = (double) hostHash();
		if ((rc = put(mine + D_FLAGS, P.flags))) return rc;
		for (int l = 0; l < NL; l++) {
			LevelHost &L = *g->levels[l];
			double    *q = mine + D_LEVELS + (size_t) l * LW;
			if (L.nremote > 0 && ((rc = put(q, L.ghost.p)) || (rc = put(q + HW, L.ghost_alt.p)))) return rc;
			if (L.cf_buf[0] && ((rc = put(q + 2 * HW, L.cf_buf[0])) || (rc = put(q + 3 * HW, L.cf_buf[1])))) return rc;
			for (size_t i = 0; i < L.fx.peers.size(); i++) { // where rank fx.peers[i]'s layers land in my ghost buffers (+1: 0 = nothing)
				const uint64_t o = (uint64_t) L.fx.recv_off[i] + 1;
				q[4 * HW + 2 * L.fx.peers[i]] = (double) (uint32_t) (o & 0xFFFFFFFFu), q[4 * HW + 2 * L.fx.peers[i] + 1] = (double) (uint32_t) (o >> 32);
			}
		}
		return TE_OK;
	};
	if (!local_rc && (local_rc = publish())) local_msg = te_last_error();
	mine[D_OK] = local_rc ? 0.0 : 1.0;
	int rc;
	for (size_t i = 0; i < dir.size(); i += 8) { // (a failed reduction is the transport under this one failing: nothing left to agree through)
		const int n = (int) std::min<size_t>(8, dir.size() - i);
		if (hipMemcpyAsync(g->result.p, &dir[i], n * sizeof(double), hipMemcpyHostToDevice, g->stream) != hipSuccess) {
			(void) hipGetLastError();
			pushTeardown(g);
			return te::fail(TE_EHIP, "te_gmg_use_push: hipMemcpyAsync(directory)");
		}
		if ((rc = finishReduce(g, n, 0, true))) { // (what this set-up allocated and mapped so far does not outlive it: a later attempt starts clean)
			const std::string m = te_last_error();
			pushTeardown(g);
			return te::fail(rc, m);
		}
		for (int k = 0; k < n; k++) dir[i + k] = g->result_host[k];
	}
	for (int r = 0; r < R; r++) {
		if (dir[(size_t) r * W + D_OK] == 1.0 || (self && r != g->rank)) continue; // (loop-back: nobody but this rank published anything)
		pushTeardown(g); // every rank sees the same directory: all leave here together
		if (r == g->rank) return te::fail(local_rc, "te_gmg_use_push: " + local_msg);
		return te::fail(TE_ESTATE, "te_gmg_use_push: rank " + std::to_string(r) + " could not set up its buffers");
	}
	// ---- map the peers
	const uint32_t mypid = (uint32_t) getpid();
	auto open = [&](const double *src, void **out) -> int { // handle + pointer words of a peer's allocation -> a pointer usable here
		const uint64_t raw = (uint64_t) (uint32_t) src[16] | ((uint64_t) (uint32_t) src[17] << 32);
		*out               = nullptr;
		if (raw == 0) return TE_OK;
		hipIpcMemHandle_t h;
		uint32_t          w[16];
		for (int k = 0; k < 16; k++) w[k] = (uint32_t) src[k];
		memcpy(&h, w, 64);
		void *m = nullptr;
		HIPCHK(hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess));
		P.opened.push_back(m);
		*out = m;
		return TE_OK;
	};
	auto peerPtr = [&](int r, const double *src, void *own, void **out) -> int {
		const double *slice = &dir[(size_t) r * W];
		if (self || r == g->rank) { // (loop-back: this rank's own buffer stands in for the peer's)
			*out = own;
			return TE_OK;
		}
		const uint64_t nonce = (uint64_t) (uint32_t) slice[D_NONCE] | ((uint64_t) (uint32_t) slice[D_NONCE + 1] << 32);
		if ((uint32_t) slice[D_PID] == mypid && nonce == processNonce()) { // a virtual rank in this very process: its pointer as it is
			*out = (void *) (uintptr_t) ((uint64_t) (uint32_t) src[16] | ((uint64_t) (uint32_t) src[17] << 32));
			return TE_OK;
		}
		if ((uint32_t) slice[D_HOST] != hostHash())
			return te::fail(TE_EUNSUPPORTED, "te_gmg_use_push: rank " + std::to_string(r) + " runs on another node: no direct stores across nodes");
		return open(src, out);
	};
	// (a rank whose mapping fails still takes part in the closing reduction: all ranks succeed, or all fail)
	auto mapPeers = [&]() -> int {
	P.peer_flags.assign(R, nullptr);
	for (int r = 0; r < R; r++) {
		void *m = nullptr;
		if ((rc = peerPtr(r, &dir[(size_t) r * W + D_FLAGS], P.flags, &m))) return rc;
		// loop-back: my own table stands in, shifted so that "my row of the peer's table" is the peer's row of mine
		P.peer_flags[r] = self ? P.flags + ((ptrdiff_t) r - g->rank) * P.nslot : (unsigned long long *) m;
		if (!P.peer_flags[r]) return te::fail(TE_ESTATE, "te_gmg_use_push: rank " + std::to_string(r) + " published no flag table");
	}
	for (int l = 0; l < NL; l++) {
		LevelHost &L = *g->levels[l];
		if (L.nremote > 0 && !L.fx.empty()) {
			for (int b = 0; b < 2; b++) L.push_peer_ghost[b].assign(L.fx.peers.size(), nullptr);
			bool ok = true;
			for (size_t i = 0; i < L.fx.peers.size() && ok; i++) {
				const int     r = L.fx.peers[i];
				const double *q = &dir[(size_t) r * W + D_LEVELS + (size_t) l * LW];
				const uint64_t o1 = (uint64_t) (uint32_t) q[4 * HW + 2 * g->rank] | ((uint64_t) (uint32_t) q[4 * HW + 2 * g->rank + 1] << 32);
				const int64_t off = self ? L.fx.recv_off[i] : (int64_t) o1 - 1;
				if (off < 0) { // the peer expects nothing from me here although I send: the plans disagree
					ok = false;
					break;
				}
				for (int b = 0; b < 2; b++) {
					void *m = nullptr;
					// (loop-back: the OTHER buffer of this rank, so that what is being read is not overwritten)
					if ((rc = peerPtr(r, q + b * HW, b ? L.ghost.p : L.ghost_alt.p, &m))) return rc;
					if (!m) ok = false;
					L.push_peer_ghost[b][i] = m ? (double *) m + off : nullptr;
				}
			}
			if (!ok) return te::fail(TE_ESTATE, "te_gmg_use_push: a neighbour rank published no receive buffer for level " + std::to_string(l));
			// where every face of the send order goes (pack + push in one launch), and the flags that launch raises
			std::vector<PushFlag> fl;
			for (int b = 0; b < 2; b++) {
				std::vector<double *> dst((size_t) L.nremote, nullptr);
				for (size_t i = 0; i < L.fx.peers.size(); i++)
					for (int64_t k = 0; k < L.fx.send_cnt[i] / (int64_t) L.nf; k++)
						dst[(size_t) (L.fx.send_off[i] / (int64_t) L.nf + k)] = L.push_peer_ghost[b][i] + k * (int64_t) L.nf;
				int rc2 = L.push_face_dst[b].upload(dst);
				if (rc2) return rc2;
			}
			for (size_t i = 0; i < L.fx.peers.size(); i++) fl.push_back(pushFlagFor(g, L.fx.peers[i], 2 * l)); // (every peer of the plan: symmetric)
			int rc3 = L.push_face_flags.upload(fl);
			if (rc3) return rc3;
			L.push_faces = true;
		}
		if (L.cf_buf[0] && !L.tx_direct.empty()) {
			for (int b = 0; b < 2; b++) L.push_peer_cf[b].assign(L.tx_direct.peers.size(), nullptr);
			for (size_t i = 0; i < L.tx_direct.peers.size(); i++) {
				const int     r = L.tx_direct.peers[i];
				const double *q = &dir[(size_t) r * W + D_LEVELS + (size_t) l * LW];
				for (int b = 0; b < 2; b++) {
					void *m = nullptr;
					if ((rc = peerPtr(r, q + (2 + b) * HW, L.cf_buf[b ^ 1], &m))) return rc;
					if (!m) return te::fail(TE_ESTATE, "te_gmg_use_push: a rank published no coarse buffer for level " + std::to_string(l));
					L.push_peer_cf[b][i] = (double *) m;
				}
			}
			L.push_blocks = true;
		}
	}
	return TE_OK;
	};
	const int         map_rc  = mapPeers();
	const std::string map_msg = map_rc ? std::string(te_last_error()) : std::string();
	// nobody pushes before everybody has finished mapping (and zeroing): one more reduction, which also carries "somebody failed"
	double failed = map_rc ? 1.0 : 0.0;
	if (hipMemcpyAsync(g->result.p, &failed, sizeof failed, hipMemcpyHostToDevice, g->stream) != hipSuccess) {
		(void) hipGetLastError();
		pushTeardown(g);
		return te::fail(TE_EHIP, "te_gmg_use_push: hipMemcpyAsync(failed)");
	}
	if ((rc = finishReduce(g, 1, 1, true))) {
		const std::string m = te_last_error();
		pushTeardown(g);
		return te::fail(rc, m);
	}
	if (map_rc || g->result_host[0] != 0.0) {
		pushTeardown(g);
		if (map_rc) return te::fail(map_rc, map_msg);
		return te::fail(TE_ESTATE, "te_gmg_use_push: another rank could not map its peers' buffers");
	}
	P.ready = true;
	return TE_OK;
}

int te_gmg_use_push(te_gmg *g, int enable)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_use_push: null");
		// Every switch between the transports is a point where ALL ranks have finished what they had queued: the argument that
		// lets a rank run one exchange ahead of a peer (LevelHost::ghost_alt) needs every exchange to be a direct one -- a rank that
		// starts pushing while a peer still reads its buffers in a cycle of the other transport would overwrite them. Collective.
		auto meet = [&]() -> int {
			HIPCHK(hipStreamSynchronize(g->stream));
			HIPCHK(hipStreamSynchronize(g->comm_stream));
			if (g->nranks < 2 || (!g->rccl.comm && !g->allreduce)) return TE_OK;
			double one = 1.0;
			HIPCHK(hipMemcpyAsync(g->result.p, &one, sizeof one, hipMemcpyHostToDevice, g->stream));
			return finishReduce(g, 1, 0, true);
		};
		int rc;
		if (!enable) {
			if (!g->push.on) return TE_OK;
			if ((rc = meet())) return rc;
			// back to the other transport: the coarse vectors return to their own storage
			for (size_t l = 0; l + 1 < g->levels.size(); l++)
				if (g->levels[l]->cf_buf[0]) g->levels[l + 1]->f->d = g->levels[l]->cf_buf[0];
			for (auto &L : g->levels) L->ghost_par = 0;
			g->push.on = false;
			return TE_OK;
		}
		if (g->push.on) return TE_OK;
		if (g->push.rejected)
			return te::fail(TE_ESTATE, "te_gmg_use_push: te_gmg_autotune rejected the direct-store transport on this machine (its result differed from "
			                           "the other transport's, or a wait gave up)");
		if ((rc = pushSetup(g)) || (rc = meet())) return rc;
		g->push.on = true;
		return TE_OK;
	});
}

// 0: no direct-store exchange has failed; otherwise the first failure's code (pushkernels.hpp PushErr): 1 a wait gave up (its
// data never arrived within TE_PUSH_TIMEOUT seconds), 2 a peer's flag was two exchanges ahead, 3 a peer was behind when its buffer
// was overwritten, 4 this rank's own epochs were out of sequence. The results since then are garbage, and every later exchange
// of this solver returns at once. Reads the pinned host copy: no device call.
int te_gmg_push_failed(te_gmg *g)
{
	return guarded([&]() -> int { return (g && g->push.err_host) ? *g->push.err_host : 0; });
}

// moves n doubles from a scratch send buffer to a scratch receive buffer of level 0 through the same
// code path as a real exchange, with this rank as its own peer; returns TE_OK iff the data arrived intact
int te_gmg_exchange_selftest(te_gmg *g, int n)
{
	return guarded([&]() -> int {
		if (!g || n < 1) return te::fail(TE_EINVAL, "te_gmg_exchange_selftest: bad argument");
		LevelHost &L = *g->levels[0];
		if ((size_t) 2 * n > L.r->n) return te::fail(TE_EINVAL, "te_gmg_exchange_selftest: n too large");
		std::vector<double> h(n), back(n);
		for (int i = 0; i < n; i++) h[i] = 1.0 + i * 0.5;
		double *send = L.r->d, *recv = L.r->d + n;
		HIPCHK(hipMemcpyAsync(send, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, g->stream));
		HIPCHK(hipMemsetAsync(recv, 0, sizeof(double) * n, g->stream));
		ExPlan pl;
		int    me = 0;
		pl.peers  = {me};
		pl.send_off = {0}, pl.send_cnt = {n}, pl.recv_off = {0}, pl.recv_cnt = {n};
		int rc = doExchange(g, 9, pl, send, recv);
		if (rc) return rc;
		HIPCHK(hipMemcpyAsync(back.data(), recv, sizeof(double) * n, hipMemcpyDeviceToHost, g->stream));
		HIPCHK(hipStreamSynchronize(g->stream));
		for (int i = 0; i < n; i++)
			if (back[i] != h[i]) return te::fail(TE_ESTATE, "te_gmg_exchange_selftest: data mismatch");
		if (g->rccl.comm) { // the scalar reduction of te_bicgstab / te_gmg_verify_schedule: ncclAllReduce on the solver stream
			const double v[4] = {1.5, -2.25, 3.0, 0.125};
			HIPCHK(hipMemcpyAsync(g->result.p, v, sizeof v, hipMemcpyHostToDevice, g->stream));
			int r2 = g->rccl.AllReduce(g->result.p, g->result.p, 4, ncclFloat64, ncclSum, g->rccl.comm, g->stream);
			if (r2) return te::fail(TE_ESTATE, std::string("ncclAllReduce failed: ") + g->rccl.GetErrorString(r2));
			HIPCHK(hipMemcpyAsync(g->result_host, g->result.p, sizeof v, hipMemcpyDeviceToHost, g->stream));
			HIPCHK(hipStreamSynchronize(g->stream));
			for (int i = 0; i < 4; i++)
				if (g->result_host[i] != v[i] * g->nranks) return te::fail(TE_ESTATE, "te_gmg_exchange_selftest: all-reduce mismatch");
		}
		return TE_OK;
	});
}

// The communicator the native back-end really runs on: ncclCommCount / ncclCommUserRank of te_gmg_use_rccl's communicator
// (0 / -1 without one) -- so that a bench line can show that RCCL saw N ranks rather than say so.
int te_gmg_comm_info(te_gmg *g, int *rccl_nranks, int *rccl_rank)
{
	return guarded([&]() -> int {
		if (!g) return te::fail(TE_EINVAL, "te_gmg_comm_info: null");
		int n = 0, r = -1;
		if (g->rccl.comm && g->rccl.CommCount && g->rccl.CommUserRank) {
			int rc = g->rccl.CommCount(g->rccl.comm, &n);
			if (rc == 0) rc = g->rccl.CommUserRank(g->rccl.comm, &r);
			if (rc) return te::fail(TE_ESTATE, std::string("ncclCommCount failed: ") + g->rccl.GetErrorString(rc));
		}
		if (rccl_nranks) *rccl_nranks = n;
		if (rccl_rank) *rccl_rank = r;
		return TE_OK;
	});
}

// Diagnostic for the watchdog's bookkeeping: for `seconds` of wall time the host enqueues, WITHOUT ever synchronising, a
// level-0 vector kernel followed by an armed "exchange" (this rank as its own peer through the active back-end, or a
// device-to-device copy when none is set). The host runs ahead of the GPU, so the newest exchange is never complete when
// the watchdog polls; every exchange does complete within milliseconds, so a correct watchdog (deadline of the OLDEST
// outstanding exchange) stays quiet even when `seconds` exceeds TE_EXCHANGE_TIMEOUT. Returns the number of exchanges issued.
int te_gmg_watchdog_selftest(te_gmg *g, double seconds)
{
	return guarded([&]() -> int {
			if (!g || seconds <= 0) return te::fail(TE_EINVAL, "te_gmg_watchdog_selftest: bad argument");
			watchdogStart(g);
			LevelHost &L = *g->levels[0];
			const int  n = 256;
			if ((size_t) 2 * n > L.r->n) return te::fail(TE_EINVAL, "te_gmg_watchdog_selftest: level 0 too small");
			double *send = L.r->d, *recv = L.r->d + n;
			ExPlan  pl;
			pl.peers    = {g->rank};
			pl.send_off = {0}, pl.send_cnt = {n}, pl.recv_off = {0}, pl.recv_cnt = {n};
			const auto t0 = std::chrono::steady_clock::now();
			int        count = 0;
			while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
				int rc = vecop<VOP_SCALE>(L.t.get(), nullptr, nullptr, 1.0, 0, 0);
				if (rc) return rc;
				if (g->rccl.comm || g->exchange) {
					if ((rc = doExchange(g, 9, pl, send, recv))) return rc;
				} else {
					WatchdogArm arm(g, g->stream, 9);
					HIPCHK(hipMemcpyAsync(recv, send, sizeof(double) * n, hipMemcpyDeviceToDevice, g->stream));
				}
				count++;
			}
			HIPCHK(hipStreamSynchronize(g->stream));
			return count;
	});
}

} // extern "C"
