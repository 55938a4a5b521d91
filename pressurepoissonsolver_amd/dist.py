"""Ghost / transfer exchange back-ends for te_gmg_set_exchange (include/te_hip.h).

The native library packs the face layers (or restricted blocks) other ranks need into one send
buffer and asks its host for ONE exchange: "for every peer r move send[send_off[r] : +send_cnt[r]]
to r and fill recv[recv_off[r] : +recv_cnt[r]] from r". This replaces the PETSc VecScatter pairs
of src/Thunderegg/SchurHelper.h:123-150 and GMG/InterLevelComm.h:169-189.

  attach(gmg, dist)   one process per GPU, torch.distributed point-to-point: backend "nccl" is
                      RCCL over xGMI (device buffers, batched isend/irecv in one group); backend
                      "gloo" stages through host memory (CPU rehearsal on a single-GPU box).
  LocalFabric(n)      n virtual ranks inside ONE process (threads), device-to-device copies;
                      lets one GPU exercise the 2/4/8-rank plans of the native library.
"""
import ctypes as C
import threading

import numpy as np

from . import capi


class _DevArray:
    """__cuda_array_interface__ view of raw device memory (no ownership)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _plan_arrays(npeers, peers, soff, scnt, roff, rcnt):
    return ([int(peers[i]) for i in range(npeers)], [int(soff[i]) for i in range(npeers)],
            [int(scnt[i]) for i in range(npeers)], [int(roff[i]) for i in range(npeers)],
            [int(rcnt[i]) for i in range(npeers)])


def p2p_exchange(dist, send, recv, peers, soff, scnt, roff, rcnt):
    """The torch.distributed exchange itself, on torch tensors (device tensors for nccl, CPU tensors
    for gloo). One batch = one NCCL group: all sends and receives of this exchange progress
    together, so the order of peers cannot deadlock."""
    ops = []
    for r, so, sc, ro, rc in zip(peers, soff, scnt, roff, rcnt):
        if rc > 0:
            ops.append(dist.P2POp(dist.irecv, recv[ro:ro + rc], r))
        if sc > 0:
            ops.append(dist.P2POp(dist.isend, send[so:so + sc], r))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def attach(gmg, dist):
    """Bind gmg's exchange callback to torch.distributed (already initialised).

    The tensors aliasing the library's buffers and the P2POp lists are built once per distinct
    exchange (the plans are static), so the steady-state callback is: enter the solver stream,
    batch_isend_irecv, wait."""
    import torch
    backend = dist.get_backend()
    cache = {}

    def build(send_ptr, recv_ptr, npeers, peers, soff, scnt, roff, rcnt, stream):
        pl = _plan_arrays(npeers, peers, soff, scnt, roff, rcnt)
        ns = max((o + c for o, c in zip(pl[1], pl[2])), default=0)
        nr = max((o + c for o, c in zip(pl[3], pl[4])), default=0)
        ext = torch.cuda.ExternalStream(int(stream))
        with torch.cuda.stream(ext):
            send = torch.as_tensor(_DevArray(send_ptr, max(ns, 1)), device="cuda")
            recv = torch.as_tensor(_DevArray(recv_ptr, max(nr, 1)), device="cuda")
        ent = {"ext": ext, "send": send, "recv": recv, "pl": pl, "nr": nr}
        if backend == "nccl":
            ops = []
            for r, so, sc, ro, rc in zip(*pl):
                if rc > 0:
                    ops.append(dist.P2POp(dist.irecv, recv[ro:ro + rc], r))
                if sc > 0:
                    ops.append(dist.P2POp(dist.isend, send[so:so + sc], r))
            ent["ops"] = ops
        return ent

    def cb(user, tag, send_ptr, recv_ptr, npeers, peers, soff, scnt, roff, rcnt, stream):
        try:
            # the library issues the same exchange on two streams (overlapped on its communication stream, or in
            # line on the solver stream): the stream is part of the key, an entry never serves another stream
            key = (tag, send_ptr, recv_ptr, npeers, int(stream or 0))
            ent = cache.get(key)
            if ent is None:
                ent = cache[key] = build(send_ptr, recv_ptr, npeers, peers, soff, scnt, roff, rcnt, stream)
            with torch.cuda.stream(ent["ext"]):
                if backend == "nccl":
                    # enqueued behind the pack kernel on the solver stream; wait() orders the solver
                    # stream behind the transfers - no host synchronisation
                    if ent["ops"]:
                        for w in dist.batch_isend_irecv(ent["ops"]):
                            w.wait()
                else:
                    pl = ent["pl"]
                    hs = ent["send"].cpu()  # synchronises the solver stream
                    hr = torch.empty(max(ent["nr"], 1), dtype=torch.float64)
                    p2p_exchange(dist, hs, hr, *pl)
                    for ro, rc in zip(pl[3], pl[4]):
                        if rc > 0:
                            ent["recv"][ro:ro + rc].copy_(hr[ro:ro + rc])
                    ent["ext"].synchronize()
            return 0
        except Exception:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    gmg._cb = capi.EXCHANGE_FN(cb)
    capi.check(capi.lib().te_gmg_set_exchange(gmg.h, gmg._cb, None))
    # scalar reductions of te_bicgstab / te_gmg_verify_schedule (Vector.h:294,306,319 MPI_Allreduce)
    gmg.set_allreduce(lambda vals, op: allreduce(dist, vals, op))


def rccl_library():
    """The librccl.so the process already uses (torch's), so that only one RCCL runtime is loaded."""
    import glob
    import os
    import torch
    hits = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))
    return hits[0] if hits else "/opt/rocm/lib/librccl.so"


def attach_rccl(gmg, dist, rank, world):
    """Give the native library its own RCCL communicator (ncclCommInitRank with an id made on rank 0 and
    broadcast through torch.distributed); exchanges then never enter Python. Raises TeError on failure."""
    import torch
    lib = rccl_library().encode()
    ident = C.create_string_buffer(128)
    if rank == 0:
        capi.check(capi.lib().te_rccl_unique_id(lib, ident))
    if dist is not None and world > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(list(ident.raw), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        ident = C.create_string_buffer(bytes(t.cpu().tolist()), 128)
    capi.check(capi.lib().te_gmg_use_rccl(gmg.h, lib, ident, rank, world))


class LocalFabric:
    """n virtual ranks in one process. Each rank runs in its own thread (`run`). An exchange is a set of
    point-to-point rendezvous: the sender posts (pointer, count, tag) to the (src, dst) mailbox, the
    receiver copies device-to-device and acknowledges; ranks whose plan is empty do not take part
    (exactly the semantics of the torch.distributed binding)."""

    def __init__(self, nranks):
        import queue
        self.n = nranks
        self.barrier = threading.Barrier(nranks)
        self.mail = {(s, d): queue.Queue() for s in range(nranks) for d in range(nranks)}
        self.acks = {(s, d): queue.Queue() for s in range(nranks) for d in range(nranks)}
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemcpy.restype = C.c_int
        self.hip.hipDeviceSynchronize.restype = C.c_int
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.hip.hipStreamSynchronize.restype = C.c_int
        self.errors = []
        self.timeout = 120.0

    def attach(self, gmg, rank):
        fab = self

        def cb(user, tag, send_ptr, recv_ptr, npeers, peers, soff, scnt, roff, rcnt, stream):
            try:
                pl = _plan_arrays(npeers, peers, soff, scnt, roff, rcnt)
                capi.check(capi.lib().te_gmg_sync(gmg.h))  # my packed data is complete (solver stream) ...
                assert fab.hip.hipStreamSynchronize(C.c_void_p(stream)) == 0  # ... and whatever `stream` still holds
                for r, so, sc in zip(pl[0], pl[1], pl[2]):
                    if sc > 0:
                        fab.mail[(rank, r)].put((int(send_ptr) + 8 * so, sc, tag))
                got = []
                for r, ro, rc in zip(pl[0], pl[3], pl[4]):
                    if rc == 0:
                        continue
                    sp, sc, stag = fab.mail[(r, rank)].get(timeout=fab.timeout)
                    assert stag == tag and sc == rc, f"plan mismatch {rank}<-{r}: tag {stag}/{tag} cnt {sc}/{rc}"
                    err = fab.hip.hipMemcpy(int(recv_ptr) + 8 * ro, sp, 8 * rc, 3)  # DeviceToDevice
                    assert err == 0, f"hipMemcpy failed: {err}"
                    got.append(r)
                # a device-to-device hipMemcpy is only ORDERED on the null stream, not complete on return;
                # the solver streams are non-blocking, so finish the copies before anyone proceeds
                assert fab.hip.hipDeviceSynchronize() == 0
                for r in got:
                    fab.acks[(r, rank)].put(tag)
                for r, sc in zip(pl[0], pl[2]):  # nobody may repack before its receivers have copied
                    if sc > 0:
                        fab.acks[(rank, r)].get(timeout=fab.timeout)
                return 0
            except Exception as e:
                fab.errors.append(e)
                return 1

        gmg._cb = capi.EXCHANGE_FN(cb)
        capi.check(capi.lib().te_gmg_set_exchange(gmg.h, gmg._cb, None))
        gmg.set_allreduce(self.allreduce(rank))

    def allreduce(self, rank):
        """Deterministic sum / max over the virtual ranks (rank order): red(values, op=0) -> list."""
        fab = self
        if not hasattr(fab, "_red"):
            fab._red = [None] * fab.n

        def red(values, op=0):
            fab._red[rank] = list(values)
            fab.barrier.wait(timeout=fab.timeout)
            f = max if op else sum
            out = [f(fab._red[r][i] for r in range(fab.n)) for i in range(len(values))]
            fab.barrier.wait(timeout=fab.timeout)
            return out

        return red

    def run(self, fn):
        """fn(rank) in one thread per rank; returns the list of results."""
        out, errs = [None] * self.n, []

        def work(r):
            try:
                out[r] = fn(r)
            except Exception as e:
                errs.append(e)
                self.barrier.abort()

        ts = [threading.Thread(target=work, args=(r,)) for r in range(self.n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs or self.errors:
            raise (self.errors + errs)[0]
        return out


def allreduce(dist, values, op=0):
    """sum (op 0) or max (op 1) of a few host scalars over ranks (norms / dots: Vector.h:294,306,319)."""
    if dist is None:
        return list(values)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM)
    return t.tolist()


def allreduce_sum(dist, values):
    return allreduce(dist, values, 0)
