"""Share-exchange transports for Python hosts.

* `attach_rccl(engine, ...)` - the product path: the native RCCL transport of libcognn_hip.so (include/cognn_exchange.h,
  csrc/exchange_rccl.hip).  Python only bootstraps it (the 128-byte communicator id travels over the existing
  torch.distributed group, or over the library's own TCP rendezvous); every round afterwards is ncclSend/ncclRecv issued
  from C++ on a communication stream - no interpreter in the loop.
* `make_exchange[_async]` - the same rounds over torch.distributed p2p ops, kept as the TEST transport: "gloo" lets the
  multi-rank engine logic run on CPU-only machines and lets several ranks share one GPU (host-staged), which RCCL cannot.

Both replace the reference's TCP paths (CommSync::send/recvShareVecVec, include/comm_sync.h:245-277; TaskComm; Engine's
channel mesh, include/engine.h:157-201): one logical message = one p2p op on a flat uint64 buffer, all messages of a
round issued as one p2p group."""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from .engine_api import EXCHANGE_FN, EXCHANGE_WAIT_FN, EXCHANGE_WAIT_ROUND_FN, RCCL_ID_BYTES


class RcclExchange:
    """Owner of a cognn_rccl_exchange handle."""

    def __init__(self, lib, handle, rank, world):
        self.lib = lib; self.h = handle; self.rank = rank; self.world = world

    def stats(self):
        r = ctypes.c_int64(); s = ctypes.c_int64(); v = ctypes.c_int64()
        if self.lib.cognn_rccl_exchange_stats(self.h, ctypes.byref(r), ctypes.byref(s), ctypes.byref(v)) != 0:
            raise RuntimeError(self.lib.cognn_exchange_last_error().decode())
        ms = ctypes.c_double()
        if self.lib.cognn_rccl_exchange_time(self.h, ctypes.byref(ms)) != 0:
            raise RuntimeError(self.lib.cognn_exchange_last_error().decode())
        return {"rounds": r.value, "bytes_sent": s.value, "bytes_received": v.value, "comm_ms": ms.value}

    def ranks(self):
        """(ncclCommCount, ncclCommUserRank, sum of an all-reduce of ones over the communicator); collective."""
        c = ctypes.c_int32(); r = ctypes.c_int32(); ones = ctypes.c_int64()
        if self.lib.cognn_rccl_exchange_ranks(self.h, ctypes.byref(c), ctypes.byref(r), ctypes.byref(ones)) != 0:
            raise RuntimeError(self.lib.cognn_exchange_last_error().decode())
        return c.value, r.value, ones.value

    def barrier(self):
        if self.lib.cognn_rccl_exchange_barrier(self.h) != 0:
            raise RuntimeError(self.lib.cognn_exchange_last_error().decode())

    def close(self):
        if self.h:
            self.lib.cognn_rccl_exchange_destroy(self.h)
            self.h = None


def create_rccl_exchange(rank, world, device_index, stream, group=None, tcp=None):
    """Creates the native transport for this rank.  The communicator id comes from rank 0 through `group` (any
    torch.distributed backend) or, with tcp=(addr, port), through the library's own rendezvous."""
    from . import capi
    lib = capi.load()
    if not hasattr(lib, "cognn_rccl_exchange_create"):
        raise capi.CognnError("this engine library was built without the RCCL transport")
    ident = (ctypes.c_uint8 * RCCL_ID_BYTES)()
    if tcp is not None:
        if lib.cognn_rccl_rendezvous_tcp(tcp[0].encode(), int(tcp[1]), rank, world, 120.0, ident) != 0:
            raise capi.CognnError(lib.cognn_exchange_last_error().decode())
    else:
        if rank == 0 and lib.cognn_rccl_unique_id(ident) != 0:
            raise capi.CognnError(lib.cognn_exchange_last_error().decode())
        if world > 1:
            box = [bytes(ident) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            ctypes.memmove(ident, box[0], RCCL_ID_BYTES)
    h = ctypes.c_void_p()
    if lib.cognn_rccl_exchange_create(ident, rank, world, device_index, ctypes.c_void_p(stream), ctypes.byref(h)) != 0:
        raise capi.CognnError(lib.cognn_exchange_last_error().decode())
    return RcclExchange(lib, h, rank, world)


def attach_rccl(engine, device_index, stream=None, group=None, tcp=None):
    """Gives `engine` the native RCCL transport (see module docstring); returns the RcclExchange (keep it alive)."""
    if stream is None:
        stream = torch.cuda.current_stream(device_index).cuda_stream
    x = create_rccl_exchange(engine.rank, engine.world, device_index, stream, group=group, tcp=tcp)
    if x.lib.cognn_engine_set_exchange_rccl(engine.h, x.h) != 0:
        raise RuntimeError(x.lib.cognn_exchange_last_error().decode())
    engine._xfn = x
    return x


class _DevBuf:
    """Raw device pointer exposed through __cuda_array_interface__ so torch can alias it without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _wrap(ptr, nbytes, device):
    if device.type == "cuda":
        return torch.as_tensor(_DevBuf(ptr, nbytes), device=device)
    buf = (ctypes.c_uint8 * nbytes).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.uint8))


def make_exchange(device, group=None, host_staged=False):
    """Returns the cognn_exchange_fn callback for cognn_engine_set_exchange().

    host_staged: move every message through a host bounce buffer (device -> host, p2p, host -> device).  For process
    groups whose backend cannot send device memory (gloo) - e.g. several ranks sharing one GPU in the tests, or a cluster
    without GPU-direct transport; with RCCL leave it off."""

    # The engine's buffers are persistent, so the same transfer lists recur every iteration: the wrapped tensors and the
    # P2POp lists are built once per distinct list and reused (wrapping a raw pointer costs tens of microseconds each, a
    # pass has dozens of rounds).
    cache = {}

    def _build(xfers, n):
        ops, post, pre = [], [], []
        for i in range(n):
            x = xfers[i]
            t = _wrap(x.ptr, x.bytes, device)
            if host_staged and device.type == "cuda":
                h = torch.empty(x.bytes, dtype=torch.uint8)
                (pre if x.is_send else post).append((t, h))
                ops.append(dist.P2POp(dist.isend if x.is_send else dist.irecv, h, x.peer, group))
            else:
                ops.append(dist.P2POp(dist.isend if x.is_send else dist.irecv, t, x.peer, group))
        return ops, pre, post

    def _exchange(user, xfers, n):
        try:
            if n <= 0:
                return 0
            key = tuple((xfers[i].ptr, xfers[i].bytes, xfers[i].peer, xfers[i].is_send) for i in range(n))
            ent = cache.get(key)
            if ent is None:
                ent = cache[key] = _build(xfers, n)
            ops, pre, post = ent
            for t, h in pre:
                h.copy_(t)                                  # device -> host, synchronises with the engine's (default) stream
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for t, h in post:
                t.copy_(h)
            return 0
        except Exception as ex:  # noqa: BLE001 - the C caller only understands a status code
            print("cognn exchange failed: %r" % (ex,), flush=True)
            return 1

    return EXCHANGE_FN(_exchange)


def make_exchange_async(device, group=None, host_staged=False, per_round=False):
    """(begin, wait) callbacks for cognn_engine_set_exchange_async(): begin enqueues a round and returns, wait completes every
    enqueued round - the engine runs the kernels of the sides whose peer is local in between.  With RCCL the p2p group runs on
    the communicator's stream (ordered after the kernels already launched), wait makes the engine's stream wait for it.
    per_round: also return the third callback (cognn_exchange_wait_round_fn: completes the rounds up to a given one and leaves
    the later ones in flight - the chunked pipelines of COGNN_OPT_EXCHANGE_CHUNKS), for cognn_engine_set_exchange_async2()."""
    cache = {}
    inflight = []                                       # (works, [(device tensor, host tensor)] to copy back)
    counts = [0, 0]                                     # rounds begun / completed

    def _build(xfers, n):
        ops, pre, post = [], [], []
        for i in range(n):
            x = xfers[i]
            t = _wrap(x.ptr, x.bytes, device)
            if host_staged and device.type == "cuda":
                h = torch.empty(x.bytes, dtype=torch.uint8)
                (pre if x.is_send else post).append((t, h))
                ops.append(dist.P2POp(dist.isend if x.is_send else dist.irecv, h, x.peer, group))
            else:
                ops.append(dist.P2POp(dist.isend if x.is_send else dist.irecv, t, x.peer, group))
        return ops, pre, post

    def _begin(user, xfers, n):
        try:
            if n <= 0:
                return 0
            key = tuple((xfers[i].ptr, xfers[i].bytes, xfers[i].peer, xfers[i].is_send) for i in range(n))
            ent = cache.get(key)
            if ent is None:
                ent = cache[key] = _build(xfers, n)
            ops, pre, post = ent
            for t, h in pre:
                h.copy_(t)                                  # device -> host of what is already enqueued on the engine's stream
            inflight.append((dist.batch_isend_irecv(ops), post))
            counts[0] += 1
            return 0
        except Exception as ex:  # noqa: BLE001 - the C caller only understands a status code
            print("cognn exchange (begin) failed: %r" % (ex,), flush=True)
            return 1

    def _complete(upto):
        try:
            while inflight and counts[1] < upto:
                works, post = inflight.pop(0)
                for w in works:
                    w.wait()
                for t, h in post:
                    t.copy_(h)
                counts[1] += 1
            return 0
        except Exception as ex:  # noqa: BLE001
            print("cognn exchange (wait) failed: %r" % (ex,), flush=True)
            return 1

    def _wait(user):
        return _complete(counts[0])

    def _wait_round(user, rnd):
        return _complete(rnd + 1)

    if per_round:
        return EXCHANGE_FN(_begin), EXCHANGE_WAIT_FN(_wait), EXCHANGE_WAIT_ROUND_FN(_wait_round)
    return EXCHANGE_FN(_begin), EXCHANGE_WAIT_FN(_wait)
