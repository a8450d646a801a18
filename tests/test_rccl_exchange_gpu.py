"""The native RCCL transport (include/cognn_exchange.h, csrc/exchange_rccl.hip) on the one GPU of the test box.

RCCL refuses two ranks on one device, so what can run here is the world-size-1 communicator: ncclCommInitRank, the TCP
rendezvous in its rank-0 form, and rounds whose messages go from this rank to itself (ncclSend + ncclRecv to the own rank
inside one group) - the same begin / wait code path, streams and events a multi-GPU run uses.  N > 1 on real hardware is
UNMEASURED until the driver's multi-GPU run."""
import ctypes

import numpy as np
import pytest

import gpu_util as gu

pytestmark = pytest.mark.gpu


def _lib():
    from cognn_amd import capi
    return capi.load()


def test_world1_loopback_rounds_are_stream_ordered():
    import torch
    from cognn_amd import capi, dist as cdist
    from cognn_amd.engine_api import Xfer
    lib = _lib()
    stream = torch.cuda.current_stream(0).cuda_stream
    x = cdist.create_rccl_exchange(0, 1, 0, stream)
    ctx = capi.Context(0)
    n = 1 << 20
    rng = np.random.default_rng(0)
    a = gu.dev(gu.rand_u64(rng, n)); b = gu.dev_empty(n); c = gu.dev_empty(n); d = gu.dev_empty(n)
    want_b = None
    for rnd in range(3):
        # a kernel on the compute stream writes the send buffer right before the round starts ...
        ctx.call("cognn_prng_fill_u64", gu.ptr(a), 1234 + rnd, n)
        xf = (Xfer * 4)(Xfer(0, 1, a.data_ptr(), n * 8), Xfer(0, 0, b.data_ptr(), n * 8),
                        Xfer(0, 1, a.data_ptr() + 4096, 8 * 100), Xfer(0, 0, c.data_ptr(), 8 * 100))
        assert lib.cognn_rccl_exchange_begin(x.h, xf, 4) == 0, lib.cognn_exchange_last_error()
        assert lib.cognn_rccl_exchange_wait(x.h) == 0
        # ... and a kernel enqueued after the wait reads the received one
        ctx.call("cognn_add_u64", gu.ptr(d), gu.ptr(b), gu.ptr(b), n)
        ctx.sync()
        want_b = gu.host(a).copy()
        assert np.array_equal(gu.host(b), want_b)
        assert np.array_equal(gu.host(c)[:100], want_b[512:612])
        with np.errstate(over="ignore"):
            assert np.array_equal(gu.host(d), want_b + want_b)
    st = x.stats()
    assert st["rounds"] == 3 and st["bytes_sent"] == st["bytes_received"] == 3 * (n * 8 + 800)
    assert st["comm_ms"] > 0
    x.barrier()
    bad = (Xfer * 1)(Xfer(3, 1, a.data_ptr(), 8))
    assert lib.cognn_rccl_exchange_begin(x.h, bad, 1) != 0 and b"malformed" in lib.cognn_exchange_last_error()
    ctx.close()
    x.close()


def test_world1_loopback_per_round_waits():
    """cognn_rccl_exchange_wait_round: three chunk rounds in flight, each consumer waits for its own round's event only (the
    pattern of COGNN_OPT_EXCHANGE_CHUNKS); more rounds than the event ring has slots; a round that was never started is refused."""
    import torch
    from cognn_amd import capi, dist as cdist
    from cognn_amd.engine_api import Xfer
    lib = _lib()
    stream = torch.cuda.current_stream(0).cuda_stream
    x = cdist.create_rccl_exchange(0, 1, 0, stream)
    ctx = capi.Context(0)
    n = 3 << 18
    a = gu.dev_empty(n); b = gu.dev_empty(n); d = gu.dev_empty(n)
    rounds = 0
    for rep in range(30):                                    # 90 rounds > 64 ring slots
        lo = ctypes.c_int64(); hi = ctypes.c_int64()
        for c in range(3):                                   # open chunk c (a kernel under the chunk window), send it at once
            ctx.call("cognn_ctx_set_chunk", c, 3)
            ctx.call("cognn_prng_fill_u64", gu.ptr(a), 77 + rep, n)
            ctx.call("cognn_ctx_set_chunk", 0, 1)
            lib.cognn_chunk_range(n, c, 3, ctypes.byref(lo), ctypes.byref(hi))
            xf = (Xfer * 2)(Xfer(0, 1, a.data_ptr() + 8 * lo.value, 8 * (hi.value - lo.value)),
                            Xfer(0, 0, b.data_ptr() + 8 * lo.value, 8 * (hi.value - lo.value)))
            assert lib.cognn_rccl_exchange_begin(x.h, xf, 2) == 0, lib.cognn_exchange_last_error()
            rounds += 1
        for c in range(3):                                   # close chunk c after ITS round
            assert lib.cognn_rccl_exchange_wait_round(x.h, rounds - 3 + c) == 0, lib.cognn_exchange_last_error()
            ctx.call("cognn_ctx_set_chunk", c, 3)
            ctx.call("cognn_add_u64", gu.ptr(d), gu.ptr(b), gu.ptr(b), n)
            ctx.call("cognn_ctx_set_chunk", 0, 1)
        if rep % 10 == 9:
            ctx.sync()
            want = gu.host(a).copy()
            assert np.array_equal(gu.host(b), want)
            with np.errstate(over="ignore"):
                assert np.array_equal(gu.host(d), want + want)
    assert lib.cognn_rccl_exchange_wait_round(x.h, rounds) != 0 and b"not been started" in lib.cognn_exchange_last_error()
    assert lib.cognn_rccl_exchange_wait(x.h) == 0
    ctx.sync()
    ctx.close()
    x.close()


def test_tcp_rendezvous_rank0_and_engine_attach():
    """The harness bootstrap path: rendezvous (world 1: no listener needed), communicator, engine attach; a world-1 engine
    never starts a round, and its results equal the plain single-rank run."""
    import cognn_oracle as co
    from cognn_amd import dist as cdist
    from cognn_amd.engine import Engine, GnnParam
    k, V = 2, 60
    src, dst = co.synth_graph(V, 150, 1)
    part = np.array([v % k for v in range(V)], dtype=np.int32)
    feats, labels = co.synth_features(V, 24, 5, 2, density=0.2)
    kw = dict(num_labels=5, input_dim=24, hidden_dim=8, num_samples=V, learning_rate=0.5)
    oracle = co.OracleEngine(k, src, dst, part, feats, labels, co.GnnParam(**kw), seed=7)
    eng = Engine(k, src, dst, part, GnnParam(**kw), seed=7)
    x = cdist.attach_rccl(eng, 0, tcp=("127.0.0.1", 0))
    eng.set_global_data(feats, labels)
    eng.start()
    for it in range(6):
        oracle.iteration(it)
        eng.run(it, it + 1)
    for P in range(k):
        a, b = oracle.shares(P)
        assert np.array_equal(eng.shares(P, 0), a) and np.array_equal(eng.shares(P, 1), b)
    assert x.stats()["rounds"] == 0
    eng.close()
    x.close()
    lib = _lib()
    ident = (ctypes.c_uint8 * 128)()
    assert lib.cognn_rccl_rendezvous_tcp(b"not-an-address", 0, 0, 1, 1.0, ident) != 0
    assert lib.cognn_rccl_rendezvous_tcp(b"127.0.0.1", 1, 1, 2, 0.3, ident) != 0        # nobody listens: times out with a message
    assert b"not reachable" in lib.cognn_exchange_last_error()
