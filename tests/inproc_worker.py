"""Worker of the in-process multi-rank tests: `world` engine ranks as THREADS of one process, each with its own engine (rank r of
`world`) and an in-process mailbox transport.  A GPU box admits only a handful of processes on its card, so this is how the
north-star layout - 8 parties, one per rank, world 8 (the reference's k x k mesh, include/engine.h:157-201) - runs the real HIP
kernels on one GPU; it also puts eight engines / contexts side by side in one address space (nothing process-global may differ
between them: dealer keys carry the epoch salt, errors are thread-local, libc's rand() - the Glorot initialisation - runs under a lock;
that last one was found by this worker: two engines interleaving srand(42) / rand() started from different weights).  Output format = tests/multirank_worker.py's, one file
per rank."""
import ctypes
import json
import os
import queue
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def make_mailbox_exchange(rank, boxes, device, wrap, stats):
    """(begin, wait, wait_round): a send is copied out of the engine's buffer at begin() and posted to the (rank -> peer) queue;
    receives are filled at wait() in issue order - messages between two ranks match by order, as RCCL matches them."""
    import torch
    from cognn_amd.engine_api import EXCHANGE_FN, EXCHANGE_WAIT_FN, EXCHANGE_WAIT_ROUND_FN
    pending = []                                            # rounds begun and not completed: (round number, [(ptr, bytes, peer)])
    count = [0]

    def _begin(user, xfers, n):
        try:
            recvs = []
            for i in range(n):
                x = xfers[i]
                if x.is_send:
                    # snapshot at begin(); on the GPU the message stays on the device (a device-to-device copy on this rank's
                    # stream, complete before it is posted - what a peer-to-peer transport does)
                    snap = wrap(x.ptr, x.bytes, device).clone()
                    if device.type == "cuda":
                        torch.cuda.current_stream().synchronize()
                    boxes[(rank, int(x.peer))].put(snap)
                    stats["bytes"] += int(x.bytes)
                else:
                    recvs.append((int(x.ptr), int(x.bytes), int(x.peer)))
            pending.append((count[0], recvs))
            count[0] += 1
            stats["rounds"] += 1
            return 0
        except Exception as ex:  # noqa: BLE001
            print("mailbox exchange (begin) failed on rank %d: %r" % (rank, ex), flush=True)
            return 1

    def _complete(upto):
        try:
            while pending and pending[0][0] <= upto:
                _, recvs = pending.pop(0)
                for ptr, nbytes, peer in recvs:
                    data = boxes[(peer, rank)].get(timeout=300)
                    assert data.numel() == nbytes, (rank, peer, data.numel(), nbytes)
                    wrap(ptr, nbytes, device).copy_(data)
                    if device.type == "cuda":
                        # the snapshot was allocated on the SENDER's stream: without this the caching allocator hands its block back to
                        # that stream's pool as soon as the reference below is dropped - while this rank's copy may still be queued behind
                        # its own kernels (a round in which a rank only receives does not drain its stream first) - and the sender's
                        # next snapshot overwrote it: one wrong weight share in ~7 % of the runs of a configuration a soak run found
                        data.record_stream(torch.cuda.current_stream())
            if device.type == "cuda":
                torch.cuda.current_stream().synchronize()
            return 0
        except Exception as ex:  # noqa: BLE001
            print("mailbox exchange (wait) failed on rank %d: %r" % (rank, ex), flush=True)
            return 1

    return (EXCHANGE_FN(_begin), EXCHANGE_WAIT_FN(lambda user: _complete(1 << 62)),
            EXCHANGE_WAIT_ROUND_FN(lambda user, rnd: _complete(int(rnd))))


def main():
    cfg = json.loads(sys.argv[1])
    import torch
    import cognn_oracle as co
    from cognn_amd import capi
    hip = cfg.get("backend") == "hip"
    if not hip:
        capi.LIB_PATH = os.path.join(ROOT, "oracle", "libcognn_engine_cpu.so")          # test infrastructure: the plain-C++ reference backend
    capi.load()
    from cognn_amd import dist as cdist
    from cognn_amd.engine import Engine, GnnParam
    k, V, world = cfg["k"], cfg["V"], cfg["world"]
    src, dst = co.synth_graph(V, cfg["Eu"], cfg["gseed"])
    part = np.array([v % k for v in range(V)], dtype=np.int32)
    feats, labels = co.synth_features(V, cfg["in"], cfg["lab"], cfg["gseed"] + 1, density=cfg.get("density", 0.2))
    gp = GnnParam(**dict(dict(num_labels=cfg["lab"], input_dim=cfg["in"], hidden_dim=cfg["hid"], num_samples=V, learning_rate=0.5), **cfg.get("param", {})))
    placement = cfg.get("placement", "party")
    device = torch.device("cuda", 0) if hip else torch.device("cpu")
    boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world) if a != b}
    m = k // world
    step = 6 if cfg.get("whole_epochs") else 1
    errors = []
    stats = [{"rounds": 0, "bytes": 0} for _ in range(world)]

    def rank_main(rank):
        try:
            stream = 0
            if hip:
                torch.cuda.set_device(0)
                if cfg.get("private_streams", True):        # one stream per rank (as every rank of a real run has): the engine's launches and
                    ts = torch.cuda.Stream(device=0)        # the transport's copies of this thread are ordered on it, not on the shared null stream
                    torch.cuda.set_stream(ts)
                    stream = ts.cuda_stream
            eng = Engine(k, src, dst, part, gp, seed=cfg["seed"], variant=cfg["variant"], rank=rank, world=world, stream=stream, placement=placement)
            eng.set_exchange(make_mailbox_exchange(rank, boxes, device, cdist._wrap, stats[rank]))
            if cfg.get("chunks", 1) > 1:
                eng.exchange_chunks(cfg["chunks"])
            eng.set_global_data(feats, labels)
            eng.start()
            if "pair_fusion" in cfg:
                eng.pair_fusion(bool(cfg["pair_fusion"]))
            if cfg.get("exchanged_openings"):
                eng.public_openings(False)
            if cfg.get("packed_openings"):
                eng.packed_openings(True)
            out = {}
            for it0 in range(0, cfg["iters"], step):
                eng.run(it0, it0 + step)
                it = it0 + step - 1
                for o in range(k):
                    if o // m == rank:
                        out["it%d_o%d_s0" % (it, o)] = eng.shares(o, 0)
                        for l in range(2):
                            out["it%d_o%d_s0_w%d" % (it, o, l)] = eng.weight(o, 0, l)
                    if (o if placement == "vertex-set" else (o + 1) % k) // m == rank:
                        out["it%d_o%d_s1" % (it, o)] = eng.shares(o, 1)
                        for l in range(2):
                            out["it%d_o%d_s1_w%d" % (it, o, l)] = eng.weight(o, 1, l)
            out["exchange_stats"] = np.array([stats[rank]["rounds"], stats[rank]["bytes"]])
            np.savez(cfg["out"] + ".rank%d.npz" % rank, **out)
            eng.close()
        except BaseException as ex:  # noqa: BLE001 - reported by the main thread
            errors.append((rank, repr(ex)))
            print("rank %d failed: %r" % (rank, ex), flush=True)
            os._exit(3)                                     # (the peers would otherwise sit in their queues until they time out)

    threads = [threading.Thread(target=rank_main, args=(r,), name="rank%d" % r) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        print("errors: %r" % (errors,), flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()
