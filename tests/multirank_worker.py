"""Worker of tests/test_multirank_cpu.py: one rank of a world_size>1 engine run over gloo, with the engine's
host code linked against the plain-C++ reference backend (oracle/libcognn_engine_cpu.so — test infrastructure)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    cfg = json.loads(sys.argv[1])
    import torch
    import torch.distributed as dist
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cognn_oracle as co
    from cognn_amd import capi
    hip = cfg.get("backend") == "hip"                      # tests/test_multirank_gpu.py: ranks share cuda:0, gloo moves host copies
    if not hip:
        capi.LIB_PATH = os.path.join(ROOT, "oracle", "libcognn_engine_cpu.so")          # test infrastructure: the plain-C++ reference backend
        capi.load()
    from cognn_amd import dist as cdist
    from cognn_amd.engine import Engine, GnnParam
    k = cfg["k"]; V = cfg["V"]
    src, dst = co.synth_graph(V, cfg["Eu"], cfg["gseed"])
    part = np.array([v % k for v in range(V)], dtype=np.int32)
    feats, labels = co.synth_features(V, cfg["in"], cfg["lab"], cfg["gseed"] + 1, density=cfg.get("density", 0.2))
    gp = GnnParam(**dict(dict(num_labels=cfg["lab"], input_dim=cfg["in"], hidden_dim=cfg["hid"], num_samples=V, learning_rate=0.5), **cfg.get("param", {})))
    placement = cfg.get("placement", "party")              # "vertex-set": a rank holds both shares of its parties' vertex sets
    if "placement_by_rank" in cfg:                         # a deliberately inconsistent launch (test_mismatched_configuration_fails_fast)
        placement = cfg["placement_by_rank"][rank]
    eng = Engine(k, src, dst, part, gp, seed=cfg["seed"], variant=cfg["variant"], rank=rank, world=world, stream=0, placement=placement)
    # asynchronous exchange (begin / wait) unless the case asks for the blocking callback
    mk = cdist.make_exchange if cfg.get("blocking_exchange") else cdist.make_exchange_async
    hostile_stats = None
    if cfg.get("hostile"):                                 # late, poisoned, shuffled delivery (tests/hostile_transport.py; CPU backend only)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hostile_transport
        fns, hostile_stats = hostile_transport.make_hostile_exchange(cfg["hostile"], skip_waits=cfg.get("hostile_skip_waits", 0),
                                                                     per_round=bool(cfg.get("per_round", cfg.get("chunks", 1) > 1)))
        eng.set_exchange(fns)
    elif cfg.get("chunks", 1) > 1 and not cfg.get("blocking_exchange") and cfg.get("per_round", True):
        eng.set_exchange(mk(torch.device("cuda", 0), host_staged=True, per_round=True) if hip else mk(torch.device("cpu"), per_round=True))
    else:
        eng.set_exchange(mk(torch.device("cuda", 0), host_staged=True) if hip else mk(torch.device("cpu")))
    if cfg.get("chunks", 1) > 1:                           # open -> exchange -> close of the cross-rank sides in row chunks
        eng.exchange_chunks(cfg["chunks"])
    eng.set_global_data(feats, labels)
    if "placement_by_rank" in cfg:
        try:
            eng.start()
        except capi.CognnError as ex:
            print("START FAILED: %s" % ex, flush=True)
            eng.close()
            dist.destroy_process_group()
            sys.exit(7)
    else:
        eng.start()
    if cfg.get("exchanged_openings"):                      # every opening travels as two shares (COGNN_OPT_PUBLIC_OPENINGS off)
        eng.public_openings(False)
    if "pair_fusion" in cfg:
        eng.pair_fusion(bool(cfg["pair_fusion"]))
    if cfg.get("packed_openings"):                         # opened truncation shares as 6 bytes on the wire
        eng.packed_openings(True)
    out = {}
    m = k // world
    step = 6 if cfg.get("whole_epochs") else 1             # whole epochs per call: nothing is read between their GAS iterations
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
    if hostile_stats is not None:
        out["hostile_rounds"] = np.array([hostile_stats["rounds"], hostile_stats["max_inflight"], hostile_stats.get("left_inflight", 0)])
    np.savez(cfg["out"] + ".rank%d.npz" % rank, **out)
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
