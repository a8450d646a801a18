"""One rank of a multi-process run driven from Python: the counterpart of harness.cpp's main() for hosts that already
live in a torch.distributed job (bin/gcn-optimize -c 1 is the C++ equivalent without Python).  Reads the reference's file
formats, hosts the parties of this rank on the engine and writes one log per hosted party with the reference's log lines
(`::<tag> took X seconds`, accuracy block — tools/plot/plot_accuracy.py:17-24, plot_duration_breakdown_and_comm.py:23-46,99).
Shares travel over the library's native RCCL transport (backend "nccl"); backend "gloo" uses the torch.distributed test
transport of cognn_amd/dist.py.

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python -m cognn_amd.worker -t <parties> -m <iters> -s <setting> \
        [--variant optimize-gcn|optimize-gcn-inference|original-gcn] <edge file> <vertex file> <partition file> <output file> <config file>
"""
import argparse
import os
import sys
import time

import numpy as np


def read_edge_list(path):
    """`src dst [weight]` per line, '#' comments and blank lines skipped (graph_io_util.h:17-22,121-147)."""
    src, dst = [], []
    with open(path) as f:
        for line in f:
            if not line.strip() or line[0] == "#":
                continue
            t = line.split()
            src.append(int(t[0])); dst.append(int(t[1]))
    return np.array(src, dtype=np.int64), np.array(dst, dtype=np.int64)


def read_partition(path):
    """`vid tid` per line (graph_io_util.h:67-73); vids must be dense."""
    rows = []
    with open(path) as f:
        for line in f:
            if not line.strip() or line[0] == "#":
                continue
            t = line.split()
            rows.append((int(t[0]), int(t[1])))
    part = np.full(len(rows), -1, dtype=np.int32)
    for v, t in rows:
        if v >= len(rows) or part[v] != -1:
            raise ValueError("partition file: vertex ids must be dense and unique")
        part[v] = t
    return part


def read_vertex_rows(path, wanted, input_dim):
    """`vid f_0 ... f_{in-1} label` (harness.cpp:21-48, kernel_harness.h:37-44); returns {vid: (features, label)} for wanted vids."""
    out = {}
    with open(path) as f:
        for line in f:
            if not line.strip() or line[0] == "#":
                continue
            t = line.split()
            vid = int(t[0])
            if vid in wanted:
                out[vid] = (np.array(t[1:1 + input_dim], dtype=np.float64), int(t[1 + input_dim]))
    return out


def fnv1a(s):
    h = 0xcbf29ce484222325
    for c in s.encode():
        h = ((h ^ c) * 0x100000001b3) & ((1 << 64) - 1)
    return h


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-t", type=int, required=True, help="number of parties")
    ap.add_argument("-g", type=int, default=0)
    ap.add_argument("-m", type=int, default=1000)
    ap.add_argument("-p", type=int, default=16)
    ap.add_argument("-s", default="")
    ap.add_argument("-n", type=int, default=0)
    ap.add_argument("-c", type=int, default=0)
    ap.add_argument("-r", type=int, default=1)
    ap.add_argument("-u", action="store_true")
    ap.add_argument("--variant", default="optimize-gcn", choices=["optimize-gcn", "optimize-gcn-inference", "original-gcn"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--log-dir", default=None)
    ap.add_argument("--log-prefix", default="gcn_test_", help="log file name = <prefix><party>.log")
    ap.add_argument("files", nargs=5, help="edge list, vertex list, partition, output, config")
    a = ap.parse_args(argv)
    if a.r != 1:
        raise SystemExit("Only the no-dummy-edge mode (-r 1) is supported.")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    on_gpu = a.backend == "nccl"
    if on_gpu:
        torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(a.backend, rank=rank, world_size=world,
                                **({"device_id": torch.device("cuda", local_rank)} if on_gpu else {}))
    from .engine import Engine, GnnParam
    edge, vertex, partf, _out, cfgf = a.files
    t_pre = time.perf_counter()
    gp = GnnParam.read_config(cfgf)
    src, dst = read_edge_list(edge)
    part = read_partition(partf)
    if a.g and a.g != a.t:
        if a.g % a.t:
            raise SystemExit("Number of threads must be a divisor of number of graph tiles.")
        part = part // (a.g // a.t)                                # tileMergeFactor (graph_io_util.h:76)
    eng = Engine(a.t, src, dst, part, gp, seed=fnv1a(a.s), variant=a.variant, rank=rank, world=world,
                 device=local_rank if on_gpu else 0, stream=None if on_gpu else 0, undirected=a.u, verbose=True)
    xch = None
    if world > 1:
        from . import dist as cdist
        if on_gpu:
            xch = cdist.attach_rccl(eng, local_rank)
        else:
            eng.set_exchange(cdist.make_exchange_async(torch.device("cpu")))
    vids = {p: eng.party_vids(p) for p in eng.hosted}
    wanted = set(int(v) for p in eng.hosted for v in vids[p])
    rows = read_vertex_rows(vertex, wanted, gp.input_dim)
    for p in eng.hosted:
        feats = np.stack([rows[int(v)][0] for v in vids[p]]) if len(vids[p]) else np.zeros((0, gp.input_dim))
        labels = np.array([rows[int(v)][1] for v in vids[p]], dtype=np.int32)
        eng.set_party_data(p, feats, labels)
    eng.start()
    logs = {}
    for p in eng.hosted:
        logs[p] = open(os.path.join(a.log_dir, "%s%d.log" % (a.log_prefix, p)), "w") if a.log_dir else sys.stdout
        print("%d Initialize graph algo kernel" % p, file=logs[p])
        print("::preprocess took %f seconds" % (time.perf_counter() - t_pre), file=logs[p])
    # offline phase one epoch ahead with the on-disk cache keyed by -s, like bin/gcn-optimize (harness_main.cpp)
    epoch = (2 if a.variant == "original-gcn" else 3) * gp.num_layers     # getEpochLayerNum
    cache_dir = os.path.join("preprocess", a.s.replace(" ", "_"))
    use_cache = not os.environ.get("COGNN_NO_PREPROCESS_CACHE")
    if not a.n and use_cache:
        os.makedirs(cache_dir, exist_ok=True)
    reused = 0
    for it in range(a.m):
        if it % epoch == 0:
            t0 = time.perf_counter()
            it1 = min(it + epoch, a.m)
            if not a.n:
                eng.offline(it, it1)
                if use_cache:
                    eng.offline_save(cache_dir)
            else:
                reused += eng.offline_load(cache_dir, it, it1)
            eng.sync()
            if it == 0:
                for p in eng.hosted:
                    if not a.n:
                        print("::preprocess_OM took %f seconds" % (time.perf_counter() - t0), file=logs[p])
                    else:
                        print("%d Reused %d offline products from %s" % (p, reused, cache_dir), file=logs[p])
        t0 = time.perf_counter()
        eng.run(it, it + 1)
        eng.sync()
        ph = eng.phase_seconds()
        e = it % epoch
        apply_only = e != 0 and e % gp.num_layers == 0             # ss_...h:709
        for p in eng.hosted:
            f = logs[p]
            print("tid-> %d, iteration-> %d" % (p, it), file=f)
            if not apply_only:                                     # the tags of the reference's print_duration sites; see harness_main.cpp
                for tag, sec in (("PreScatterComp Client", ph["prescatter"]), ("PreScatterComp Server", ph["prescatter"]),
                                 ("Scatter_preparation", 0.0), ("Scatter_computation", 0.0), ("premerging", ph["message_passing"]),
                                 ("premerged_extraction", 0.0), ("Gather_preparation", 0.0), ("Gather_computation", ph["gather_scale"])):
                    print("::%s took %f seconds" % (tag, sec), file=f)
            if e == gp.num_layers - 1:
                m = eng.metrics(p)
                print("--------", file=f)
                print("cross-entropy-loss = %f" % m["loss"], file=f)
                print("full set accuracy = %f" % (100.0 * m["full"]), file=f)   # per cent, README.md:226-236
                print("training set accuracy = %f" % (100.0 * m["train"]), file=f)
                print("border training set accuracy = %f" % (100.0 * m["border_train"]), file=f)
                print("test set accuracy = %f" % (100.0 * m["test"]), file=f)
                print("border test set accuracy = %f" % (100.0 * m["border_test"]), file=f)
                print("the number of vertices is %d, the number of border vertices is %d" % (int(m["n"]), int(m["n_border"])), file=f)
            if not apply_only:
                print("::Apply_computation took %f seconds" % (ph["apply"] + ph["weight_average"]), file=f)
            print("::iteration took %f seconds" % (time.perf_counter() - t0), file=f)
    for p in eng.hosted:
        print("%d Finish algo kernel" % p, file=logs[p])
        if logs[p] is not sys.stdout:
            logs[p].close()
    if xch is not None:
        xch.barrier()
    eng.close()
    if xch is not None:
        xch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
