"""One rank of a multi-process run: the Python counterpart of harness.cpp's main() for one-party-per-GPU (or several
parties per GPU) runs.  Reads the reference's file formats, hosts the parties of this rank on the engine, exchanges shares
over torch.distributed (RCCL), and writes one log per hosted party with the reference's log lines
(`::iteration took X seconds`, accuracy block — tools/plot/plot_accuracy.py:17-24, plot_duration_breakdown_and_comm.py:99).

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python -m cognn_amd.worker -t <parties> -m <iters> -s <setting> \
        [--variant optimize-gcn|optimize-gcn-inference] <edge file> <vertex file> <partition file> <output file> <config file>
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
    ap.add_argument("--variant", default="optimize-gcn", choices=["optimize-gcn", "optimize-gcn-inference"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--lib", default=None, help="engine library (tests: oracle/libcognn_engine_cpu.so)")
    ap.add_argument("--log-dir", default=None)
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
    from . import capi
    capi.load(a.lib)
    from .engine import Engine, GnnParam
    edge, vertex, partf, _out, cfgf = a.files
    t_pre = time.perf_counter()
    gp = GnnParam.read_config(cfgf)
    src, dst = read_edge_list(edge)
    part = read_partition(partf)
    eng = Engine(a.t, src, dst, part, gp, seed=fnv1a(a.s), variant=a.variant, rank=rank, world=world,
                 device=local_rank if on_gpu else 0, stream=None if on_gpu else 0, undirected=a.u)
    if world > 1:
        from . import dist as cdist
        eng.set_exchange(cdist.make_exchange_async(torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")))
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
        logs[p] = open(os.path.join(a.log_dir, "gcn_test_%d.log" % p), "w") if a.log_dir else sys.stdout
        print("%d Initialize graph algo kernel" % p, file=logs[p])
        print("::preprocess took %f seconds" % (time.perf_counter() - t_pre), file=logs[p])
    if not a.n:
        t0 = time.perf_counter()
        eng.offline(0, a.m)
        for p in eng.hosted:
            print("::preprocess_OM took %f seconds" % (time.perf_counter() - t0), file=logs[p])
    epoch = 3 * gp.num_layers
    for it in range(a.m):
        t0 = time.perf_counter()
        eng.run(it, it + 1)
        if on_gpu:
            torch.cuda.synchronize()
        for p in eng.hosted:
            f = logs[p]
            print("tid-> %d, iteration-> %d" % (p, it), file=f)
            if it % epoch == gp.num_layers - 1:
                m = eng.metrics(p)
                print("--------", file=f)
                print("cross-entropy-loss = %f" % m["loss"], file=f)
                print("full set accuracy = %f" % m["full"], file=f)
                print("training set accuracy = %f" % m["train"], file=f)
                print("border training set accuracy = %f" % m["border_train"], file=f)
                print("test set accuracy = %f" % m["test"], file=f)
                print("border test set accuracy = %f" % m["border_test"], file=f)
                print("the number of vertices is %d, the number of border vertices is %d" % (int(m["n"]), int(m["n_border"])), file=f)
            print("::iteration took %f seconds" % (time.perf_counter() - t0), file=f)
    for p in eng.hosted:
        print("%d Finish algo kernel" % p, file=logs[p])
        if logs[p] is not sys.stdout:
            logs[p].close()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
