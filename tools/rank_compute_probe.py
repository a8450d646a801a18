#!/usr/bin/env python3
"""Timing-only probe: the device time ONE rank of an N-rank run spends on its own kernels per pass, measured on a single GPU
with a transport that moves nothing (inboxes keep whatever they held: the shares are wrong on purpose), plus what that rank
would put on its links.  Together they bound what a multi-GPU placement can reach before the run exists.

  python tools/rank_compute_probe.py --world 8 [--chunks 4] [--workload config5] [--steps 10]     one layout
  python tools/rank_compute_probe.py --table [--chunks 4] [--link-gbs 77] [--round-us 20]          worlds 2, 4, 8: the table of DESIGN §7

Columns of the table (per pass, rank 0; the vid % k partition makes all ranks alike):
  kernels      device time of the rank's launches with the null transport, unchunked / with COGNN_OPT_EXCHANGE_CHUNKS = C
  gather+gemm  of that: aggregate + partial-sum launches and the Beaver product phase (engine timers); the rest is the element-wise
               open / close kernels, softmax and weight-sized work - what chunking can move under the link
  rounds       exchange rounds per pass, unchunked / chunked
  busiest      bytes sent to the peer rank that receives most (one direction of one link)
  MODEL        t_link = busiest / link rate + rounds x per-round latency;
               unchunked  = kernels + t_link          (an opening has to arrive before its close can run: dependent steps add up;
                                                        what the interior-first ordering and the split aggregate already overlap is
                                                        not credited - an upper bound)
               chunked    = kernels_C + t_link_C - min(element-wise_C, t_link_C) x (1 - 1/C)
                                                       (chunk c's round travels behind chunk c+1's kernels: all but one chunk of the
                                                        shorter of the two hides)
The model is arithmetic on measured kernel times and counted bytes, NOT a measurement of a multi-GPU run."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402


def probe(world, rank, workload, steps, chunks, graph=None, placement="party", packed=False):
    import torch
    import bench
    from cognn_amd.engine import Engine, GnnParam
    from cognn_amd.engine_api import EXCHANGE_FN, EXCHANGE_WAIT_FN, EXCHANGE_WAIT_ROUND_FN
    k, lv, le, in_dim, hid, lab, variant, iters = bench.WORKLOADS[workload]
    V, Eu = 1 << lv, 1 << (le - 1)
    src, dst = graph if graph is not None else bench.synth_graph(V, Eu, 0xC06A11)
    part = (np.arange(V) % k).astype(np.int32)
    gp = GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, num_edges=len(src))
    eng = Engine(k, src, dst, part, gp, seed=0xC06A11, variant=variant, rank=rank, world=world, device=0, placement=placement)
    sent = {}
    rounds = [0]

    def begin(user, xfers, n):
        for i in range(n):
            if xfers[i].is_send:
                sent[xfers[i].peer] = sent.get(xfers[i].peer, 0) + xfers[i].bytes
        rounds[0] += 1
        return 0

    fns = (EXCHANGE_FN(begin), EXCHANGE_WAIT_FN(lambda user: 0), EXCHANGE_WAIT_ROUND_FN(lambda user, r: 0))
    eng.set_exchange(fns)
    for P in eng.hosted:
        vids = eng.party_vids(P)
        rng = np.random.default_rng(0xC06A12 + P)
        eng.set_party_data(P, (rng.random((len(vids), in_dim)) < 0.01).astype(np.float64), rng.integers(0, lab, size=len(vids)))
    eng.start()
    eng.retain_offline(True)
    if "inference" in variant:
        eng.forward_only(True)
    if chunks > 1:
        eng.exchange_chunks(chunks)
    if packed:
        eng.packed_openings(True)                             # opened truncation / ReLU-product shares as 6 bytes per element
    eng.offline(0, iters)
    eng.run(0, iters)
    torch.cuda.synchronize()
    sent.clear(); rounds[0] = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run(0, iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    nrounds = rounds[0] // steps
    per_peer = {p: b / steps for p, b in sent.items()}
    eng.enable_timing(True)                                   # a second set of passes under the engine's event timers
    for _ in range(steps):
        eng.run(0, iters)
    torch.cuda.synchronize()
    heavy = sum(eng.timing(kind)[1] for kind in (0, 1, 2)) / steps      # aggregate, partial sums, Beaver product phase (ms)
    eng.close()
    return {"kernels_ms": dt * 1e3, "heavy_ms": heavy, "rounds": nrounds, "busiest_mb": max(per_peer.values()) / 1e6 if per_peer else 0.0,
            "sent_mb": sum(per_peer.values()) / 1e6, "graph": (src, dst)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--workload", default="config5")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--table", action="store_true")
    ap.add_argument("--packed", action="store_true", help="COGNN_OPT_PACKED_OPENINGS: one layout with it; --table adds the packed columns anyway")
    ap.add_argument("--placement", default="party", choices=["party", "vertex-set"], help="which rank holds which share (cognn_engine_config::placement)")
    ap.add_argument("--link-gbs", type=float, default=77.0, help="xGMI rate per link and direction assumed by the model")
    ap.add_argument("--round-us", type=float, default=20.0, help="fixed cost per exchange round assumed by the model")
    a = ap.parse_args()
    if not a.table:
        r = probe(a.world, a.rank, a.workload, a.steps, a.chunks, placement=a.placement, packed=a.packed)
        print("placement %s: " % a.placement, end="")
        print("rank %d of %d, %s, chunks %d: %.3f ms of kernels per pass (null transport; %.3f ms of it aggregate + partial sums + products), "
              "%d rounds, %.1f MB sent per pass, %.1f MB to the busiest peer"
              % (a.rank, a.world, a.workload, a.chunks, r["kernels_ms"], r["heavy_ms"], r["rounds"], r["sent_mb"], r["busiest_mb"]))
        return
    C = max(a.chunks, 2)
    graph = None
    print("| world | kernels ms (C=1 / C=%d) | gather+gemm ms | rounds (C=1 / C=%d) | busiest link MB | t_link ms (C=1 / C=%d) | MODEL pass ms unchunked | MODEL pass ms chunked "
          "| packed: kernels ms | packed: busiest link MB | packed: MODEL pass ms |" % (C, C, C))
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for world in (2, 4, 8):
        r1 = probe(world, 0, a.workload, a.steps, 1, graph, a.placement)
        graph = r1["graph"]
        rc = probe(world, 0, a.workload, a.steps, C, graph, a.placement)
        link = lambda r: r["busiest_mb"] * 1e6 / (a.link_gbs * 1e9) * 1e3 + r["rounds"] * a.round_us * 1e-3
        t1, tc = link(r1), link(rc)
        ew_c = max(rc["kernels_ms"] - rc["heavy_ms"], 0.0)
        unchunked = r1["kernels_ms"] + t1
        chunked = rc["kernels_ms"] + tc - min(ew_c, tc) * (1.0 - 1.0 / C)
        rp = probe(world, 0, a.workload, a.steps, 1, graph, a.placement, packed=True)      # 6-byte openings (pack / unpack launches included)
        print("| %d | %.2f / %.2f | %.2f | %d / %d | %.0f | %.2f / %.2f | %.2f | %.2f | %.2f | %.0f | %.2f |"
              % (world, r1["kernels_ms"], rc["kernels_ms"], r1["heavy_ms"], r1["rounds"], rc["rounds"], r1["busiest_mb"], t1, tc, unchunked, chunked,
                 rp["kernels_ms"], rp["busiest_mb"], rp["kernels_ms"] + link(rp)))
    print("(placement %s; model: %.0f GB/s per link and direction, %.0f us per round; kernels measured on one MI355X with a transport that moves nothing)"
          % (a.placement, a.link_gbs, a.round_us))


if __name__ == "__main__":
    main()
