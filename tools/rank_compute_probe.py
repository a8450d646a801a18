#!/usr/bin/env python3
"""Timing-only probe: the device time ONE rank of an N-rank run spends on its own kernels per pass, measured on a single GPU
with a transport that moves nothing (inboxes keep whatever they held: the shares are wrong on purpose).  Together with the bytes
per link and pass this bounds what a multi-GPU placement can reach before the run exists.
usage: python tools/rank_compute_probe.py --world 8 [--workload config5] [--steps 10]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--workload", default="config5")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    import torch
    import bench
    from cognn_amd.engine import Engine, GnnParam
    from cognn_amd.engine_api import EXCHANGE_FN, EXCHANGE_WAIT_FN
    k, lv, le, in_dim, hid, lab, variant, iters = bench.WORKLOADS[a.workload]
    V, Eu = 1 << lv, 1 << (le - 1)
    src, dst = bench.synth_graph(V, Eu, 0xC06A11)
    part = (np.arange(V) % k).astype(np.int32)
    gp = GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, num_edges=len(src))
    eng = Engine(k, src, dst, part, gp, seed=0xC06A11, variant=variant, rank=a.rank, world=a.world, device=0)
    sent = [0, 0]

    def begin(user, xfers, n):
        for i in range(n):
            if xfers[i].is_send:
                sent[0] += xfers[i].bytes
        sent[1] += 1
        return 0

    fns = (EXCHANGE_FN(begin), EXCHANGE_WAIT_FN(lambda user: 0))
    eng.set_exchange(fns)
    for P in eng.hosted:
        vids = eng.party_vids(P)
        rng = np.random.default_rng(0xC06A12 + P)
        eng.set_party_data(P, (rng.random((len(vids), in_dim)) < 0.01).astype(np.float64), rng.integers(0, lab, size=len(vids)))
    eng.start()
    eng.retain_offline(True)
    if "inference" in variant:
        eng.forward_only(True)
    eng.offline(0, iters)
    eng.run(0, iters)
    torch.cuda.synchronize()
    sent[0] = sent[1] = 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.run(0, iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("rank %d of %d, %s: %.3f ms of kernels per pass (null transport), %d rounds and %.1f MB sent per pass"
          % (a.rank, a.world, a.workload, dt * 1e3, sent[1] // a.steps, sent[0] / a.steps / 1e6))
    eng.close()


if __name__ == "__main__":
    main()
