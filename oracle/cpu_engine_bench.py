#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only): times the engine's host code linked against the plain-C++
reference backend (oracle/libcognn_engine_cpu.so, OpenMP over the independent loop iterations) on a synthetic workload and
prints one JSON line.  Usage: cpu_engine_bench.py K LOG2_V LOG2_E IN HID LAB VARIANT ITERS STEPS
(LOG2_V / LOG2_E written as =N give the exact vertex / directed-edge count instead of a power of two)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402


def main():
    k, in_dim, hid, lab = int(sys.argv[1]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    size = lambda a: int(a[1:]) if a.startswith("=") else 1 << int(a)
    variant, iters, steps = sys.argv[7], int(sys.argv[8]), int(sys.argv[9])
    import cognn_oracle as co
    from cognn_amd import capi
    capi.LIB_PATH = os.path.join(ROOT, "oracle", "libcognn_engine_cpu.so")   # test infrastructure: the plain-C++ reference backend
    capi.load()
    from cognn_amd.engine import Engine, GnnParam
    V, Eu = size(sys.argv[2]), size(sys.argv[3]) // 2
    src, dst = co.synth_graph(V, Eu, 0xC06A11)
    part = (np.arange(V) % k).astype(np.int32)
    gp = GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, num_edges=len(src))
    eng = Engine(k, src, dst, part, gp, seed=0xC06A11, variant=variant, stream=0)
    for P in eng.hosted:
        vids = eng.party_vids(P)
        rng = np.random.default_rng(0xC06A12 + P)
        eng.set_party_data(P, (rng.random((len(vids), in_dim)) < 0.01).astype(np.float64), rng.integers(0, lab, size=len(vids)))
    eng.start()
    # the plain-C++ backend's pair chain is a reference composition of the per-side entry points (extra copies, no OpenMP): the CPU
    # baseline runs the per-side loops themselves
    eng.pair_fusion(os.environ.get("COGNN_CPU_BENCH_PAIR_FUSION", "0") == "1")
    eng.retain_offline(True)                              # the passes replay the same iterations, like bench.py's GPU steps: the
    eng.offline(0, iters)                                 # dealer phase stays outside the timed region
    eng.run(0, iters)                                     # warm-up pass
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        eng.run(0, iters)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    eng.close()
    print(json.dumps({"seconds_per_pass": dt, "all_passes": times, "edges": int(len(src)), "threads": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count()))}))


if __name__ == "__main__":
    main()
