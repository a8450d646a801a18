"""Worker of tests/test_hostlogic_*.py: engine host logic that does not depend on the arithmetic backend - bounded device
memory over many epochs, the offline cache's header validation, replay retention - on the CPU reference backend ("cpu") or
the HIP library ("hip").  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def make(k, V, Eu, in_dim, hid, lab, seed=5, verbose=False):
    import cognn_oracle as co
    from cognn_amd.engine import Engine, GnnParam
    src, dst = co.synth_graph(V, Eu, 2)
    part = np.array([v % k for v in range(V)], dtype=np.int32)
    feats, labels = co.synth_features(V, in_dim, lab, 3, density=0.3)
    gp = GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    kw = {"stream": 0} if BACKEND == "cpu" else {}
    eng = Engine(k, src, dst, part, gp, seed=seed, verbose=verbose, **kw)
    eng.set_global_data(feats, labels)
    eng.start()
    return eng


def main():
    global BACKEND
    BACKEND, tmp = sys.argv[1], sys.argv[2]
    from cognn_amd import capi
    if BACKEND == "cpu":
        capi.LIB_PATH = os.path.join(ROOT, "oracle", "libcognn_engine_cpu.so")   # test infrastructure: the plain-C++ reference backend
    out = {}
    # 1. memory stays bounded: deal one epoch ahead, run it, repeat
    eng = make(3, 60, 150, 12, 6, 4, verbose=True)
    mem = []
    for ep in range(6):
        eng.offline(6 * ep, 6 * ep + 6)
        eng.run(6 * ep, 6 * ep + 6)
        eng.sync()
        mem.append(eng.memory())
    out["mem"] = mem
    out["phases"] = eng.phase_seconds()
    ref_shares = eng.shares(0, 0).tolist()
    # 2. offline cache: written by this run's next epoch, accepted by an identical engine, rejected by a different shape / seed
    eng.offline(36, 42)
    cache = os.path.join(tmp, "cache")
    os.makedirs(cache, exist_ok=True)
    eng.offline_save(cache)
    out["files"] = len(os.listdir(cache))
    eng.close()
    same = make(3, 60, 150, 12, 6, 4)
    out["loaded_same"] = same.offline_load(cache, 36, 42)
    same.close()
    for name, args in (("other_hidden", (3, 60, 150, 12, 8, 4)), ("other_graph", (3, 66, 150, 12, 6, 4)), ("other_parties", (2, 60, 150, 12, 6, 4))):
        e2 = make(*args)
        out["loaded_" + name] = e2.offline_load(cache, 36, 42)
        e2.run(0, 6)                                   # and it still runs (dealing on demand)
        e2.close()
    e3 = make(3, 60, 150, 12, 6, 4, seed=6)
    out["loaded_other_seed"] = e3.offline_load(cache, 36, 42)
    e3.close()
    # a truncated file is ignored as well
    victim = sorted(os.listdir(cache))[0]
    data = open(os.path.join(cache, victim), "rb").read()
    open(os.path.join(cache, victim), "wb").write(data[:-8])
    e4 = make(3, 60, 150, 12, 6, 4)
    out["loaded_truncated"] = e4.offline_load(cache, 36, 42)
    e4.close()
    # 3. replay with retention keeps the shares and reproduces the results; without it the second pass deals on demand (same results)
    res = {}
    for keep in (True, False):
        e5 = make(3, 60, 150, 12, 6, 4)
        e5.retain_offline(keep)
        e5.offline(0, 2)
        e5.run(0, 2); a = e5.shares(1, 1).copy(); m0 = e5.memory()
        e5.run(0, 2); b = e5.shares(1, 1).copy(); m1 = e5.memory()
        res[str(keep)] = {"same": bool(np.array_equal(a, b)), "grew": m1[0] - m0[0]}
        e5.close()
    out["replay"] = res
    # 3b. product shares dealt and handed back unconsumed: the next deal reuses their buffers, the run that follows deals on demand
    e7 = make(3, 60, 150, 12, 6, 4)
    e7.offline(0, 6); m0 = e7.memory()
    n0 = e7.offline_discard(0, 6)
    e7.offline(6, 12); m1 = e7.memory()
    n1 = e7.offline_discard(0, 6)                      # (nothing left of those iterations)
    e7.run(0, 6); got = e7.shares(1, 1).copy()
    e8 = make(3, 60, 150, 12, 6, 4)
    e8.run(0, 6)
    out["discard"] = {"first": n0, "again": n1, "grew": m1[0] - m0[0], "same": bool(np.array_equal(got, e8.shares(1, 1)))}
    e7.close(); e8.close()
    # 4. whole epochs in one call take the paths that only exist across GAS iterations (the deferred ReLU' selection, the backward
    #    PreScatter scale run ahead by the chain that truncates g): same shares and weights as one call per iteration, same memory
    #    from the second epoch on
    digests, mems = [], []
    for whole in (True, False):
        e6 = make(3, 60, 150, 12, 6, 4)
        for ep in range(3):
            if whole:
                e6.run(6 * ep, 6 * ep + 6)
            else:
                for it in range(6 * ep, 6 * ep + 6):
                    e6.run(it, it + 1)
            if whole:
                mems.append(e6.memory()[1])
        d = []
        for P in range(3):
            for sd in (0, 1):
                d.append(e6.shares(P, sd).tolist())
                d += [e6.weight(P, sd, l).tolist() for l in (0, 1)]
        digests.append(d)
        e6.close()
    out["epoch_calls_identical"] = digests[0] == digests[1]
    out["epoch_calls_memory"] = mems
    out["ref_rows"] = len(ref_shares)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
