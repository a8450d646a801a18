#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the build container:
    python tests/golden/make_golden.py
* reference_configs.json  — the GNNParam values of /root/reference/build_from_source/config/*.txt
  (data files of the reference; only their key/value content is recorded).
* glorot_srand42.json     — first values of initWeight(dim0, dim1) (gcn.h:838-852) from this libc's
  srand(42)/rand().
* original_gcn_small_run.json — the same for oracle/original_gcn.py (config 1's variant), 2 parties, 3 epochs.
* oracle_small_run.json   — SHA-256 of every party's shares after each GAS iteration of a small seeded
  run of oracle/cognn_oracle.py.  SELF-GENERATED: it pins the oracle against regressions, it is not a
  reference vector (the reference has none; parity unpinned).
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import cognn_oracle as co  # noqa: E402


def configs():
    d = "/root/reference/build_from_source/config"
    out = {}
    for name in sorted(os.listdir(d)):
        toks = open(os.path.join(d, name)).read().split()
        out[name] = {toks[i]: toks[i + 2] for i in range(0, len(toks) - 2, 3)}
    return out


def glorot():
    out = {}
    for d0, d1 in ((1433, 16), (16, 7), (2, 3)):
        w = co.init_weight(d0, d1)
        out["%dx%d" % (d0, d1)] = {"first8": [repr(float(x)) for x in w.reshape(-1)[:8]],
                                   "sum": repr(float(w.sum()))}
    return out


def small_run():
    k, V, Eu = 3, 40, 90
    src, dst = co.synth_graph(V, Eu, 5)
    part = [v % k for v in range(V)]
    feats, labels = co.synth_features(V, 12, 4, 6, density=0.25)
    p = co.GnnParam(num_labels=4, input_dim=12, hidden_dim=6, num_samples=V, learning_rate=0.5)
    o = co.OracleEngine(k, src, dst, part, feats, labels, p, seed=0xC06A11)
    digests = []
    for it in range(12):
        o.iteration(it)
        h = hashlib.sha256()
        for P in range(k):
            a, b = o.shares(P)
            h.update(np.ascontiguousarray(a).tobytes()); h.update(np.ascontiguousarray(b).tobytes())
            for l in range(2):
                h.update(np.ascontiguousarray(o.states[P].localWeight[l]).tobytes())
                h.update(np.ascontiguousarray(o.states[P].remoteWeight[l]).tobytes())
        digests.append(h.hexdigest())
    return {"k": k, "V": V, "Eu": Eu, "graph_seed": 5, "feat_seed": 6, "seed": 0xC06A11, "in": 12, "hid": 6, "lab": 4,
            "digests": digests, "metrics": [{kk: (round(v, 12) if isinstance(v, float) else v) for kk, v in m.items()} for m in o.metrics]}


def _digest(o, k):
    h = hashlib.sha256()
    for P in range(k):
        a, b = o.shares(P)
        h.update(np.ascontiguousarray(a).tobytes()); h.update(np.ascontiguousarray(b).tobytes())
        for l in range(2):
            h.update(np.ascontiguousarray(o.states[P].localWeight[l]).tobytes())
            h.update(np.ascontiguousarray(o.states[P].remoteWeight[l]).tobytes())
    return h.hexdigest()


def original_small_run():
    import original_gcn as og
    k, V, Eu = 2, 40, 90
    src, dst = co.synth_graph(V, Eu, 5)
    part = [v % k for v in range(V)]
    feats, labels = co.synth_features(V, 12, 4, 6, density=0.25)
    p = co.GnnParam(num_labels=4, input_dim=12, hidden_dim=6, num_samples=V, learning_rate=0.5)
    o = og.OriginalOracleEngine(k, src, dst, part, feats, labels, p, seed=0xC06A11)
    digests = []
    for it in range(12):
        o.iteration(it)
        digests.append(_digest(o, k))
    return {"k": k, "V": V, "Eu": Eu, "graph_seed": 5, "feat_seed": 6, "seed": 0xC06A11, "in": 12, "hid": 6, "lab": 4,
            "digests": digests, "metrics": [{kk: (round(v, 12) if isinstance(v, float) else v) for kk, v in m.items()} for m in o.metrics]}


if __name__ == "__main__":
    if os.path.isdir("/root/reference"):
        json.dump(configs(), open(os.path.join(HERE, "reference_configs.json"), "w"), indent=1, sort_keys=True)
    json.dump(glorot(), open(os.path.join(HERE, "glorot_srand42.json"), "w"), indent=1, sort_keys=True)
    json.dump(small_run(), open(os.path.join(HERE, "oracle_small_run.json"), "w"), indent=1, sort_keys=True)
    json.dump(original_small_run(), open(os.path.join(HERE, "original_gcn_small_run.json"), "w"), indent=1, sort_keys=True)
    print("golden fixtures written")
