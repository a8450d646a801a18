"""Worker of tests/test_edge_cases_*.py: runs one degenerate-input case through the engine (CPU reference backend or HIP)
and compares every party's shares with the oracle after every GAS iteration.  Exit code 0 = bit-exact."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def cases():
    import cognn_oracle as co
    z = np.zeros(0, dtype=np.int64)
    s, d = co.synth_graph(12, 20, 2)
    hub_s = np.array([0] * 9 + list(range(1, 10)), dtype=np.int64)
    hub_d = np.array(list(range(1, 10)) + [0] * 9, dtype=np.int64)
    s5, d5 = co.synth_graph(40, 90, 4)
    return {
        # name: (k, V, src, dst, part, variant, iters)
        "no-edges": (2, 10, z, z, [v % 2 for v in range(10)], "optimize-gcn", 6),
        "empty-party": (3, 12, s, d, [v % 2 for v in range(12)], "optimize-gcn", 6),           # party 2 owns no vertex
        "one-vertex-each": (2, 2, np.array([0, 1]), np.array([1, 0]), [0, 1], "optimize-gcn", 6),
        "star": (2, 10, hub_s, hub_d, [v % 2 for v in range(10)], "optimize-gcn", 6),
        "duplicate-edges": (2, 6, np.array([0, 0, 1, 1, 2, 2, 3]), np.array([1, 1, 0, 0, 3, 3, 2]), [v % 2 for v in range(6)],
                            "optimize-gcn", 6),
        "all-in-one-party": (2, 9, np.array([0, 1, 2, 3]), np.array([1, 2, 3, 4]), [0] * 9, "optimize-gcn-inference", 2),
        "skewed-partition": (4, 40, s5, d5, [0 if v < 30 else 1 + (v % 3) for v in range(40)], "optimize-gcn", 6),
    }


def main():
    name, backend = sys.argv[1], sys.argv[2]
    import cognn_oracle as co
    from cognn_amd import capi
    if backend == "cpu":
        capi.LIB_PATH = os.path.join(ROOT, "oracle", "libcognn_engine_cpu.so")          # test infrastructure: the plain-C++ reference backend
        capi.load()
    from cognn_amd.engine import Engine, GnnParam
    k, V, src, dst, part, variant, iters = cases()[name]
    in_dim, hid, lab = 6, 4, 3
    feats, labels = co.synth_features(V, in_dim, lab, 5, density=0.3)
    op = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    oracle = co.OracleEngine(k, src, dst, part, feats, labels, op, seed=3, variant=variant)
    gp = GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    kw = {"stream": 0} if backend == "cpu" else {}
    eng = Engine(k, np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64), np.asarray(part, dtype=np.int32), gp, seed=3,
                 variant=variant, **kw)
    eng.set_global_data(feats, labels)
    eng.start()
    for it in range(iters):
        oracle.iteration(it)
        eng.run(it, it + 1)
        for P in range(k):
            a, b = oracle.shares(P)
            if not (np.array_equal(eng.shares(P, 0), a) and np.array_equal(eng.shares(P, 1), b)):
                print("MISMATCH case %s iteration %d owner %d" % (name, it, P))
                sys.exit(1)
    eng.close()


if __name__ == "__main__":
    main()
