"""bench.py's `small` workload - the bench configuration at 1/64 scale: 8 parties co-located on one GPU, 2^14 vertices /
2^18 directed edges, in=128 hid=64 labels=16, i.e. the grouped MFMA products, the fragment-ordered feature opening and the
Gather with the pair chain as its epilogue - against the numpy oracle, bit for bit on EVERY row: after GAS iterations 0 and 1
of a plain run, and after the bench's own sequence (forward-only stores, retained offline products, the pass replayed)."""
import os
import sys

import numpy as np
import pytest

import cognn_oracle as co

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _inputs(k, lv, le, in_dim, lab):
    import bench
    V, Eu = 1 << lv, 1 << (le - 1)
    src, dst = bench.synth_graph(V, Eu, 0xC06A11)
    part = (np.arange(V) % k).astype(np.int32)
    feats = np.zeros((V, in_dim)); labels = np.zeros(V, dtype=np.int64)
    for P in range(k):                                       # the generator of bench.py, per party
        vids = np.arange(P, V, k)
        rng = np.random.default_rng(0xC06A12 + P)
        feats[vids] = (rng.random((len(vids), in_dim)) < 0.01).astype(np.float64)
        labels[vids] = rng.integers(0, lab, size=len(vids))
    return V, src, dst, part, feats, labels


@pytest.mark.parametrize("hid", [64, 16])
def test_small_workload_matches_the_oracle_on_every_row(hid):
    import bench
    from cognn_amd.engine import Engine, GnnParam
    k, lv, le, in_dim, _, lab, variant, iters = bench.WORKLOADS["small"]
    V, src, dst, part, feats, labels = _inputs(k, lv, le, in_dim, lab)
    o = co.OracleEngine(k, src, dst, part, feats, labels, co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V),
                        seed=0xC06A11, variant=variant)
    gp = GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, num_edges=len(src))

    def engine():
        e = Engine(k, src, dst, part, gp, seed=0xC06A11, variant=variant)
        e.set_global_data(feats, labels)
        e.start()
        return e
    plain = engine()
    want = []
    for it in range(iters):
        plain.run(it, it + 1)
        o.iteration(it)
        want.append([o.shares(P) for P in range(k)])
        for P in range(k):
            a, b = want[it][P]
            assert np.array_equal(plain.shares(P, 0), a) and np.array_equal(plain.shares(P, 1), b), "iteration %d party %d" % (it, P)
    for P in range(k):
        m = [x for x in o.metrics if x["party"] == P][-1]
        g = plain.metrics(P)
        assert abs(g["loss"] - m["loss"]) < 1e-9 and g["full"] == pytest.approx(m["full"], abs=1e-12)
    plain.close()
    # the bench's sequence: forward-only stores, retained offline products, the pass run three times
    b = engine()
    b.retain_offline(True); b.forward_only(True); b.offline(0, iters)
    for _ in range(3):
        b.run(0, iters)
    for P in range(k):
        assert np.array_equal(b.shares(P, 0), want[-1][P][0]) and np.array_equal(b.shares(P, 1), want[-1][P][1]), "bench sequence, party %d" % P
    b.close()
    # the dealt form of the online phase: the same sequence with the pairs' dealer values read from HBM (COGNN_OPT_DEALER_STREAMS)
    d = engine()
    d.retain_offline(True); d.forward_only(True); d.dealer_streams(True); d.offline(0, iters)
    for _ in range(2):                                       # first pass deals, second reads what the first dealt
        d.run(0, iters)
    for P in range(k):
        assert np.array_equal(d.shares(P, 0), want[-1][P][0]) and np.array_equal(d.shares(P, 1), want[-1][P][1]), "dealer streams, party %d" % P
    d.close()
    # ... and the corrections-only form (COGNN_OPT_DEALER_STREAMS = 2): own-seed values regenerated, c_1 / r_1 / r'_1 / g read
    m = engine()
    m.retain_offline(True); m.forward_only(True); m.dealer_minimal(True); m.offline(0, iters)
    for _ in range(2):
        m.run(0, iters)
    for P in range(k):
        assert np.array_equal(m.shares(P, 0), want[-1][P][0]) and np.array_equal(m.shares(P, 1), want[-1][P][1]), "dealer-minimal form, party %d" % P
    m.close()
