"""BASELINE.json configs[0]: 2-party original-gcn on Cora, CPU reference path (no GPU).

oracle/original_gcn.py restates algo_kernels/vertex_centric/original-gcn/gcn.h (4 GAS iterations per epoch, per-edge
two-normaliser Scatter, self-row scale inside the forward Gather, fused forward/backward Apply, weight averaging after
both backward iterations).  PARITY UNPINNED (no reference vectors exist): what pins it is (a) the float64 plaintext
of the same schedule within fixed-point tolerance, (b) per-op identities of the Scatter scale, (c) a self-generated
digest fixture (tests/golden/make_golden.py) as a regression pin.  Datasets are not shipped: Cora is a shape-matched
synthetic stand-in (2708 vertices, 10556 directed edges, 1433 features, 7 labels, vid % 2 partition)."""
import hashlib
import json
import os

import numpy as np
import pytest

import cognn_oracle as co
import original_gcn as og

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
U64 = np.uint64


def _mk(k, V, Eu, in_dim, hid, lab, seed=99, lr=0.5, density=0.3, gseed=3, fseed=4):
    src, dst = co.synth_graph(V, Eu, gseed)
    part = [v % k for v in range(V)]
    feats, labels = co.synth_features(V, in_dim, lab, fseed, density=density)
    p = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=lr)
    return og.OriginalOracleEngine(k, src, dst, part, feats, labels, p, seed=seed)


def test_schedule_constants():
    o = _mk(2, 20, 30, 5, 4, 3)
    assert o.epoch_len() == 4                                                  # original-gcn/gcn.h:842-845
    assert [o.mp_width(i) for i in range(8)] == [5, 4, 3, 4] * 2               # :807-830
    assert [o.co_forward_layer(i) for i in range(4)] == [0, 1, 1, 0]           # :337-340
    apply_only = [(i % 4 != 0 and (i % 4) % 2 == 0) for i in range(4)]         # ss_...h:709
    assert apply_only == [False, False, True, False]


def test_scatter_scales_each_edge_by_both_normalisers():
    """ScatterComp (:211-251): the client supplies the source normaliser (and the destination one for local edges), the
    server the destination normaliser of remote edges (ss_...h:800 vs :1041-1043); the reconstructed per-edge value is
    x * fx(n0) * fx(n1) up to one LSB per truncation."""
    o = _mk(2, 30, 60, 4, 3, 3)
    for P in range(2):
        gs = o.states[P]
        for i in range(2):
            E = len(gs.updateSrcVertexPos[i])
            rng = np.random.default_rng(P * 2 + i)
            x = rng.integers(-(1 << 20), 1 << 20, size=(E, 4)).astype(np.int64).astype(U64)
            with np.errstate(over="ignore"):
                xB = co.prng_shape(77, x.shape); xA = x - xB
                yA, yB = o._scatter_pair(P, i, 0, xA, xB)
                y = (yA + yB).astype(np.int64)
            n0 = co.normalizer(gs.updateSrcOutDeg[i]).astype(np.int64)
            din = gs.updateDstInDeg[i] if i == P else o.states[i].remoteUpdateDstInDeg[P]
            n1 = co.normalizer(din).astype(np.int64)
            step = (x.astype(np.int64) * n0[:, None]) >> 16                    # each truncation yields floor + {-1, 0, +1}
            lo = (((step - 1) * n1[:, None]) >> 16) - 1                        # normalisers are >= 0: monotone in step
            hi = (((step + 1) * n1[:, None]) >> 16) + 1
            assert ((y >= lo) & (y <= hi)).all()
            if i != P:
                assert (np.asarray(gs.updateDstInDeg[i]) == 0).all()          # ss_...h:499: the client does not know them


@pytest.mark.parametrize("k", [2, 3, 4])
def test_reconstruction_tracks_float64_plaintext(k):
    o = _mk(k, 48, 110, 10, 6, 4)
    pl = og.OriginalPlainEngine(o)
    for it in range(12):                                                       # three epochs
        o.iteration(it); pl.iteration(it)
        for P in range(k):
            r = o.reconstruct(P)
            assert r.shape == pl.X[P].shape
            if r.shape[1]:
                assert np.abs(r - pl.X[P]).max() < 3e-4
            for l in range(2):
                assert np.abs(o.weight(P, l) - pl.W[P][l]).max() < 3e-4
        # every party holds the same averaged weights after a backward iteration (:659-711)
        if it % 4 >= 2:
            layer = o.co_forward_layer(it)
            with np.errstate(over="ignore"):
                w = [o.states[P].localWeight[layer] + o.states[o.co(P)].remoteWeight[layer] for P in range(k)]
            for P in range(1, k):
                assert np.array_equal(w[0], w[P])
    assert len(o.metrics) == 3 * k and {m["iter"] for m in o.metrics} == {1, 5, 9}


def test_forward_gather_scales_self_row_once_and_backward_not_at_all():
    """GatherComp :365-381: forward iterations multiply the vertex row by (inDeg+1)^-1/2 when updateSrcTid == 0; the backward
    iteration adds the updates to the unscaled row."""
    o = _mk(2, 24, 0, 4, 3, 3)                                                 # no edges: every vertex has only its dummy self source
    pl = og.OriginalPlainEngine(o)
    x0 = [o.reconstruct(P).copy() for P in range(2)]
    o._prescatter_pair(0, 0); o._prescatter_pair(1, 0)
    for P in range(2):
        o._message_passing_it(P, 0)
    for P in range(2):
        o._extend_updates(P)
    for P in range(2):
        o._gather_pair(P, 0)
        s = 2.0 ** -0.5                                                        # inflated in-degree 1 (ss_...h:411-418)
        assert np.abs(o.reconstruct(P) - x0[P] * s).max() < 1e-4               # dummy updates masked, self row scaled once
    del pl


def test_small_run_digests():
    gold = json.load(open(os.path.join(GOLD, "original_gcn_small_run.json")))
    src, dst = co.synth_graph(gold["V"], gold["Eu"], gold["graph_seed"])
    part = [v % gold["k"] for v in range(gold["V"])]
    feats, labels = co.synth_features(gold["V"], gold["in"], gold["lab"], gold["feat_seed"], density=0.25)
    p = co.GnnParam(num_labels=gold["lab"], input_dim=gold["in"], hidden_dim=gold["hid"], num_samples=gold["V"], learning_rate=0.5)
    o = og.OriginalOracleEngine(gold["k"], src, dst, part, feats, labels, p, seed=gold["seed"])
    for it, want in enumerate(gold["digests"]):
        o.iteration(it)
        h = hashlib.sha256()
        for P in range(gold["k"]):
            a, b = o.shares(P)
            h.update(np.ascontiguousarray(a).tobytes()); h.update(np.ascontiguousarray(b).tobytes())
            for l in range(2):
                h.update(np.ascontiguousarray(o.states[P].localWeight[l]).tobytes())
                h.update(np.ascontiguousarray(o.states[P].remoteWeight[l]).tobytes())
        assert h.hexdigest() == want, "original-gcn oracle changed at iteration %d" % it
    for got, want in zip(o.metrics, gold["metrics"]):
        for key, v in want.items():
            assert got[key] == pytest.approx(v, abs=1e-9)


def test_config1_two_party_cora_shaped_two_epochs():
    """configs[0] itself: 2 parties, Cora's shape (build_from_source/config/cora_config.txt), 8 GAS iterations = 2 epochs as
    in the reference's smallest run.  Every iteration's reconstructed vertex tensor and weights track the float64 plaintext
    of the same schedule; the loss printed by the client (gcn.h:539) falls from epoch 1 to epoch 2."""
    cfg = json.load(open(os.path.join(GOLD, "reference_configs.json")))["cora_config.txt"]
    V, E = int(cfg["num_samples"]), int(cfg["num_edges"])
    src, dst = co.synth_graph(V, E // 2, 1)
    part = [v % 2 for v in range(V)]
    feats, labels = co.synth_features(V, int(cfg["input_dim"]), int(cfg["num_labels"]), 2, density=0.0127)
    p = co.GnnParam(num_labels=int(cfg["num_labels"]), input_dim=int(cfg["input_dim"]), hidden_dim=int(cfg["hidden_dim"]),
                    num_samples=V, learning_rate=float(cfg["learning_rate"]), train_ratio=float(cfg["train_ratio"]),
                    val_ratio=float(cfg["val_ratio"]), test_ratio=float(cfg["test_ratio"]))
    o = og.OriginalOracleEngine(2, src, dst, part, feats, labels, p, seed=0xC06A11)
    pl = og.OriginalPlainEngine(o)
    for it in range(8):
        o.iteration(it); pl.iteration(it)
        for P in range(2):
            r = o.reconstruct(P)
            assert r.shape == pl.X[P].shape
            if r.shape[1]:
                assert np.abs(r - pl.X[P]).max() < 1e-3        # K = 1433 products accumulate +-1 LSB input errors
            for l in range(2):
                assert np.abs(o.weight(P, l) - pl.W[P][l]).max() < 1e-3
    for P in range(2):
        losses = [m["loss"] for m in o.metrics if m["party"] == P]
        assert len(losses) == 2 and losses[1] < losses[0]
        assert o.reconstruct(P).shape == (len(o.states[P].localVertexPos), 0)   # g is skipped for the first layer
