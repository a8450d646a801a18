"""End-to-end parity: the fused HIP engine vs the oracle's restatement of the reference schedule,
bit-exact on every party's shares after every GAS iteration (parity unpinned w.r.t. the reference)."""
import numpy as np
import pytest

import cognn_oracle as co

pytestmark = pytest.mark.gpu


def _setup(k, V, Eu, in_dim, hid, lab, variant="optimize-gcn", seed=7, gseed=1, **oracle_kw):
    from cognn_amd.engine import Engine, GnnParam
    src, dst = co.synth_graph(V, Eu, gseed)
    part = np.array([v % k for v in range(V)], dtype=np.int32)
    feats, labels = co.synth_features(V, in_dim, lab, gseed + 1, density=0.2)
    op = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    oracle = co.OracleEngine(k, src, dst, part, feats, labels, op, seed=seed, variant=variant, **oracle_kw)
    gp = GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    eng = Engine(k, src, dst, part, gp, seed=seed, variant=variant)
    eng.set_global_data(feats, labels)
    eng.start()
    return oracle, eng


def _compare(oracle, eng, k, it):
    for P in range(k):
        a, b = oracle.shares(P)
        ga, gb = eng.shares(P, 0), eng.shares(P, 1)
        assert ga.shape == a.shape and gb.shape == b.shape, (it, P, ga.shape, a.shape)
        assert np.array_equal(ga, a), "iter %d owner %d: owner share differs" % (it, P)
        assert np.array_equal(gb, b), "iter %d owner %d: co-party share differs" % (it, P)
        c = (P + 1) % k
        for l in range(2):
            assert np.array_equal(eng.weight(P, 0, l), oracle.states[P].localWeight[l]), (it, P, l)
            assert np.array_equal(eng.weight(P, 1, l), oracle.states[c].remoteWeight[l]), (it, P, l)


@pytest.mark.parametrize("pair_fusion", [True, False, "exchanged"])
@pytest.mark.parametrize("k", [2, 3, 4])
def test_training_two_epochs_bit_exact(k, pair_fusion):
    """pair_fusion: co-located share-holders run their steps as pair chains (exchange in registers, the default) or through the
    per-side open / close kernels a one-party-per-GPU run uses - with the opening after a truncation derived by both parties
    (COGNN_OPT_PUBLIC_OPENINGS, the default) or "exchanged" as two shares; the shares are the oracle's every way."""
    oracle, eng = _setup(k, 60, 150, 24, 8, 5)
    eng.pair_fusion(pair_fusion is True)
    eng.public_openings(pair_fusion != "exchanged")
    for P in range(k):
        assert list(eng.party_vids(P)) == oracle.states[P].localVertexPos
        t, i, b = eng.party_degrees(P)
        assert list(i) == oracle.states[P].localVertexInDeg
        assert list(b.astype(bool)) == oracle.states[P].isLocalVertexBorder
    _compare(oracle, eng, k, -1)
    for it in range(12):
        oracle.iteration(it)
        eng.run(it, it + 1)
        _compare(oracle, eng, k, it)
        if it % 6 == 1:
            for P in range(k):
                m = eng.metrics(P)
                om = [x for x in oracle.metrics if x["party"] == P and x["iter"] == it][0]
                for key in ("full", "train", "border_train", "test", "border_test"):
                    assert abs(m[key] - om[key]) < 1e-12, (key, m, om)
                assert abs(m["loss"] - om["loss"]) < 1e-9          # fp tolerance: atomic summation order
    eng.close()


@pytest.mark.parametrize("k,V,Eu,in_dim,hid,lab", [(2, 70, 170, 20, 8, 4), (3, 120, 400, 33, 16, 7), (8, 400, 1500, 24, 16, 6)])
def test_recorded_epochs_bit_exact(k, V, Eu, in_dim, hid, lab):
    """COGNN_OPT_GRAPH_EPOCHS: epoch 0 runs eagerly, epoch 1 while it is recorded (hipGraph), epochs 2-4 as replays of that
    recording under their own epoch salt; also several epochs per call.  Shares, weights and metrics after every epoch are those of
    the oracle that renews the feature operand's Beaver mask every epoch, as recorded epochs do (their kernel arguments may not depend
    on the epoch).  (The eager form deals that mask once; the two forms differ by single LSBs - the carry the 48-bit truncation opening
    drops depends on how the product is split into shares.)"""
    oracle, eng = _setup(k, V, Eu, in_dim, hid, lab, seed=33, renew_feature_mask=True)
    eng.graph_epochs(True)

    def check(ep):
        _compare(oracle, eng, k, 6 * ep + 5)
        for P in range(k):
            m = eng.metrics(P)
            om = [x for x in oracle.metrics if x["party"] == P and x["iter"] == 6 * ep + 1][0]
            for key in ("full", "train", "border_train", "test", "border_test"):
                assert abs(m[key] - om[key]) < 1e-12, (ep, key, m, om)
            assert abs(m["loss"] - om["loss"]) < 1e-9
    for ep in range(3):
        for it in range(6 * ep, 6 * ep + 6):
            oracle.iteration(it)
        eng.offline(6 * ep, 6 * ep + 6)                       # (a no-op here: a recorded epoch deals inside the recording)
        eng.run(6 * ep, 6 * ep + 6)
        check(ep)
    for it in range(18, 30):                                  # two epochs in one call
        oracle.iteration(it)
    eng.run(18, 30)
    check(4)
    # an iteration-by-iteration stretch afterwards (eager) continues from the replayed state
    for it in range(30, 36):
        oracle.iteration(it)
        eng.run(it, it + 1)
        _compare(oracle, eng, k, it)
    eng.close()


@pytest.mark.parametrize("pair_fusion", [False, True])
def test_products_dealt_on_use_on_launch_lanes(monkeypatch, pair_fusion):
    """The per-side product path (COGNN_GEMM_PER_SIDE: what shapes the grouped launch does not serve take anyway) at a size where its
    launch lanes are in use, with the product shares dealt where they are used (no offline call: the loop then stays on one stream,
    its buffers come from the pool the consumers feed) - four epochs, every p = 1 side dealing into the buffers the epoch before
    released, pairs co-located or not."""
    monkeypatch.setenv("COGNN_GEMM_PER_SIDE", "1")
    oracle, eng = _setup(4, 2400, 9000, 128, 64, 16, seed=5, gseed=9)
    eng.pair_fusion(pair_fusion)
    for it in range(24):
        oracle.iteration(it)
        eng.run(it, it + 1)
        if it % 6 == 5:
            _compare(oracle, eng, 4, it)
    eng.close()


def test_recorded_and_eager_engines_interleaved():
    """Two engines in one process, one replaying recorded epochs (device-side epoch salt, private stream), one launching eagerly
    (host-salted keys), alternating epoch by epoch for four epochs: each ends every epoch in its own oracle's shares and weights -
    the recorded engine's salt is back to 0, and its ownership released, whenever its call returns.  A third engine that also
    records runs after the first is closed (one recorded-epoch engine at a time holds the device salt)."""
    import ctypes
    import torch
    from cognn_amd import capi
    oa, ea = _setup(3, 120, 400, 33, 16, 7, seed=33, renew_feature_mask=True)
    ob, eb = _setup(2, 70, 170, 20, 8, 4, seed=21, gseed=5)
    ea.graph_epochs(True)
    raw = capi.Context(0)                                     # a third user: the raw C ABI on the legacy default stream
    t0 = torch.zeros(1 << 16, dtype=torch.int64, device="cuda"); t1 = torch.ones(1 << 16, dtype=torch.int64, device="cuda")
    for ep in range(4):
        for it in range(6 * ep, 6 * ep + 6):
            oa.iteration(it); ob.iteration(it)
        ea.run(6 * ep, 6 * ep + 6)                            # (no sync in between: B's launches queue behind A's on another stream)
        eb.run(6 * ep, 6 * ep + 3)
        eb.run(6 * ep + 3, 6 * ep + 6)
        # default-stream work of other users between the replays, each followed by a wait: with the engine's zeroing recorded as
        # hipMemsetAsync nodes this is what made replays wrong (tools/repro_graph_memset_node.py)
        raw.call("cognn_add_u64", ctypes.c_void_p(t0.data_ptr()), ctypes.c_void_p(t0.data_ptr()), ctypes.c_void_p(t1.data_ptr()), 1 << 16)
        raw.sync()
        t0.add_(1); torch.cuda.synchronize()
        _compare(oa, ea, 3, 6 * ep + 5)
        _compare(ob, eb, 2, 6 * ep + 5)
    assert int(t0[0].item()) == 8
    raw.close()
    ea.close()
    oc, ec = _setup(2, 70, 170, 20, 8, 4, seed=5, gseed=2, renew_feature_mask=True)
    ec.graph_epochs(True)
    for ep in range(3):
        for it in range(6 * ep, 6 * ep + 6):
            oc.iteration(it); ob.iteration(24 + it)
        ec.run(6 * ep, 6 * ep + 6)
        eb.run(24 + 6 * ep, 24 + 6 * ep + 6)
        _compare(oc, ec, 2, 6 * ep + 5)
        _compare(ob, eb, 2, 24 + 6 * ep + 5)
    eb.close(); ec.close()


@pytest.mark.parametrize("k", [2, 3])
def test_whole_epochs_in_one_call_bit_exact(k):
    """cognn_engine_run over whole epochs without reading anything in between: the paths that only exist across GAS iterations
    (the opening the ReLU leaves for the next product, the backward ReLU' selection deferred into the row-scale chain of the
    next iteration) end in the oracle's weights and shares."""
    oracle, eng = _setup(k, 70, 170, 20, 8, 4, seed=21)
    for ep in range(2):
        for it in range(6 * ep, 6 * ep + 6):
            oracle.iteration(it)
        eng.run(6 * ep, 6 * ep + 6)
        _compare(oracle, eng, k, 6 * ep + 5)
    # ... and with a reader between the ReLU' and its consumer on one side only
    for it in range(12, 18):
        oracle.iteration(it)
        eng.run(it, it + 1)
        if it == 16:
            eng.shares(0, 1)
    _compare(oracle, eng, k, 17)
    eng.close()


def test_forward_only_inference_pass_matches():
    """COGNN_OPT_FORWARD_ONLY (what bench.py and `gcn-inference-optimize -m 2` set): the prediction layer's shares and metrics are those of
    the full path; a backward iteration is refused."""
    from cognn_amd import capi
    k = 4
    oracle, eng = _setup(k, 80, 240, 16, 16, 4, variant="optimize-gcn-inference", seed=11)
    eng.forward_only(True)
    for it in range(2):
        oracle.iteration(it)
    eng.run(0, 2)
    _compare(oracle, eng, k, 1)
    for P in range(k):
        m = eng.metrics(P)
        om = [x for x in oracle.metrics if x["party"] == P][0]
        assert abs(m["loss"] - om["loss"]) < 1e-9 and abs(m["full"] - om["full"]) < 1e-12
    with pytest.raises(capi.CognnError, match="FORWARD_ONLY"):
        eng.run(2, 3)
    eng.close()


def test_inference_variant_and_offline_phase():
    k = 4
    oracle, eng = _setup(k, 80, 240, 16, 16, 4, variant="optimize-gcn-inference", seed=11)
    eng.offline(0, 2)
    for it in range(2):
        oracle.iteration(it)
    eng.run(0, 2)
    _compare(oracle, eng, k, 1)
    # a second pass over the same iterations reproduces the same shares (bench steps repeat iterations 0-1)
    eng.run(0, 2)
    _compare(oracle, eng, k, 1)
    eng.close()


@pytest.mark.parametrize("pair_fusion", [True, False])
def test_odd_dims_cora_like_shapes(pair_fusion):
    """labels=7 (odd row width -> scalar gather path), hidden 16, several isolated vertices."""
    k = 2
    oracle, eng = _setup(k, 90, 100, 33, 16, 7, seed=3, gseed=5)
    eng.pair_fusion(pair_fusion)
    for it in range(6):
        oracle.iteration(it)
        eng.run(it, it + 1)
        _compare(oracle, eng, k, it)
    eng.close()


def test_reconstruction_tracks_plaintext_gcn():
    k = 3
    oracle, eng = _setup(k, 60, 150, 24, 8, 5, seed=21)
    pl = co.PlainEngine(oracle)
    for it in range(6):
        pl.iteration(it)
        oracle.iteration(it)
        eng.run(it, it + 1)
        for P in range(k):
            with np.errstate(over="ignore"):
                rec = co.fx_decode(eng.shares(P, 0) + eng.shares(P, 1))
            if rec.shape[1]:
                assert np.abs(rec - pl.X[P]).max() < 2e-4      # fixed-point tolerance (f=16, +-1 LSB truncations)
    eng.close()


# ---- BASELINE.json configs[1..3]: dataset-shaped synthetic stand-ins (the real Planetoid files are not available offline;
# sizes from build_from_source/config/*.txt, partition vid % k as in tools/data_transform.py:19-27) ----
DATASET_SHAPES = {
    # name: (parties, vertices, directed edges, input_dim, hidden, labels, lr, train_ratio)
    "cora-2p": (2, 2708, 10556, 1433, 16, 7, 0.5, 0.2),
    "citeseer-2p": (2, 3312, 10016, 3703, 16, 6, 0.8, 0.2),
    "pubmed-4p": (4, 19717, 128146, 500, 16, 3, 8.0, 0.05),
}


@pytest.mark.parametrize("name,iters", [("cora-2p", 12), ("citeseer-2p", 6), ("pubmed-4p", 6)])
def test_dataset_shaped_training_bit_exact(name, iters):
    from cognn_amd.engine import Engine, GnnParam
    k, V, E, in_dim, hid, lab, lr, tr = DATASET_SHAPES[name]
    src, dst = co.synth_graph(V, E // 2, 1)
    part = np.array([v % k for v in range(V)], dtype=np.int32)
    feats, labels = co.synth_features(V, in_dim, lab, 2, density=0.01)
    kw = dict(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=lr, train_ratio=tr,
              val_ratio=0.2 if tr == 0.2 else 0.15, test_ratio=0.6 if tr == 0.2 else 0.8)
    oracle = co.OracleEngine(k, src, dst, part, feats, labels, co.GnnParam(**kw), seed=0xC06A11)
    eng = Engine(k, src, dst, part, GnnParam(**kw), seed=0xC06A11)
    eng.set_global_data(feats, labels)
    eng.start()
    if name == "citeseer-2p":
        eng.pair_fusion(False)                       # one of the dataset shapes through the per-side kernels
    for it in range(iters):
        oracle.iteration(it)
        eng.run(it, it + 1)
        if it % 6 in (1, 5) or it == iters - 1:
            _compare(oracle, eng, k, it)
    eng.close()
