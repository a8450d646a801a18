"""Randomised end-to-end parity: random party counts, graph sizes, partitions (incl. uneven ones) and layer widths through the
HIP engine vs the oracle, bit-exact on every party's shares after every GAS iteration of one training epoch (or one inference
pass).  COGNN_FUZZ_SCALE multiplies the number of seeds for soak runs."""
import os

import numpy as np
import pytest

import cognn_oracle as co

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get("COGNN_FUZZ_SCALE", "1"))


@pytest.mark.parametrize("seed", range(10 * SCALE))
def test_engine_random_configuration(seed):
    from cognn_amd.engine import Engine, GnnParam
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(2, 6))
    V = int(rng.integers(k, 260))
    if seed % 5 == 0:                                                     # enough rows per party for the MFMA product kernels (M >= 256)
        k = int(rng.integers(2, 4))
        V = int(rng.integers(600, 3000))
    max_pairs = V * (V - 1) // 2
    Eu = int(min(max_pairs, rng.integers(0, 4 * V + 1)))
    in_dim = int(rng.choice([3, 8, 17, 32, 50, 129]))
    hid = int(rng.choice([2, 5, 8, 16, 33, 64]))
    lab = int(rng.choice([2, 3, 7, 16]))
    variant = "optimize-gcn" if seed % 3 else "optimize-gcn-inference"
    iters = 6 if variant == "optimize-gcn" else 2
    src, dst = co.synth_graph(V, Eu, 100 + seed) if Eu > 0 else (np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    if seed % 2:
        part = rng.integers(0, k, size=V).astype(np.int32)                # uneven, possibly empty parties
    else:
        part = (np.arange(V) % k).astype(np.int32)
    feats, labels = co.synth_features(V, in_dim, lab, 200 + seed, density=0.25)
    op = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    oracle = co.OracleEngine(k, src, dst, list(part), feats, labels, op, seed=seed, variant=variant)
    gp = GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    eng = Engine(k, src, dst, part, gp, seed=seed, variant=variant)
    eng.set_global_data(feats, labels)
    eng.start()
    eng.pair_fusion(seed % 4 != 3)                                        # every fourth configuration through the per-side kernels
    eng.public_openings(seed % 8 != 7)                                    # ... half of those with every opening exchanged as two shares
    try:
        for it in range(iters):
            oracle.iteration(it)
            eng.run(it, it + 1)
            for P in range(k):
                a, b = oracle.shares(P)
                assert np.array_equal(eng.shares(P, 0), a) and np.array_equal(eng.shares(P, 1), b), \
                    dict(seed=seed, k=k, V=V, Eu=Eu, dims=(in_dim, hid, lab), variant=variant, it=it, P=P)
                c = (P + 1) % k
                for l in range(2):
                    assert np.array_equal(eng.weight(P, 0, l), oracle.states[P].localWeight[l])
                    assert np.array_equal(eng.weight(P, 1, l), oracle.states[c].remoteWeight[l])
    finally:
        eng.close()
