"""The sci:: / oblivious-mapper / prefix_network_aggregate drop-in shim on the GPU: tests/shim_iteration.cpp (plain g++,
the reference's call shapes and client / server thread structure, two parties in one process) runs GAS iterations 0 and 1 of
gcn-optimize through include/cognn_sci_shim.hpp -> libcognn_hip.so; every share it produces must equal the oracle's, bit
for bit, under the shim's dealer addressing (parity unpinned w.r.t. the reference, as everywhere)."""
import subprocess

import numpy as np
import pytest

import cognn_oracle as co
import shim_util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,Eu,in_dim,hid,lab", [(40, 90, 12, 8, 4), (90, 100, 33, 16, 7)])
def test_two_party_iterations_through_the_shim_match_the_oracle(tmp_path, V, Eu, in_dim, hid, lab):
    exe = shim_util.build()
    src, dst = co.synth_graph(V, Eu, 3)
    part = [v % 2 for v in range(V)]
    feats, labels = co.synth_features(V, in_dim, lab, 4, density=0.25)
    p = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    o = shim_util.ShimKeyedOracle(2, src, dst, part, feats, labels, p, seed=0xC06A11)
    shim_util.write_input(tmp_path / "in.bin", o, 2)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    per_iter, probs = shim_util.read_output(tmp_path / "out.bin", 2)
    for it in range(2):
        o.iteration(it)
        for t in range(2):
            a, b = o.shares(t)
            ga, gb = per_iter[it][t]
            assert ga.shape == a.shape and gb.shape == b.shape
            assert np.array_equal(ga, a), "iteration %d owner %d: client share differs" % (it, t)
            assert np.array_equal(gb, b), "iteration %d owner %d: server share differs" % (it, t)
    # getPlainShareVecVec (gcn.h:604): the revealed probabilities are the oracle's Q16 softmax
    for t in range(2):
        with np.errstate(over="ignore"):
            pfx = o.states[t].localInter[1]["p"] + o.states[1 - t].remoteInter[1]["p"]
        assert np.array_equal(probs[t], pfx)
