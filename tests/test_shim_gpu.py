"""The drop-in boundary on the GPU, both seams: tests/gas_epochs.cpp (plain g++) runs two full gcn-optimize training epochs
(GAS iterations 0-11: forward, prediction, the apply-only iterations, backward products with W^T and h_t, ReLU', gradient
scale / apply, weight averaging) of k = 2 and k = 3 parties through include/cognn_gas_kernel.hpp - the GAS operator API of
ss_vertex_centric_algo_kernel.h:78-133 and its client / server thread structure - over include/cognn_sci_shim.hpp -> C ABI ->
HIP, once on device-resident tensors (`device`) and once on the reference's nested host vectors (`host`).  Every share after
every iteration, every party's weight shares and the revealed probabilities must equal the oracle's, bit for bit, under the
shim's dealer addressing (parity unpinned w.r.t. the reference, as everywhere)."""
import subprocess

import numpy as np
import pytest

import cognn_oracle as co
import shim_util

pytestmark = pytest.mark.gpu


def run_case(tmp_path, mode, k, V, Eu, in_dim, hid, lab, iters, graph_seed=3):
    exe = shim_util.build()
    src, dst = co.synth_graph(V, Eu, graph_seed)
    part = [v % k for v in range(V)]
    feats, labels = co.synth_features(V, in_dim, lab, 4, density=0.25)
    p = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    o = shim_util.ShimKeyedOracle(k, src, dst, part, feats, labels, p, seed=0xC06A11)
    shim_util.write_input(tmp_path / "in.bin", o, iters)
    r = subprocess.run([exe, mode, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    per_iter, weights, probs, metrics = shim_util.read_output(tmp_path / "out.bin", k, iters)
    for it in range(iters):
        o.iteration(it)
        for t in range(k):
            a, b = o.shares(t)
            ga, gb = per_iter[it][t]
            if a.shape[1] == 0:                              # vertexInterData["g"] of the first layer is empty (gcn.h:702-708)
                assert ga.size == 0 and gb.size == 0
                continue
            assert ga.shape == a.shape and gb.shape == b.shape, "iteration %d owner %d: shape %s vs %s" % (it, t, ga.shape, a.shape)
            assert np.array_equal(ga, a), "iteration %d owner %d: client share differs" % (it, t)
            assert np.array_equal(gb, b), "iteration %d owner %d: server share differs" % (it, t)
    for t in range(k):
        for l in range(2):
            assert np.array_equal(weights[t][l][0], o.states[t].localWeight[l]), "party %d layer %d: local weight share differs" % (t, l)
            assert np.array_equal(weights[t][l][1], o.states[t].remoteWeight[l]), "party %d layer %d: remote weight share differs" % (t, l)
    # getPlainShareVecVec (gcn.h:604): the revealed probabilities are the oracle's Q16 softmax; the client's metric lines follow
    if iters >= 2:
        for t in range(k):
            with np.errstate(over="ignore"):
                pfx = o.states[t].localInter[1]["p"] + o.states[o.co(t)].remoteInter[1]["p"]
            want = np.where(pfx == 0, np.uint64(66), pfx)    # plainP == 0 -> 0.001 before the loss (gcn.h:613-615): llround(0.001 * 2^16)
            assert np.array_equal(probs[t], want)
            m = [x for x in o.metrics if x["party"] == t][-1]
            got = metrics[t]
            assert abs(got[0] - m["loss"]) < 2e-6
            for j, key in enumerate(("full", "train", "border_train", "test", "border_test")):
                assert abs(got[1 + j] - 100.0 * m[key]) < 2e-4, (key, got[1 + j], m[key])


@pytest.mark.parametrize("mode", ["device", "host"])
@pytest.mark.parametrize("k,V,Eu,in_dim,hid,lab", [(2, 40, 90, 12, 8, 4), (3, 60, 140, 12, 8, 4)])
def test_two_training_epochs_through_the_gas_kernel_match_the_oracle(tmp_path, mode, k, V, Eu, in_dim, hid, lab):
    run_case(tmp_path, mode, k, V, Eu, in_dim, hid, lab, 12)


def test_device_mode_at_a_wider_shape_and_four_parties(tmp_path):
    """k = 4: two delegating servers per owner; widths that are not multiples of the kernels' tiles."""
    run_case(tmp_path, "device", 4, 90, 200, 33, 16, 7, 6, graph_seed=5)


def test_forward_iterations_on_host_vectors_at_the_round_2_shapes(tmp_path):
    run_case(tmp_path, "host", 2, 90, 100, 33, 16, 7, 2)
