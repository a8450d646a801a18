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


def run_case(tmp_path, mode, k, V, Eu, in_dim, hid, lab, iters, graph_seed=3, original=False, inference=False):
    exe = shim_util.build()
    src, dst = co.synth_graph(V, Eu, graph_seed)
    part = [v % k for v in range(V)]
    feats, labels = co.synth_features(V, in_dim, lab, 4, density=0.25)
    p = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    if original:                                             # the unoptimised kernel's callbacks (GCNOriginalEdgeCentricAlgoKernel)
        o = shim_util.keyed_original_oracle(k, src, dst, part, feats, labels, p, seed=0xC06A11)
        mode = "o" + mode
    elif inference:                                          # optimize-gcn-inference/gcn.h run through whole training epochs
        o = shim_util.ShimKeyedOracle(k, src, dst, part, feats, labels, p, seed=0xC06A11, variant="optimize-gcn-inference")
        mode = "i" + mode
    else:
        o = shim_util.ShimKeyedOracle(k, src, dst, part, feats, labels, p, seed=0xC06A11)
    shim_util.write_input(tmp_path / "in.bin", o, iters, original=original)
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


@pytest.mark.parametrize("mode,k,V,Eu", [("device", 2, 40, 90), ("host", 2, 40, 90), ("device", 3, 60, 140), ("device", 4, 50, 60)])
def test_original_gcn_epochs_through_the_gas_kernel_match_the_oracle(tmp_path, mode, k, V, Eu):
    """The upper seam with the callbacks of the unoptimised kernel (original-gcn/gcn.h: copy PreScatter, per-edge two-normaliser
    ScatterComp, forward self scale in GatherComp, fused ForwardNN / Prediction / BackwardNNInit / BackwardNN Apply, weight average
    after both backward iterations; 4 GAS iterations per epoch): two epochs against oracle/original_gcn.py under the shim's dealer
    addressing, every share after every iteration and the weights, k = 2, 3, 4 (sparse: dummy self entries, empty Scatter instances)."""
    run_case(tmp_path, mode, k, V, Eu, 12, 8, 4, 8, original=True)


@pytest.mark.parametrize("k,V,Eu", [(2, 40, 90), (3, 60, 140)])
def test_inference_variant_epochs_through_the_gas_kernel(tmp_path, k, V, Eu):
    """GCNEdgeCentricAlgoKernel::inferenceVariant (optimize-gcn-inference/gcn.h: every party scales its updated weights by 1 / k itself,
    the averaging step does not): two whole epochs, bit-exact vs the oracle's inference variant."""
    run_case(tmp_path, "device", k, V, Eu, 12, 8, 4, 12, inference=True)


def test_device_mode_at_a_wider_shape_and_four_parties(tmp_path):
    """k = 4: two delegating servers per owner; widths that are not multiples of the kernels' tiles."""
    run_case(tmp_path, "device", 4, 90, 200, 33, 16, 7, 6, graph_seed=5)


def test_forward_iterations_on_host_vectors_at_the_round_2_shapes(tmp_path):
    run_case(tmp_path, "host", 2, 90, 100, 33, 16, 7, 2)


def test_on_algo_kernel_start_deals_the_oracles_shares(tmp_path):
    """onAlgoKernelStart (gcn.h:819-887): features * (inDeg + 1)^-1/2 in double, Glorot weights from srand(42), both split with
    CryptoUtil::intoShares - the owner's feature share is the oracle's, bit for bit; the weights reconstruct to the libc fixture."""
    import struct
    exe = shim_util.build()
    k, V, in_dim, hid, lab, seed = 2, 30, 9, 5, 3, 0xC06A11
    src, dst = co.synth_graph(V, 50, 2)
    feats, labels = co.synth_features(V, in_dim, lab, 4, density=0.3)
    p = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V)
    o = co.OracleEngine(k, src, dst, [v % k for v in range(V)], feats, labels, p, seed=seed)
    with open(tmp_path / "in.bin", "wb") as f:
        for v in (k, seed, in_dim, hid, lab):
            shim_util._vec(f, [v])
        for t in range(k):
            gs = o.states[t]
            vids = gs.localVertexPos
            shim_util._vec(f, [len(vids)])
            for v in vids:
                for x in np.asarray(feats)[v]:
                    shim_util._real(f, x)
            shim_util._vec(f, [gs.true_in_deg[v] for v in vids])
    r = subprocess.run([exe, "start", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    data = open(tmp_path / "out.bin", "rb").read()
    pos = 0

    def mat():
        nonlocal pos
        rr, c = struct.unpack_from("<QQ", data, pos); pos += 16
        a = np.frombuffer(data, dtype=np.uint64, count=rr * c, offset=pos).reshape(rr, c); pos += 8 * rr * c
        return a
    w_plain = [co.fx_encode(co.init_weight(in_dim, hid)), co.fx_encode(co.init_weight(hid, lab))]
    for t in range(k):
        s0, s1 = mat(), mat()
        assert np.array_equal(s0, o.states[t].localVertexSvvBackup) and np.array_equal(s1, o.states[t].featShare1)
        for l in range(2):
            w0, w1 = mat(), mat()
            with np.errstate(over="ignore"):
                assert np.array_equal(w0 + w1, w_plain[l])
    assert pos == len(data)


def _read_mats(path):
    import struct
    raw = open(path, "rb").read()
    out, off = [], 0
    while off < len(raw):
        r, c = struct.unpack_from("<QQ", raw, off); off += 16
        out.append(np.frombuffer(raw, dtype=np.uint64, count=r * c, offset=off).reshape(r, c)); off += 8 * r * c
    return out


def test_original_gcn_fused_ops_of_the_shim(tmp_path):
    """The op names of the unoptimised kernel (original-gcn/gcn.h:243,459,493,586,622: two-normaliser VectorScale, ForwardNN,
    ForwardNNPrediction, BackwardNNInit, BackwardNN) through the shim, client and server threads over a LocalPipe: each equals the
    sequence of single ops it is defined as (same seed, same inputs), on device-resident tensors and on nested host vectors alike;
    the two-normaliser scale reconstructs to x . n0 . n1 within the two truncations' slack."""
    exe = shim_util.build("shim_original_ops")
    got = {}
    for form in ("fused", "prim"):
        for where in ("device", "host"):
            out = tmp_path / ("%s_%s.bin" % (form, where))
            r = subprocess.run([exe, form, where, str(out)], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            got[(form, where)] = _read_mats(out)
    ref = got[("prim", "device")]
    assert len(ref) == 18                                       # 9 tensors per role
    for key, mats in got.items():
        assert len(mats) == len(ref), key
        for i, (a, b) in enumerate(zip(mats, ref)):
            assert a.shape == b.shape and np.array_equal(a, b), (key, i)
    with np.errstate(over="ignore"):
        sc = (ref[0] + ref[9]).astype(np.int64)                 # both roles' shares of the scaled tensor
    assert np.abs(sc).max() < (1 << 22) and np.abs(sc).max() > 0   # Q16 values of magnitude < 2^19 scaled by two factors < 1
    assert np.abs(sc[0::7]).max() <= 1                          # rows whose second normaliser is 0 reconstruct to 0 (+ 1 ulp of truncation)
