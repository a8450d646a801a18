"""The drop-in headers (include/cognn_sci_shim.hpp, include/cognn_gas_kernel.hpp) build with plain g++ against the C ABI, and a program using them fails loudly
- not silently on some CPU path - when there is no HIP device."""
import subprocess

import numpy as np

import cognn_oracle as co
import shim_util


def test_shim_program_builds_and_refuses_to_run_without_a_gpu(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    exe = shim_util.build()
    import torch
    if torch.cuda.is_available():
        return
    V = 12
    src, dst = co.synth_graph(V, 20, 1)
    feats, labels = co.synth_features(V, 6, 3, 2, density=0.3)
    o = shim_util.ShimKeyedOracle(2, src, dst, [v % 2 for v in range(V)], feats, labels,
                                  co.GnnParam(num_labels=3, input_dim=6, hidden_dim=4, num_samples=V), seed=5)
    shim_util.write_input(tmp_path / "in.bin", o, 2)
    r = subprocess.run([exe, "device", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no HIP device" in r.stderr
    # ... and so does the program that drives the op names of the unoptimised kernel (original-gcn) through the shim
    exe2 = shim_util.build("shim_original_ops")
    r = subprocess.run([exe2, "fused", "device", str(tmp_path / "o.bin")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no HIP device" in r.stderr


def test_shim_host_codecs():
    """CryptoUtil stand-ins restated in the header agree with the oracle's definitions (compiled into a tiny probe)."""
    import os
    import tempfile
    code = r'''
#include "cognn_sci_shim.hpp"
#include <cstdio>
int main() {
    const double v[5] = {0.0, 1.0, -1.5, 0.1234567, -3.99999};
    CryptoUtil::sharingSeedIs(77, 3);
    for (double x : v) {
        uint64_t a, b; CryptoUtil::intoShares(x, a, b);
        printf("%llu %llu %llu %.10f\n", (unsigned long long)CryptoUtil::encodeDoubleAsFixedPoint(x), (unsigned long long)a, (unsigned long long)b,
               CryptoUtil::mergeShareAsDouble(a, b));
    }
    ShareTensor t = transpose(ShareTensor{{1, 2, 3}, {4, 5, 6}});
    printf("%zu %zu %llu\n", t.size(), t[0].size(), (unsigned long long)t[2][1]);
    ShareVec h = toShareVec(2, 4);
    printf("%llu %llu\n", (unsigned long long)h[2], (unsigned long long)h[1]);
    return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.cpp"), "w").write(code)
        subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(shim_util.ROOT, "include"), os.path.join(d, "p.cpp"),
                               "-L" + os.path.join(shim_util.ROOT, "cognn_amd"), "-lcognn_hip", "-Wl,-rpath," + os.path.join(shim_util.ROOT, "cognn_amd"),
                               "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", os.path.join(d, "p")])
        out = subprocess.run([os.path.join(d, "p")], capture_output=True, text=True, check=True).stdout.split("\n")
    vals = [0.0, 1.0, -1.5, 0.1234567, -3.99999]
    key = co.stream_key(77, 3, 0, co.OP_SHARE_FEAT, 0)
    mask = co.prng(key, 5)
    for i, x in enumerate(vals):
        fx, a, b, merged = out[i].split()
        assert int(fx) == int(co.fx_encode(np.array([x]))[0])
        assert int(b) == int(mask[i]) and (int(a) + int(b)) % (1 << 64) == int(fx)
        assert abs(float(merged) - x) < 2.0 ** -16
    assert out[5].split() == ["3", "2", "6"] and out[6].split() == [str(1 << 16), "0"]
