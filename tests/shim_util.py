"""Helpers shared by tests/test_shim_cpu.py and tests/test_shim_gpu.py: build tests/shim_iteration.cpp against
include/cognn_sci_shim.hpp + libcognn_hip.so with g++, write its input file from the oracle's state, and restate the shim's
dealer addressing (call counter per session) on top of the oracle so that the outputs can be compared bit for bit."""
import os
import struct
import subprocess

import numpy as np

import cognn_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "shim_iteration")


def build():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    src = os.path.join(ROOT, "tests", "shim_iteration.cpp")
    deps = [src, os.path.join(ROOT, "include", "cognn_sci_shim.hpp"), os.path.join(ROOT, "include", "cognn_hip.h"),
            os.path.join(ROOT, "cognn_amd", "libcognn_hip.so")]
    if os.path.exists(BIN) and all(os.path.getmtime(BIN) >= os.path.getmtime(d) for d in deps):
        return BIN
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I" + os.path.join(ROOT, "include"), src,
                           "-L" + os.path.join(ROOT, "cognn_amd"), "-lcognn_hip", "-Wl,-rpath," + os.path.join(ROOT, "cognn_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", BIN])
    return BIN


# (GAS iteration, engine op id) -> (call number of the owner's session, shim op id): the order in which the callbacks of
# iterations 0 and 1 reach the protocol functions (gcn.h:233,247,476,549,578)
SHIM_CALLS = {
    (0, co.OP_PS_GEMM): (0, 10), (0, co.OP_PS_GEMM_TRUNC): (0, 11),
    (0, co.OP_GA_SCALE): (1, 12), (0, co.OP_GA_SCALE_TRUNC): (1, 13),
    (0, co.OP_AP_RELU): (2, 16),
    (1, co.OP_PS_GEMM): (3, 10), (1, co.OP_PS_GEMM_TRUNC): (3, 11),
    (1, co.OP_PS_SCALE): (4, 12), (1, co.OP_PS_SCALE_TRUNC): (4, 13),
    (1, co.OP_GA_SCALE): (5, 12), (1, co.OP_GA_SCALE_TRUNC): (5, 13),
    (1, co.OP_AP_SOFTMAX): (6, 17),
}


class ShimKeyedOracle(co.OracleEngine):
    """The oracle with the shim's dealer addressing: call number c of owner P draws from (seed, P, c, shim op)."""

    def key_of(self, owner, it, op):
        c, sop = SHIM_CALLS[(it, op)]
        return lambda slot: co.stream_key(self.seed, owner, c, sop, slot)

    def key_of_feature_gemm(self, owner, it):
        return self.key_of(owner, it, co.OP_PS_GEMM)


def _vec(f, a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64))
    f.write(struct.pack("<Q", a.size)); f.write(a.tobytes())


def _mat(f, m):
    m = np.asarray(m, dtype=np.uint64)
    _vec(f, [m.shape[0]]); _vec(f, [m.shape[1]]); _vec(f, m.reshape(-1))


def write_input(path, o, iters):
    """State of a freshly started 2-party OracleEngine in the order tests/shim_iteration.cpp reads it."""
    assert o.k == 2
    with open(path, "wb") as f:
        _vec(f, [o.seed]); _vec(f, [iters])
        for t in range(2):
            gs = o.states[t]
            n = len(gs.localVertexPos)
            _vec(f, gs.localVertexPos); _vec(f, gs.localVertexInDeg); _vec(f, gs.labels); _vec(f, [int(n * o.param.train_ratio)])
            for j in range(2):
                _vec(f, gs.updateSrcVertexPos[j]); _vec(f, gs.updateDstVertexPos[j]); _vec(f, gs.remoteMirrorVertexPos[j])
                _vec(f, [int(b) for b in gs.isGatherDstVertexDummy[j]])
            _mat(f, gs.localVertexSvv)
            _mat(f, gs.remoteVertexSvvs[1 - t])
            for l in range(2):
                _mat(f, gs.localWeight[l])
            for l in range(2):
                _mat(f, gs.remoteWeight[l])


def read_output(path, iters):
    data = open(path, "rb").read()
    pos = 0

    def mat():
        nonlocal pos
        r, c = struct.unpack_from("<QQ", data, pos); pos += 16
        a = np.frombuffer(data, dtype=np.uint64, count=r * c, offset=pos).reshape(r, c); pos += 8 * r * c
        return a
    per_iter = [[(mat(), mat()) for _ in range(2)] for _ in range(iters)]
    probs = [mat() for _ in range(2)]
    assert pos == len(data)
    return per_iter, probs
