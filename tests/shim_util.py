"""Helpers shared by tests/test_shim_cpu.py and tests/test_shim_gpu.py: build tests/gas_epochs.cpp against
include/cognn_gas_kernel.hpp + include/cognn_sci_shim.hpp + libcognn_hip.so with g++, write its input file from the oracle's
state, and restate the shim's dealer addressing (call counter per session) on top of the oracle so that the outputs can be
compared bit for bit."""
import os
import struct
import subprocess

import numpy as np

import cognn_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "gas_epochs")


def build(name="gas_epochs"):
    BIN = os.path.join(ROOT, "tests", "_build", name)
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    src = os.path.join(ROOT, "tests", name + ".cpp")
    deps = [src, os.path.join(ROOT, "include", "cognn_gas_kernel.hpp"), os.path.join(ROOT, "include", "cognn_sci_shim.hpp"),
            os.path.join(ROOT, "include", "cognn_hip.h"), os.path.join(ROOT, "cognn_amd", "libcognn_hip.so")]
    if os.path.exists(BIN) and all(os.path.getmtime(BIN) >= os.path.getmtime(d) for d in deps):
        return BIN
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I" + os.path.join(ROOT, "include"), src,
                           "-L" + os.path.join(ROOT, "cognn_amd"), "-lcognn_hip", "-Wl,-rpath," + os.path.join(ROOT, "cognn_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", BIN])
    return BIN


# shim op ids (include/cognn_sci_shim.hpp): product, its truncation, row scale, its truncation, ReLU, softmax, MatrixScale, ApplyGradient
S_GEMM, S_GEMM_T, S_SCALE, S_SCALE_T, S_RELU, S_SOFTMAX, S_MSCALE, S_LR = 10, 11, 12, 13, 16, 17, 20, 21


def shim_calls(owner, iters, inference=False):
    """(GAS iteration, engine op id) -> (call number of the owner's (owner, co) session, shim op id): the order in which the
    callbacks of gcn.h reach the protocol functions - PreScatterComp (:233,247), GatherComp (:476), ApplyComp (:549,578,665,671,
    676,678,705,710,723,730) and, in the session of party 0's pair, the weight-averaging scale (:764)."""
    m = {}
    c = 0
    for it in range(iters):
        e = it % 6
        if e in (0, 1):
            m[(it, co.OP_PS_GEMM)] = (c, S_GEMM); m[(it, co.OP_PS_GEMM_TRUNC)] = (c, S_GEMM_T); c += 1
            if e == 1:
                m[(it, co.OP_PS_SCALE)] = (c, S_SCALE); m[(it, co.OP_PS_SCALE_TRUNC)] = (c, S_SCALE_T); c += 1
            m[(it, co.OP_GA_SCALE)] = (c, S_SCALE); m[(it, co.OP_GA_SCALE_TRUNC)] = (c, S_SCALE_T); c += 1
            if e == 0:
                m[(it, co.OP_AP_RELU)] = (c, S_RELU); c += 1
            else:
                m[(it, co.OP_AP_SOFTMAX)] = (c, S_SOFTMAX); c += 1
        elif e == 2:
            m[(it, co.OP_AP_GEMM)] = (c, S_GEMM); m[(it, co.OP_AP_GEMM_TRUNC)] = (c, S_GEMM_T); c += 1
        elif e == 4:
            c += 1                                           # twoPartyGCNBackwardNNWithoutAH: the sign protocol on z (its result is public)
        else:                                                # e in (3, 5)
            m[(it, co.OP_PS_SCALE)] = (c, S_SCALE); m[(it, co.OP_PS_SCALE_TRUNC)] = (c, S_SCALE_T); c += 1
            if e == 3:
                m[(it, co.OP_GA_SCALE)] = (c, S_SCALE); m[(it, co.OP_GA_SCALE_TRUNC)] = (c, S_SCALE_T); c += 1
            m[(it, co.OP_AP_GEMM)] = (c, S_GEMM); m[(it, co.OP_AP_GEMM_TRUNC)] = (c, S_GEMM_T); c += 1
            m[(it, co.OP_AP_GSCALE_TRUNC)] = (c, S_MSCALE); c += 1
            m[(it, co.OP_AP_LR_TRUNC)] = (c, S_LR); c += 1
            if inference or owner == 0:                     # inference variant: every party scales its own weights (:680-681); else the average, in party 0's pair
                m[(it, co.OP_WAVG_TRUNC)] = (c, S_MSCALE); c += 1
    return m


class ShimKeyedOracle(co.OracleEngine):
    """The oracle with the shim's dealer addressing: call number c of owner P draws from (seed, P, c, shim op)."""
    MAX_ITERS = 24
    WGRAD_MASK_REUSED = False      # every shim product deals a fresh mask for its left operand as passed (logical order of the transposed tensor)

    def key_of(self, owner, it, op):
        if owner == co.OWNER_WAVG:                           # twoPartyGCNMatrixScale(..., 1 - tileIndex, tileIndex + 1): the pair of party 0
            owner = 0
        if not hasattr(self, "_calls"):
            self._calls = {}
        if owner not in self._calls:
            self._calls[owner] = shim_calls(owner, self.MAX_ITERS, inference=self.variant == "optimize-gcn-inference")
        c, sop = self._calls[owner][(it, op)]
        return lambda slot: co.stream_key(self.seed, owner, c, sop, slot)

    # the shim deals fresh masks per product call (the engine reuses the forward product's): the oracle follows the shim here, mask
    # for mask - since the truncation opening is formed from the top 48 bits of the two shares, the way a product is split into its two
    # shares reaches the truncated result (a dropped carry), so the masks matter
    def key_of_feature_gemm(self, owner, it):
        return self.key_of(owner, it, co.OP_PS_GEMM)

    def key_of_feature_wgrad(self, owner, it):
        return self.key_of(owner, it, co.OP_AP_GEMM)

    def key_of_hidden_wgrad(self, owner, it, it_fwd):
        return self.key_of(owner, it, co.OP_AP_GEMM)


def shim_calls_original(P, k, iters):
    """The unoptimised kernel (original-gcn/gcn.h, 4 GAS iterations per epoch) through include/cognn_gas_kernel.hpp: call numbers of
    the sessions of data owner P.  Returns (own, scatter): own[(it, engine op)] = (call number, shim op) in the session of P and its
    co-party - Gather's self scale (:365), ForwardNN / Prediction (:459,493), BackwardNNInit (:586: g first, then d), BackwardNN (:622:
    the sign protocol on z, then d), MatrixScale / ApplyGradient (:590,593,634,642), the averaging scale in party 0's pair (:676) -
    and scatter[(it, destination party i)] = (server tid of the session, call number): ScatterComp is one call of the session
    (P, i) - for P's local edges and for the edges towards its co-party both in the session (P, co(P)), local edges first."""
    cop = (P + 1) % k
    own, scatter = {}, {}
    cnt = {i: 0 for i in range(k) if i != P}                 # per session (P, server i)
    for it in range(iters):
        e = it % 4
        if e != 2:                                           # not the apply-only iteration
            scatter[(it, P)] = (cop, cnt[cop]); cnt[cop] += 1
            for i in range(k):
                if i != P:
                    scatter[(it, i)] = (i, cnt[i]); cnt[i] += 1
        c = cnt[cop]
        if e < 2:
            own[(it, co.OP_GA_SCALE)] = (c, S_SCALE); own[(it, co.OP_GA_SCALE_TRUNC)] = (c, S_SCALE_T); c += 1
            own[(it, OP_AP_FWD_GEMM)] = (c, S_GEMM); own[(it, OP_AP_FWD_GEMM_TRUNC)] = (c, S_GEMM_T); c += 1
            if e == 0:
                own[(it, co.OP_AP_RELU)] = (c, S_RELU); c += 1
            else:
                own[(it, co.OP_AP_SOFTMAX)] = (c, S_SOFTMAX); c += 1
        else:
            if e == 2:                                       # last layer: g = in . W^T, then d = ah_t . in
                own[(it, co.OP_AP_GEMM)] = (c, S_GEMM); own[(it, co.OP_AP_GEMM_TRUNC)] = (c, S_GEMM_T); c += 1
            else:                                            # first layer: the sign protocol on z (public result), no g
                c += 1
            own[(it, OP_AP_DGEMM)] = (c, S_GEMM); own[(it, OP_AP_DGEMM_TRUNC)] = (c, S_GEMM_T); c += 1
            own[(it, co.OP_AP_GSCALE_TRUNC)] = (c, S_MSCALE); c += 1
            own[(it, co.OP_AP_LR_TRUNC)] = (c, S_LR); c += 1
            if P == 0:
                own[(it, co.OP_WAVG_TRUNC)] = (c, S_MSCALE); c += 1
        cnt[cop] = c
    return own, scatter


OP_AP_DGEMM, OP_AP_DGEMM_TRUNC, OP_AP_FWD_GEMM, OP_AP_FWD_GEMM_TRUNC = 34, 35, 36, 37      # oracle/original_gcn.py


def keyed_original_oracle(*args, **kw):
    """oracle/original_gcn.py with the shim's dealer addressing (see shim_calls_original)."""
    import original_gcn

    class ShimKeyedOriginalOracle(original_gcn.OriginalOracleEngine):
        MAX_ITERS = 16

        def _maps(self, P):
            if not hasattr(self, "_calls"):
                self._calls = {}
            if P not in self._calls:
                self._calls[P] = shim_calls_original(P, self.k, self.MAX_ITERS)
            return self._calls[P]

        def key_of(self, owner, it, op):
            if owner >= 0x10000 and owner != co.OWNER_WAVG:  # a Scatter instance (client P, destination party i): pair_tag
                P, i = (owner - 0x10000) // 256, (owner - 0x10000) % 256
                server, c = self._maps(P)[1][(it, i)]
                tag = 0x10000 + P * 256 + server
                return lambda slot: co.stream_key(self.seed, tag, c, op, slot)
            if owner == co.OWNER_WAVG:
                owner = 0
            c, sop = self._maps(owner)[0][(it, op)]
            return lambda slot: co.stream_key(self.seed, owner, c, sop, slot)

    return ShimKeyedOriginalOracle(*args, **kw)


def _vec(f, a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64))
    f.write(struct.pack("<Q", a.size)); f.write(a.tobytes())


def _real(f, x):
    f.write(struct.pack("<Q", 1)); f.write(struct.pack("<d", float(x)))


def _mat(f, m):
    m = np.asarray(m, dtype=np.uint64)
    _vec(f, [m.shape[0]]); _vec(f, [m.shape[1]]); _vec(f, m.reshape(-1))


def write_input(path, o, iters, original=False):
    """State of a freshly started k-party OracleEngine in the order tests/gas_epochs.cpp reads it (original: with the per-edge degree
    vectors the unoptimised kernel's ScatterComp reads)."""
    k = o.k
    p = o.param
    with open(path, "wb") as f:
        _vec(f, [k]); _vec(f, [o.seed]); _vec(f, [iters])
        _vec(f, [p.input_dim]); _vec(f, [p.hidden_dim]); _vec(f, [p.num_labels])
        _real(f, p.learning_rate); _real(f, p.train_ratio); _real(f, p.val_ratio); _real(f, p.test_ratio)
        for t in range(k):
            gs = o.states[t]
            _vec(f, gs.localVertexPos); _vec(f, gs.localVertexInDeg); _vec(f, gs.labels); _vec(f, [int(b) for b in gs.isLocalVertexBorder])
            for j in range(k):
                _vec(f, gs.updateSrcVertexPos[j]); _vec(f, gs.updateDstVertexPos[j]); _vec(f, gs.remoteMirrorVertexPos[j])
                _vec(f, [int(b) for b in gs.isGatherDstVertexDummy[j]])
                if original:
                    _vec(f, gs.updateSrcOutDeg[j]); _vec(f, gs.updateDstInDeg[j]); _vec(f, gs.remoteUpdateDstInDeg[j])
            _mat(f, gs.localVertexSvv)
            for j in range(k):
                if j != t:
                    _mat(f, gs.remoteVertexSvvs[j])
            for l in range(2):
                _mat(f, gs.localWeight[l])
            for l in range(2):
                _mat(f, gs.remoteWeight[l])


def read_output(path, k, iters):
    data = open(path, "rb").read()
    pos = 0

    def mat():
        nonlocal pos
        r, c = struct.unpack_from("<QQ", data, pos); pos += 16
        a = np.frombuffer(data, dtype=np.uint64, count=r * c, offset=pos).reshape(r, c); pos += 8 * r * c
        return a
    per_iter = [[(mat(), mat()) for _ in range(k)] for _ in range(iters)]
    weights = [[(mat(), mat()) for _ in range(2)] for _ in range(k)]          # [party][layer] = (local, remote)
    probs = []
    metrics = []
    for _ in range(k):
        probs.append(mat())
        metrics.append(mat().reshape(-1).astype(np.int64) / 1e6)
    assert pos == len(data)
    return per_iter, weights, probs, metrics
