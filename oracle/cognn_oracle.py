"""CPU ORACLE (test infrastructure, NOT product code).

Numpy restatement of CoGNN's secret-shared GCN hot path, used only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.  The product
(cognn_amd/) never imports this module.

PARITY UNPINNED: the reference tree holds no golden vectors / tests for this path and
delegates every arithmetic op on shares to external libraries that are absent
(SURVEY.md F1-F5, §8c).  What IS restated line by line from the reference:
  * graph -> index-array preprocessing   include/ss_vertex_centric_algo_kernel.h:279-534
  * degree accounting                    include/graph.h:607-633, include/graph_io_util.h:167-177
  * per-iteration client/server schedule include/ss_vertex_centric_algo_kernel.h:680-910, 912-1189
  * GAS callbacks, init, constants       algo_kernels/vertex_centric/optimize-gcn/gcn.h:198-948
  * inference variant deltas             algo_kernels/vertex_centric/optimize-gcn-inference/gcn.h:680-681,732-733,943
The share arithmetic (fixed point, dealer, Beaver, truncation, ReLU, softmax) follows the
definitions frozen in DESIGN.md §3; the HIP engine implements the
same definitions with a different (fused CSR) algorithm, so agreement is a real check.

The reference runs one process per party with k-1 client and k-1 server threads; this
restatement emulates all k parties sequentially in one process, keeping each party's
state separate (class PartyState mirrors GraphSummary, ss_...h:24-58).
"""
import ctypes
import math
import numpy as np

U64 = np.uint64
MASK64 = (1 << 64) - 1

# ----------------------------------------------------------------------------------------
# Fixed point + counter PRNG
# ----------------------------------------------------------------------------------------
SCALER_BIT_LENGTH = 16            # reference: external constant, only "< 31" is pinned (gcn.h:191)
FX_ONE = 1 << SCALER_BIT_LENGTH
GAMMA = 0x9E3779B97F4A7C15
M1 = 0xBF58476D1CE4E5B9
M2 = 0x94D049BB133111EB
RELU_T_MIN = 1 << 17              # the ReLU's multiplier t is at least 2^17: |z t| >= 2^17 for z != 0, so the 48-bit reading of w = z t has z's sign
TRUNC_OFFSET = 1 << 61            # makes x+offset non-negative for |x| < 2^61
TRUNC_MASK = (1 << 62) - 1

# dealer op ids / slots (must match cognn_amd/csrc/cognn_spec.h)
OP_SHARE_FEAT, OP_SHARE_W = 1, 2
(OP_PS_GEMM, OP_PS_GEMM_TRUNC, OP_PS_SCALE, OP_PS_SCALE_TRUNC, OP_GA_SCALE, OP_GA_SCALE_TRUNC,
 OP_AP_RELU, OP_AP_SOFTMAX, OP_AP_GEMM, OP_AP_GEMM_TRUNC, OP_AP_GSCALE_TRUNC, OP_AP_LR_TRUNC,
 OP_WAVG_TRUNC) = range(10, 23)
(SL_A0, SL_A1, SL_B0, SL_B1, SL_C0, SL_R, SL_R0, SL_RP0, SL_T, SL_T0, SL_RHO) = range(11)
OWNER_WAVG = 0xFFFF


def mix64_int(z):
    z &= MASK64
    z ^= z >> 30
    z = (z * M1) & MASK64
    z ^= z >> 27
    z = (z * M2) & MASK64
    z ^= z >> 31
    return z


def derive(key, tag):
    return mix64_int(((key ^ mix64_int(tag + GAMMA)) + GAMMA) & MASK64)


def stream_key(seed, owner, it, op, slot):
    return derive(derive(derive(derive(seed, owner), it), op), slot)


def mix64_np(z):
    z = z.astype(U64, copy=True)
    z ^= z >> U64(30)
    z *= U64(M1)
    z ^= z >> U64(27)
    z *= U64(M2)
    z ^= z >> U64(31)
    return z


PRNG_C = (0x97D6730E, 0xC140D344, 0xF849CBC2, 0xEC6F6B54)    # even multipliers: every round is a bijection


def prng(key, n, start=0):
    """prng(key, idx), idx = start..start+n-1: four multiply-fold rounds on z = idx ^ key -
    z += lo32(z) * C_i, then lo32(z) ^= hi32(z) between rounds (the middle fold also adds hi32(key))."""
    lo_mask, hi_mask = U64(0xFFFFFFFF), U64(0xFFFFFFFF00000000)
    with np.errstate(over="ignore"):
        z = np.arange(start, start + n, dtype=U64) ^ U64(key)
        key_hi = U64(key >> 32)
        for i, c in enumerate(PRNG_C):
            z = z + (z & lo_mask) * U64(c)
            if i < 3:
                lo = (z & lo_mask) ^ (z >> U64(32))
                if i == 1:
                    lo = (lo + key_hi) & lo_mask
                z = (z & hi_mask) | lo
        return z


def prng_shape(key, shape):
    n = int(np.prod(shape))
    return prng(key, n).reshape(shape)


def limb_value(w):
    """The signed-digit reading of 64-bit words: sum_i int8(byte_i(w)) * 256^i mod 2^64 = w - (((w >> 7) & 0x0101..01) << 8).
    A bijection of the words.  The Beaver A masks of the ring products are DEFINED this way: the GPU product kernels
    split operands into signed 8-bit limbs, and for a mask defined like this the limbs are the bytes of the PRNG word."""
    w = np.asarray(w, dtype=U64)
    with np.errstate(over="ignore"):
        return w - (((w >> U64(7)) & U64(0x0101010101010101)) << U64(8))


def gemm_mask_shape(key, shape):
    """A mask of a Beaver product's left operand (limb form), logical row-major element order."""
    return limb_value(prng_shape(key, shape))


def fx_encode(x):
    """CryptoUtil::encodeDoubleAsFixedPoint stand-in (gcn.h:220): llround(x*2^f), two's complement."""
    a = np.asarray(x, dtype=np.float64) * FX_ONE
    t = np.trunc(a)
    frac = a - t                                                  # exact
    r = t + (frac >= 0.5).astype(np.float64) - (frac <= -0.5).astype(np.float64)   # llround
    return r.astype(np.int64).astype(U64)


def fx_encode_trunc(x):
    """static_cast<uint64_t>(v*(1<<SCALER_BIT_LENGTH)) as written in gcn.h:676,678,764."""
    return U64(int(x * FX_ONE))


def fx_decode(v):
    return np.asarray(v, dtype=U64).astype(np.int64).astype(np.float64) / FX_ONE


def pow_neg_half(deg_plus_one):
    """libm pow(x, -0.5) element by element (numpy's vectorised pow may differ from libm by an ulp)."""
    x = np.asarray(deg_plus_one, dtype=np.float64)
    uniq, inv = np.unique(x, return_inverse=True)
    vals = np.array([math.pow(float(u), -0.5) for u in uniq], dtype=np.float64)
    return vals[inv].reshape(x.shape)


def normalizer(deg):
    """gcn.h:219-221 / 471-474 / 536-539: deg==0 ? 0 : fx(pow(deg+1,-0.5))."""
    deg = np.asarray(deg, dtype=np.float64)
    return np.where(deg == 0, U64(0), fx_encode(pow_neg_half(deg + 1.0)))


# ----------------------------------------------------------------------------------------
# Two-party share arithmetic (stand-ins for the external sci:: ops)
# Every function takes both parties' shares and returns both parties' outputs; the values
# each side computes depend only on its own share, its own dealer streams and the opened
# (exchanged) values, exactly as in the HIP engine.
# ----------------------------------------------------------------------------------------
def open_hi48(c0, c1):
    """What both parties form from two opened shares of a truncation (or of the ReLU's masked product): the top 48 bits of each,
    added mod 2^48."""
    with np.errstate(over="ignore"):
        return ((np.asarray(c0, dtype=U64) >> U64(SCALER_BIT_LENGTH)) + (np.asarray(c1, dtype=U64) >> U64(SCALER_BIT_LENGTH))) & U64(0xFFFFFFFFFFFF)


def trunc_pair(x0, x1, key_of):
    """Dealer-assisted truncation by f bits (T2, bounded mask). key_of(slot)->stream key."""
    shape = x0.shape
    with np.errstate(over="ignore"):
        r = prng_shape(key_of(SL_R), shape) & U64(TRUNC_MASK)
        r0 = prng_shape(key_of(SL_R0), shape)
        r1 = r - r0
        rp = r >> U64(SCALER_BIT_LENGTH)
        rp0 = prng_shape(key_of(SL_RP0), shape)
        rp1 = rp - rp0
        c0 = x0 + r0 + U64(TRUNC_OFFSET)
        c1 = x1 + r1
        # opened: both parties add the TOP 48 BITS of the two shares mod 2^48 (the carry out of the low 16 bits is dropped, so 6 bytes
        # of an opened share are all that ever travels): (c >> f) - {0, 1}
        o0 = open_hi48(c0, c1) - U64(TRUNC_OFFSET >> SCALER_BIT_LENGTH) - rp0
        o1 = U64(0) - rp1
    return o0, o1


def ring_matmul(a, b):
    with np.errstate(over="ignore"):
        return np.matmul(a.astype(U64), b.astype(U64))


def beaver_gemm_pair(x0, x1, w0, w1, key_of, a_of_transposed=False):
    """sci::twoPartyGCNMatMul stand-in (gcn.h:233,665,671,710): Z = X.W mod 2^64 (no truncation).
    a_of_transposed: X is the transpose of a tensor whose Beaver mask was dealt for its untransposed use ([K x M] row-major
    streams); that mask - and therefore that opening - is reused."""
    M, K = x0.shape
    K2, N = w0.shape
    assert K == K2
    with np.errstate(over="ignore"):
        if a_of_transposed:
            a0 = gemm_mask_shape(key_of(SL_A0), (K, M)).T.copy(); a1 = gemm_mask_shape(key_of(SL_A1), (K, M)).T.copy()
        else:
            a0 = gemm_mask_shape(key_of(SL_A0), (M, K)); a1 = gemm_mask_shape(key_of(SL_A1), (M, K))
        b0 = prng_shape(key_of(SL_B0), (K, N)); b1 = prng_shape(key_of(SL_B1), (K, N))
        c0 = prng_shape(key_of(SL_C0), (M, N))
        c1 = ring_matmul(a0 + a1, b0 + b1) - c0               # dealer, offline
        e = (x0 - a0) + (x1 - a1)                             # opened
        f = (w0 - b0) + (w1 - b1)                             # opened
        z0 = ring_matmul(e, b0) + ring_matmul(a0, f) + c0
        z1 = ring_matmul(e, f) + ring_matmul(e, b1) + ring_matmul(a1, f) + c1
    return z0, z1


def beaver_rowscale_pair(v0, v1, s0, s1, key_of):
    """sci::twoPartyGCNVectorScale stand-in (gcn.h:247,476): Z[r,:] = V[r,:]*s[r] (no truncation).
    The scale is additively shared; the owner passes the real normaliser, the server zeros
    (ss_...h:739 vs :985-989)."""
    n, F = v0.shape
    with np.errstate(over="ignore"):
        a0 = prng_shape(key_of(SL_A0), (n, F)); a1 = prng_shape(key_of(SL_A1), (n, F))
        b0 = prng(key_of(SL_B0), n)[:, None]; b1 = prng(key_of(SL_B1), n)[:, None]
        c0 = prng_shape(key_of(SL_C0), (n, F))
        c1 = (a0 + a1) * (b0 + b1) - c0
        e = (v0 - a0) + (v1 - a1)
        g = (s0[:, None] - b0) + (s1[:, None] - b1)
        z0 = e * b0 + a0 * g + c0
        z1 = e * g + e * b1 + a1 * g + c1
    return z0, z1


def relu_pair(z0, z1, key_of):
    """sci::twoPartyGCNRelu stand-in (gcn.h:549): masked-sign ReLU (R1). Opens w = z*t with a
    dealer-shared random t in [2^17,2^20); the sign of z becomes public (stated leak)."""
    shape = z0.shape
    with np.errstate(over="ignore"):
        t = (prng_shape(key_of(SL_T), shape) & U64(0xFFFFF)) | U64(RELU_T_MIN)     # t in [2^17, 2^20)
        t0 = prng_shape(key_of(SL_T0), shape)
        t1 = t - t0
        a0 = prng_shape(key_of(SL_A0), shape); a1 = prng_shape(key_of(SL_A1), shape)
        b0 = prng_shape(key_of(SL_B0), shape); b1 = prng_shape(key_of(SL_B1), shape)
        c0 = prng_shape(key_of(SL_C0), shape)
        c1 = (a0 + a1) * (b0 + b1) - c0
        e = (z0 - a0) + (z1 - a1)
        g = (t0 - b0) + (t1 - b1)
        w0 = e * b0 + a0 * g + c0
        w1 = e * g + e * b1 + a1 * g + c1
        pos = (open_hi48(w0, w1) << U64(16)).astype(np.int64) > 0     # opened, read through its top 48 bits like a truncation
    h0 = np.where(pos, z0, U64(0))
    h1 = np.where(pos, z1, U64(0))
    return h0, h1, pos


EXP2_COEF = [1073741765, -744256846, 257890763, -59377501, 9890102, -1017428]   # 2^-x, Q30
LOG2E_Q16 = 94548


def int_softmax(z):
    """Integer-only row softmax on fixed-point logits (u64 two's complement) -> Q16 probabilities.
    Stand-in for the external softmax inside twoPartyGCNForwardNNPredictionWithoutWeight
    (gcn.h:578); all-integer so CPU and GPU agree bit for bit."""
    zs = z.astype(np.int64)
    m = zs.max(axis=1, keepdims=True)
    d = (m - zs).astype(np.int64)                              # >= 0, Q16
    big = d >= (32 << 16)
    d = np.where(big, 0, d)
    u = (d * LOG2E_Q16) >> 16
    ip = u >> 16
    fr = u & 0xFFFF
    acc = np.full(fr.shape, EXP2_COEF[5], dtype=np.int64)
    for c in EXP2_COEF[4::-1]:
        acc = c + ((acc * fr) >> 16)
    e = np.where(big, 0, acc >> ip).astype(np.int64)
    S = e.sum(axis=1, keepdims=True)
    p = ((e << 16) + (S >> 1)) // S
    return p.astype(U64)


def softmax_pair(z0, z1, labels, train_rows, key_of):
    """twoPartyGCNForwardNNPredictionWithoutWeight + getPlainShareVecVec stand-in (gcn.h:578-604).
    z is opened to the owner (the reference reveals p to the owner, gcn.h:603-604), p re-shared
    with a dealer mask; returns (p0,p1,(p-y)0,(p-y)1,plainP)."""
    with np.errstate(over="ignore"):
        z = z0 + z1
        pfx = int_softmax(z)
        rho = prng_shape(key_of(SL_RHO), z.shape)
        p0 = pfx - rho
        p1 = rho.copy()
        y = np.zeros(z.shape, dtype=U64)
        y[np.arange(z.shape[0]), labels] = U64(FX_ONE)
        d0 = p0 - y
        d1 = p1.copy()
    d0[train_rows:] = 0                                        # gcn.h:639-641
    d1[train_rows:] = 0
    return p0, p1, d0, d1, pfx.astype(np.float64) / FX_ONE


def const_scale_trunc_pair(x0, x1, cfx, key_of):
    """twoPartyGCNMatrixScale stand-in (gcn.h:676,764): share * public constant, then truncate."""
    with np.errstate(over="ignore"):
        return trunc_pair(x0 * U64(cfx), x1 * U64(cfx), key_of)


# ----------------------------------------------------------------------------------------
# Oblivious mapper (OEP) and prefix aggregation (OGA) as plain index arithmetic on one share
# ----------------------------------------------------------------------------------------
def oep_loop(src_pos, dst_pos, src, allow_missing=False):
    """client/server_oblivious_mapper_online stand-in (ss_...h:752,760,818,848), literal form:
    dst[r] = src[last index q with src_pos[q] == dst_pos[r]]; zero row if absent and allowed."""
    last = {}
    for q, p in enumerate(src_pos):
        last[int(p)] = q
    out = np.zeros((len(dst_pos), src.shape[1]), dtype=U64)
    for r, p in enumerate(dst_pos):
        q = last.get(int(p))
        if q is None:
            if not allow_missing:
                raise KeyError("oep: position %d missing in source" % int(p))
            continue
        out[r] = src[q]
    return out


def oep(src_pos, dst_pos, src, allow_missing=False):
    """Vectorised oep_loop (same result; tests/test_oracle_cpu.py checks the two against each other)."""
    sp = np.asarray(src_pos, dtype=np.int64); dp = np.asarray(dst_pos, dtype=np.int64)
    out = np.zeros((len(dp), src.shape[1]), dtype=U64)
    if len(dp) == 0:
        return out
    if len(sp) == 0:
        if not allow_missing:
            raise KeyError("oep: empty source")
        return out
    order = np.argsort(sp, kind="stable")
    ss = sp[order]
    j = np.searchsorted(ss, dp, side="right") - 1          # rightmost equal element = last occurrence
    ok = (j >= 0) & (ss[np.maximum(j, 0)] == dp)
    if not allow_missing and not ok.all():
        raise KeyError("oep: position missing in source")
    out[ok] = src[order[j[ok]]]
    return out


def prefix_network_aggregate_loop(pos, svv):
    """prefix_network_aggregate(..., ADD_AGG, ...) stand-in (gcn.h:328-335), literal form: inclusive
    prefix sum inside each run of equal consecutive pos; the run total sits at the run's last element."""
    out = svv.copy()
    with np.errstate(over="ignore"):
        for q in range(1, len(pos)):
            if pos[q] == pos[q - 1]:
                out[q] = out[q] + out[q - 1]
    return out


def prefix_network_aggregate(pos, svv):
    """Vectorised prefix_network_aggregate_loop (segmented cumulative sum mod 2^64)."""
    ps = np.asarray(pos, dtype=np.int64)
    n = len(ps)
    if n == 0:
        return svv.copy()
    with np.errstate(over="ignore"):
        cs = np.cumsum(svv, axis=0, dtype=U64)
        start = np.ones(n, dtype=bool)
        start[1:] = ps[1:] != ps[:-1]
        sidx = np.maximum.accumulate(np.where(start, np.arange(n), 0))     # index of each element's run start
        base = np.zeros_like(cs)
        has = sidx > 0
        base[has] = cs[sidx[has] - 1]
        return cs - base


# ----------------------------------------------------------------------------------------
# libc rand() Glorot init (gcn.h:838-852)
# ----------------------------------------------------------------------------------------
_libc = ctypes.CDLL(None)
_libc.rand.restype = ctypes.c_int
RAND_MAX = 2147483647


def init_weight(dim0, dim1):
    _libc.srand(42)
    limit = math.sqrt(6.0 / (dim0 + dim1))
    w = np.empty((dim0, dim1), dtype=np.float64)
    for i in range(dim0):
        for j in range(dim1):
            w[i, j] = float(_libc.rand()) / RAND_MAX * 2 * limit - limit
    return w


# ----------------------------------------------------------------------------------------
# Graph preprocessing (ss_...h:279-534, -r 1 "no dummy edges" mode)
# ----------------------------------------------------------------------------------------
class PartyState:
    """Mirror of GraphSummary (ss_...h:24-58) for one party."""
    pass


def preprocess_party(P, k, src, dst, part):
    """src,dst: directed edge list (file order); part: dict/array vid->tid. Returns PartyState with
    the index arrays of onPreprocessClient for party P (before the pos-vec exchange)."""
    gs = PartyState()
    tid = part
    V = len(part)
    own = [v for v in range(V) if tid[v] == P]
    in_deg = {v: 0 for v in own}
    out_deg = {v: 0 for v in own}
    border = {v: False for v in own}
    own_edges = []
    for s, d in zip(src, dst):
        s = int(s); d = int(d)
        if tid[s] == P:                               # graph_io_util.h:170-172, graph.h:607-633
            own_edges.append((s, d))
            out_deg[s] += 1
            if tid[d] == P:
                in_deg[d] += 1
            else:
                border[s] = True
        elif tid[d] == P:                             # graph_io_util.h:173-175
            in_deg[d] += 1
    own_edges.sort()                                  # Edge::lessFunc (graph.h:474-477)
    gs.true_in_deg = dict(in_deg)                     # what onAlgoKernelStart sees (ss_...h:177)
    srcvv = {v: [] for v in own}
    dummy = {v: [] for v in own}
    msrcvv = {}
    for s, d in own_edges:                            # ss_...h:295-314
        if tid[d] == P:
            srcvv[d].append(s); dummy[d].append(False)
        else:
            msrcvv.setdefault(d, []).append(s)
    for v in own:                                     # ss_...h:411-418
        if len(srcvv[v]) == 0:
            srcvv[v].append(v); dummy[v].append(True)
            in_deg[v] += 1; out_deg[v] += 1
    id_vecs = [[] for _ in range(k)]
    id_vecs[P] = sorted(own)
    for mv in msrcvv:
        id_vecs[tid[mv]].append(mv)
    for i in range(k):
        id_vecs[i].sort()                             # ss_...h:462-464
    gs.localVertexPos = []; gs.isLocalVertexBorder = []; gs.localVertexInDeg = []
    gs.updateSrcVertexPos = [[] for _ in range(k)]
    gs.updateDstVertexPos = [[] for _ in range(k)]
    gs.updateSrcOutDeg = [[] for _ in range(k)]
    gs.updateDstInDeg = [[] for _ in range(k)]
    gs.isGatherDstVertexDummy = [[] for _ in range(k)]
    gs.reorderedIndex = {}
    for i in range(k):                                # ss_...h:467-504
        for dst_id in id_vecs[i]:
            if i == P:
                gs.reorderedIndex[dst_id] = len(gs.localVertexPos)
                gs.localVertexPos.append(dst_id)
                gs.isLocalVertexBorder.append(border[dst_id])
                gs.localVertexInDeg.append(in_deg[dst_id])
                cur = srcvv[dst_id]
                gs.updateSrcVertexPos[i].extend(cur)
                gs.updateDstVertexPos[i].extend([dst_id] * len(cur))
                gs.updateSrcOutDeg[i].extend(out_deg[x] for x in cur)
                gs.updateDstInDeg[i].extend([in_deg[dst_id]] * len(cur))
                gs.isGatherDstVertexDummy[i].append(dummy[dst_id][0])
            else:
                cur = msrcvv[dst_id]
                gs.updateSrcVertexPos[i].extend(cur)
                gs.updateDstVertexPos[i].extend([dst_id] * len(cur))
                gs.updateSrcOutDeg[i].extend(out_deg[x] for x in cur)
                gs.updateDstInDeg[i].extend([0] * len(cur))
    gs.in_deg_after = in_deg
    return gs


def exchange_pos_vecs(states, k):
    """cs.sendPosVec / recvPosVec (ss_...h:507-534)."""
    for P in range(k):
        gs = states[P]
        gs.remoteMirrorVertexPos = [[] for _ in range(k)]
        gs.remoteUpdateDstInDeg = [[] for _ in range(k)]
        n = len(gs.localVertexPos)
        for i in range(k):
            if i == P:
                continue
            gs.remoteMirrorVertexPos[i] = list(states[i].updateDstVertexPos[P])
            gs.isGatherDstVertexDummy[i] = [True] * n
            for v in gs.remoteMirrorVertexPos[i]:
                gs.remoteUpdateDstInDeg[i].append(gs.in_deg_after[v])
                gs.isGatherDstVertexDummy[i][gs.reorderedIndex[v]] = False


# ----------------------------------------------------------------------------------------
# The engine restatement
# ----------------------------------------------------------------------------------------
class GnnParam:
    def __init__(self, num_layers=2, num_labels=7, input_dim=1433, hidden_dim=16, num_samples=2708,
                 num_edges=0, learning_rate=0.5, train_ratio=0.2, val_ratio=0.2, test_ratio=0.6):
        self.num_layers = num_layers; self.num_labels = num_labels; self.input_dim = input_dim
        self.hidden_dim = hidden_dim; self.num_samples = num_samples; self.num_edges = num_edges
        self.learning_rate = learning_rate; self.train_ratio = train_ratio
        self.val_ratio = val_ratio; self.test_ratio = test_ratio

    @staticmethod
    def read_config(path):
        """GNNParam::readConfig (include/task/task.h:106-169): 'key : value' tokens."""
        g = GnnParam()
        toks = open(path).read().split()
        i = 0
        while i + 2 < len(toks):
            key, colon = toks[i], toks[i + 1]
            if colon != ":":
                break
            val = toks[i + 2]
            if key in ("num_layers", "num_labels", "input_dim", "hidden_dim", "num_samples", "num_edges"):
                setattr(g, key, int(val))
            elif key in ("learning_rate", "train_ratio", "val_ratio", "test_ratio"):
                setattr(g, key, float(val))
            else:
                break
            i += 3
        return g


class OracleEngine:
    """Sequential emulation of k CoGNN parties running gcn-optimize / gcn-inference-optimize."""
    # the weight-gradient products reuse the Beaver mask of the forward product's left operand, indexed in that tensor's storage order
    # (DESIGN.md 3.5); a caller that deals a fresh mask for the transposed operand (the sci:: shim: tests/shim_util.py) sets it False
    WGRAD_MASK_REUSED = True

    def __init__(self, k, src, dst, part, features, labels, param, seed=0xC06A11, variant="optimize-gcn",
                 weights=None, renew_feature_mask=False):
        """renew_feature_mask: the Beaver mask of the (constant) feature operand is dealt anew in every epoch - what the engine's
        recorded epochs do (COGNN_OPT_GRAPH_EPOCHS: kernel arguments may not depend on the epoch, so the mask's key carries the
        epoch salt like every other stream) - instead of once.  The product is the same; its two SHARES, and with them the carry the
        48-bit truncation opening drops (open_hi48), are not: the two forms differ by single LSBs."""
        self.k = k; self.param = param; self.seed = seed; self.variant = variant
        self.renew_feature_mask = renew_feature_mask
        self.part = [int(t) for t in part]
        self.states = [preprocess_party(P, k, src, dst, self.part) for P in range(k)]
        exchange_pos_vecs(self.states, k)
        self.metrics = []
        self._start(features, labels, weights)

    # -- helpers ---------------------------------------------------------------------------
    def co(self, P):
        return (P + 1) % self.k

    def key_of(self, owner, it, op):
        """The stream key of GAS iteration `it`: the position inside the epoch goes through the key derivation, the epoch number
        is added to the key as epoch * GAMMA (the epoch salt, cognn_spec.h / DESIGN.md §3.2)."""
        ep = self.epoch_len()
        return lambda slot: (stream_key(self.seed, owner, it % ep, op, slot) + (it // ep) * GAMMA) & MASK64

    def key_of_feature_gemm(self, owner, it):
        """Layer-0 PreScatter product X.W0: X (the input features) is the same tensor in every epoch, so its Beaver mask A
        is dealt once (iteration 0) and E = X - A is opened once; W0's mask B and the product share C stay per-iteration
        (fixed-operand mask reuse)."""
        return lambda slot: (self._feature_mask_key(owner, it, slot) if slot in (SL_A0, SL_A1)
                             else self.key_of(owner, it, OP_PS_GEMM)(slot))

    def _feature_mask_key(self, owner, it, slot):
        salt = (it // self.epoch_len()) * GAMMA if self.renew_feature_mask else 0
        return (stream_key(self.seed, owner, 0, OP_PS_GEMM, slot) + salt) & MASK64

    def key_of_feature_wgrad(self, owner, it):
        """Layer-0 weight gradient X^T.g (gcn.h:710): the left operand is the same feature tensor, transposed, so it keeps the
        A mask of key_of_feature_gemm (indexed in storage order); B and C come from this iteration's OP_AP_GEMM streams."""
        return lambda slot: (self._feature_mask_key(owner, it, slot) if slot in (SL_A0, SL_A1)
                             else self.key_of(owner, it, OP_AP_GEMM)(slot))

    def key_of_hidden_wgrad(self, owner, it, it_fwd):
        return lambda slot: (self.key_of(owner, it_fwd, OP_PS_GEMM)(slot) if slot in (SL_A0, SL_A1)
                             else self.key_of(owner, it, OP_AP_GEMM)(slot))

    # -- onAlgoKernelStart (gcn.h:854-887) + share distribution (ss_...h:205-232) ------------
    def _start(self, features, labels, weights):
        g = self.param; k = self.k
        if weights is None:
            self.plainWeight = [init_weight(g.input_dim, g.hidden_dim), init_weight(g.hidden_dim, g.num_labels)]
        else:
            self.plainWeight = [np.array(w, dtype=np.float64) for w in weights]
        for P in range(k):
            gs = self.states[P]
            vids = gs.localVertexPos
            n = len(vids)
            feat = np.asarray(features)[vids].astype(np.float64)
            tdeg = np.array([gs.true_in_deg[v] for v in vids], dtype=np.float64)
            feat = feat * pow_neg_half(tdeg + 1.0)[:, None]          # normalizeFeatureVec, gcn.h:819-835
            gs.labels = np.asarray(labels)[vids].astype(np.int64)
            gs.plainFeat = feat
            fx = fx_encode(feat)
            with np.errstate(over="ignore"):
                s1 = prng_shape(stream_key(self.seed, P, 0, OP_SHARE_FEAT, 0), fx.shape)
                s0 = fx - s1
            gs.localVertexSvv = s0
            gs.featShare1 = s1
            gs.localWeight = []; gs.remoteWeightGen = []
            for l, w in enumerate(self.plainWeight):
                wfx = fx_encode(w)
                with np.errstate(over="ignore"):
                    w1 = prng_shape(stream_key(self.seed, P, 0, OP_SHARE_W, l), wfx.shape)
                    w0 = wfx - w1
                gs.localWeight.append(w0); gs.remoteWeightGen.append(w1)
            gs.localInter = [dict() for _ in range(g.num_layers)]
            gs.remoteInter = [dict() for _ in range(g.num_layers)]
        for P in range(k):
            gs = self.states[P]
            gs.remoteVertexSvvs = [None] * k
            for i in range(k):
                if i != P:
                    gs.remoteVertexSvvs[i] = self.states[i].featShare1.copy()   # ss_...h:209-223
            gs.localVertexSvvBackup = gs.localVertexSvv.copy()                  # ss_...h:226-227
            gs.remoteVertexSvvsBackup = [None if x is None else x.copy() for x in gs.remoteVertexSvvs]
            prev = (P + k - 1) % k
            gs.remoteWeight = [w.copy() for w in self.states[prev].remoteWeightGen]   # ss_...h:231-232
            gs.localUpdateSvvs = [None] * k
            gs.remoteUpdateSvvs = [None] * k

    # -- schedule constants (gcn.h:893-948) ------------------------------------------------
    def epoch_len(self):
        return 3 * self.param.num_layers

    def fwd_layers(self):
        return self.param.num_layers

    def co_forward_layer(self, it):
        e = it % self.epoch_len(); f = self.fwd_layers()
        return e if e < f else f - 1 - ((e - f) // 2)

    # -- PreScatterComp (gcn.h:198-255), both roles of the (owner P, co c) pair ---------------
    def _prescatter_pair(self, P, it):
        c = self.co(P); gsP = self.states[P]; gsC = self.states[c]
        e = it % self.epoch_len(); fwd = e < self.fwd_layers(); layer = self.co_forward_layer(it)
        xA = gsP.localVertexSvv; xB = gsC.remoteVertexSvvs[P]
        sA = normalizer(gsP.localVertexInDeg); sB = np.zeros(len(sA), dtype=U64)   # ss_...h:739, 985-989
        if fwd:
            gsP.localInter[layer]["h_t"] = xA.copy()          # stored untransposed; used as X^T below
            gsC.remoteInter[layer]["h_t"] = xB.copy()
            zA, zB = beaver_gemm_pair(xA, xB, gsP.localWeight[layer], gsC.remoteWeight[layer],
                                      self.key_of_feature_gemm(P, it) if layer == 0 else self.key_of(P, it, OP_PS_GEMM))
            xA, xB = trunc_pair(zA, zB, self.key_of(P, it, OP_PS_GEMM_TRUNC))
        if e != 0:
            zA, zB = beaver_rowscale_pair(xA, xB, sA, sB, self.key_of(P, it, OP_PS_SCALE))
            xA, xB = trunc_pair(zA, zB, self.key_of(P, it, OP_PS_SCALE_TRUNC))
        gsP.localVertexSvv = xA; gsC.remoteVertexSvvs[P] = xB
        for j in range(self.k):                               # ss_...h:997-1002 / :982
            if j != P and j != c:
                self.states[j].remoteVertexSvvs[P] = xB.copy()

    # -- message passing for one owner P (ss_...h:748-856 client, :1005-1080 server) ----------
    def _message_passing(self, P):
        k = self.k; c = self.co(P); gsP = self.states[P]
        lpos = gsP.localVertexPos
        # local edges: pair (P client, c server)
        for side, gsX, x in (("A", gsP, gsP.localVertexSvv), ("B", self.states[c], self.states[c].remoteVertexSvvs[P])):
            upd = oep(lpos, gsP.updateSrcVertexPos[P], x)                     # ss_...h:752 / 1011
            dup = upd.copy()                                                  # ScatterComp, gcn.h:300
            dup = prefix_network_aggregate(gsP.updateDstVertexPos[P], dup)    # UpdatePreMergeComp
            ext = oep(gsP.updateDstVertexPos[P], lpos, dup)                   # ss_...h:818 / 1057
            if side == "A":
                gsP.localUpdateSvvs[P] = ext
            else:
                gsX.remoteUpdateSvvs[c] = ext                                 # remoteUpdateSvvs[tileIndex]
        # edges P -> i: pair (P client with its own share, i server with its replica)
        for i in range(k):
            if i == P:
                continue
            gsI = self.states[i]
            updA = oep(lpos, gsP.updateSrcVertexPos[i], gsP.localVertexSvv)   # ss_...h:760
            updB = oep(lpos, gsP.updateSrcVertexPos[i], gsI.remoteVertexSvvs[P])   # ss_...h:1016
            gsP.remoteUpdateSvvs[i] = prefix_network_aggregate(gsP.updateDstVertexPos[i], updA)   # :827,835
            gsI.localUpdateSvvs[P] = prefix_network_aggregate(gsP.updateDstVertexPos[i], updB)    # :1063,1067

    def _extend_updates(self, P):
        """OEP remoteMirrorVertexPos[i] -> localVertexPos with missing allowed (ss_...h:847-854, 1074-1080)."""
        gsP = self.states[P]
        for i in range(self.k):
            if i == P:
                continue
            gsI = self.states[i]
            mpos = gsP.remoteMirrorVertexPos[i]
            gsP.localUpdateSvvs[i] = oep(mpos, gsP.localVertexPos, gsP.localUpdateSvvs[i], True)
            gsI.remoteUpdateSvvs[P] = oep(mpos, gsP.localVertexPos, gsI.remoteUpdateSvvs[P], True)

    # -- GatherComp over all source parties (gcn.h:375-494; ss_...h:866-880, 1087-1135) --------
    def _gather_pair(self, P, it):
        k = self.k; c = self.co(P); gsP = self.states[P]; gsC = self.states[c]
        vA = gsP.localVertexSvv; vB = gsC.remoteVertexSvvs[P]
        remote = [None] * k                                   # ss_...h:1089-1100
        remote[c] = gsC.remoteUpdateSvvs[P] if c != P else None
        remote[P] = gsC.remoteUpdateSvvs[c]
        for j in range(k):
            if j != c and j != P:
                remote[j] = self.states[j].remoteUpdateSvvs[P]
        with np.errstate(over="ignore"):
            for j in range(k):
                cond = ~np.array(gsP.isGatherDstVertexDummy[j], dtype=bool)   # gcn.h:454-455 (owner's mask)
                vA = vA + np.where(cond[:, None], gsP.localUpdateSvvs[j], U64(0))
                vB = vB + np.where(cond[:, None], remote[j], U64(0))
        if (it + 1) % self.epoch_len() != 0:                  # gcn.h:470
            sA = normalizer(gsP.localVertexInDeg); sB = np.zeros(len(sA), dtype=U64)
            zA, zB = beaver_rowscale_pair(vA, vB, sA, sB, self.key_of(P, it, OP_GA_SCALE))
            vA, vB = trunc_pair(zA, zB, self.key_of(P, it, OP_GA_SCALE_TRUNC))
        gsP.localVertexSvv = vA; gsC.remoteVertexSvvs[P] = vB

    # -- ApplyComp (gcn.h:515-811), both roles ----------------------------------------------
    def _apply_pair(self, P, it):
        g = self.param; c = self.co(P); gsP = self.states[P]; gsC = self.states[c]
        e = it % self.epoch_len(); f = self.fwd_layers(); fwd = e < f; layer = self.co_forward_layer(it)
        inA = gsP.localVertexSvv; inB = gsC.remoteVertexSvvs[P]
        LI = gsP.localInter[layer]; RI = gsC.remoteInter[layer]
        n = inA.shape[0]
        train = int(n * g.train_ratio); val = int(n * g.val_ratio)
        if fwd:
            if e != f - 1:                                    # GCN_FORWARD_NN, gcn.h:546-558
                LI["z"] = inA.copy(); RI["z"] = inB.copy()
                outA, outB, pos = relu_pair(inA, inB, self.key_of(P, it, OP_AP_RELU))
                LI["relu_mask"] = pos; RI["relu_mask"] = pos
            else:                                             # GCN_FORWARD_PREDICTION, gcn.h:559-643
                LI["z"] = inA.copy(); RI["z"] = inB.copy()
                p0, p1, outA, outB, plainP = softmax_pair(inA, inB, gsP.labels, train,
                                                          self.key_of(P, it, OP_AP_SOFTMAX))
                LI["p"] = p0; RI["p"] = p1
                self.metrics.append(self._metrics(P, it, plainP, train, val))
        else:
            first_of_two = ((e - f) % 2 == 0)
            WA = gsP.localWeight[layer]; WB = gsC.remoteWeight[layer]
            if first_of_two:
                if layer == f - 1:                            # gcn.h:664-669
                    zA, zB = beaver_gemm_pair(inA, inB, WA.T.copy(), WB.T.copy(), self.key_of(P, it, OP_AP_GEMM))
                    gA, gB = trunc_pair(zA, zB, self.key_of(P, it, OP_AP_GEMM_TRUNC))
                    LI["g"] = gA; RI["g"] = gB
                    outA, outB = inA.copy(), inB.copy()
                else:                                         # twoPartyGCNBackwardNNWithoutAH, gcn.h:702-708
                    pos = LI["relu_mask"]
                    outA = np.where(pos, inA, U64(0)); outB = np.where(pos, inB, U64(0))
                    LI["g"] = None; RI["g"] = None             # g' skipped for the first layer
            else:                                             # gcn.h:671-684 / 710-736
                hA = LI["h_t"].T.copy(); hB = RI["h_t"].T.copy()
                if layer == 0:
                    # h_t is the transposed input-feature tensor (gcn.h:230-231): its mask and opening are those of the
                    # layer-0 forward product (dealt once, iteration 0); B and C are fresh
                    zA, zB = beaver_gemm_pair(hA, hB, inA, inB, self.key_of_feature_wgrad(P, it), a_of_transposed=self.WGRAD_MASK_REUSED)
                else:
                    # h_t is the transposed hidden activation of this epoch: mask and opening of the layer-1 forward product
                    # (GAS iteration it - e + layer) are reused the same way
                    zA, zB = beaver_gemm_pair(hA, hB, inA, inB, self.key_of_hidden_wgrad(P, it, it - e + layer), a_of_transposed=self.WGRAD_MASK_REUSED)
                dA, dB = trunc_pair(zA, zB, self.key_of(P, it, OP_AP_GEMM_TRUNC))
                gscale = fx_encode_trunc(1.0 / train) if train > 0 else U64(0)
                dA, dB = const_scale_trunc_pair(dA, dB, gscale, self.key_of(P, it, OP_AP_GSCALE_TRUNC))
                lr = fx_encode_trunc(g.learning_rate)
                uA, uB = const_scale_trunc_pair(dA, dB, lr, self.key_of(P, it, OP_AP_LR_TRUNC))
                with np.errstate(over="ignore"):
                    gsP.localWeight[layer] = WA - uA           # twoPartyGCNApplyGradient
                    gsC.remoteWeight[layer] = WB - uB
                if self.variant == "optimize-gcn-inference":   # inference/gcn.h:680-681,732-733
                    ws = fx_encode_trunc(1.0 / self.k)
                    a, b = const_scale_trunc_pair(gsP.localWeight[layer], gsC.remoteWeight[layer], ws,
                                                  self.key_of(P, it, OP_WAVG_TRUNC))
                    gsP.localWeight[layer] = a; gsC.remoteWeight[layer] = b
                LI["d"] = dA; RI["d"] = dB
                outA = LI["g"] if LI["g"] is not None else np.zeros((n, 0), dtype=U64)
                outB = RI["g"] if RI["g"] is not None else np.zeros((n, 0), dtype=U64)
        gsP.localVertexSvv = outA; gsC.remoteVertexSvvs[P] = outB
        for j in range(self.k):                               # ss_...h:967-972, 1153-1158
            if j != P and j != c:
                self.states[j].remoteVertexSvvs[P] = outB.copy()

    def _weight_average(self, it):
        """gcn.h:747-802, executed by every party's client thread after a weight update."""
        k = self.k; layer = self.co_forward_layer(it)
        st = self.states
        with np.errstate(over="ignore"):
            sum1 = st[1].localWeight[layer].copy()            # party 1: gcn.h:753-762
            for i in range(2, k):
                sum1 = sum1 + st[i].localWeight[layer]
            sum1 = sum1 + st[1].remoteWeight[layer]
            sum0 = st[0].localWeight[layer].copy()            # party 0
            for i in range(2, k):
                sum0 = sum0 + st[i].remoteWeight[layer]
            sum0 = sum0 + st[0].remoteWeight[layer]
        if self.variant == "optimize-gcn":                    # gcn.h:763-764 (absent in inference variant)
            ws = fx_encode_trunc(1.0 / k)
            sum0, sum1 = const_scale_trunc_pair(sum0, sum1, ws, self.key_of(OWNER_WAVG, it, OP_WAVG_TRUNC))
        st[0].localWeight[layer] = sum0.copy(); st[0].remoteWeight[layer] = sum0.copy()
        st[1].localWeight[layer] = sum1.copy(); st[1].remoteWeight[layer] = sum1.copy()
        for i in range(2, k):                                 # gcn.h:767-778
            st[i].localWeight[layer] = sum1.copy()
            st[i].remoteWeight[layer] = sum0.copy()

    def _metrics(self, P, it, plainP, train, val):
        """gcn.h:611-632."""
        gs = self.states[P]
        n = plainP.shape[0]
        p = np.where(plainP == 0, 0.001, plainP)
        y = gs.labels
        loss = float(-np.log(p[np.arange(n), y]).sum() / max(n, 1))
        pred = p.argmax(axis=1)
        ok = pred == y
        border = np.array(gs.isLocalVertexBorder, dtype=bool)

        def acc(sel):
            return float(ok[sel].mean()) if sel.any() else 0.0
        idx = np.arange(n)
        tr = idx < train; te = idx >= train + val
        return dict(party=P, iter=it, loss=loss, full=acc(np.ones(n, bool)), train=acc(tr),
                    border_train=acc(tr & border), test=acc(te), border_test=acc(te & border),
                    correct=int(ok.sum()), n=n, n_border=int(border.sum()))

    # -- one GAS iteration for all parties (ss_...h:680-910 + 912-1189) -------------------------
    def iteration(self, it):
        k = self.k; ep = self.epoch_len(); f = self.fwd_layers(); e = it % ep
        if e == 0:                                            # ss_...h:695, 938
            for P in range(k):
                gs = self.states[P]
                gs.localVertexSvv = gs.localVertexSvvBackup.copy()
                gs.remoteVertexSvvs = [None if x is None else x.copy() for x in gs.remoteVertexSvvsBackup]
        if e != 0 and e % f == 0:                             # apply-only, ss_...h:709-732, 941-979
            for P in range(k):
                self._apply_pair(P, it)
            return
        for P in range(k):
            self._prescatter_pair(P, it)
        for P in range(k):
            self._message_passing(P)
        for P in range(k):
            self._extend_updates(P)
        for P in range(k):
            self._gather_pair(P, it)
        for P in range(k):
            self._apply_pair(P, it)
        if e >= f and (e - f) % 2 == 1:
            self._weight_average(it)

    def run(self, iters):
        for it in range(iters):
            self.iteration(it)

    # -- views ------------------------------------------------------------------------------
    def shares(self, P):
        """(owner share, co-party share) of party P's current vertex tensor."""
        return self.states[P].localVertexSvv, self.states[self.co(P)].remoteVertexSvvs[P]

    def reconstruct(self, P):
        a, b = self.shares(P)
        with np.errstate(over="ignore"):
            return fx_decode(a + b)

    def weight(self, P, layer):
        with np.errstate(over="ignore"):
            return fx_decode(self.states[P].localWeight[layer] + self.states[self.co(P)].remoteWeight[layer])


# ----------------------------------------------------------------------------------------
# float64 plaintext GCN following the same schedule (reconstruction tests)
# ----------------------------------------------------------------------------------------
class PlainEngine:
    """Same schedule as OracleEngine on float64 plaintext, global view. Used to check that the
    reconstructed shares track the plaintext computation within fixed-point tolerance."""

    def __init__(self, oracle):
        o = oracle; self.o = o; self.k = o.k; g = o.param
        self.W = [[w.copy() for w in o.plainWeight] for _ in range(o.k)]
        self.X0 = [o.states[P].plainFeat.copy() for P in range(o.k)]
        self.X = [x.copy() for x in self.X0]
        self.inter = [[dict() for _ in range(g.num_layers)] for _ in range(o.k)]
        self.vid_row = {}
        self.metrics = []                                     # what the client would print at every prediction layer (gcn.h:611-632)
        for P in range(o.k):
            for r, v in enumerate(o.states[P].localVertexPos):
                self.vid_row[v] = (P, r)

    def _s(self, P):
        deg = np.asarray(self.o.states[P].localVertexInDeg, dtype=np.float64)
        return np.where(deg == 0, 0.0, np.power(deg + 1.0, -0.5))

    def _adjacency(self):
        """Global (all parties' rows stacked) 0/1 adjacency of the real edges: built once from the same preprocess arrays the
        share engine uses (dummy self sources dropped), as a scipy CSR so that long runs stay fast."""
        if getattr(self, "_adj", None) is None:
            import scipy.sparse as sp
            off = np.cumsum([0] + [len(self.o.states[P].localVertexPos) for P in range(self.k)])
            rows, cols = [], []
            for P in range(self.k):
                gs = self.o.states[P]
                for i in range(self.k):
                    for s, d in zip(gs.updateSrcVertexPos[i], gs.updateDstVertexPos[i]):
                        if i == P and s == d and gs.isGatherDstVertexDummy[P][gs.reorderedIndex[d]]:
                            continue                              # dummy self source
                        (ps, rs), (pd, rd) = self.vid_row[s], self.vid_row[d]
                        rows.append(off[pd] + rd); cols.append(off[ps] + rs)
            n = int(off[-1])
            self._adj = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
            self._off = off
        return self._adj, self._off

    def _aggregate(self):
        A, off = self._adjacency()
        allx = np.vstack(self.X)
        allx = allx + A @ allx
        self.X = [allx[off[P]:off[P + 1]] for P in range(self.k)]

    def iteration(self, it):
        o = self.o; g = o.param; ep = o.epoch_len(); f = o.fwd_layers(); e = it % ep
        layer = o.co_forward_layer(it)
        if e == 0:
            self.X = [x.copy() for x in self.X0]
        apply_only = (e != 0 and e % f == 0)
        if not apply_only:
            for P in range(self.k):
                if e < f:
                    self.inter[P][layer]["h"] = self.X[P].copy()
                    self.X[P] = self.X[P] @ self.W[P][layer]
                if e != 0:
                    self.X[P] = self.X[P] * self._s(P)[:, None]
            self._aggregate()
            if (it + 1) % ep != 0:
                for P in range(self.k):
                    self.X[P] = self.X[P] * self._s(P)[:, None]
        for P in range(self.k):
            I = self.inter[P][layer]; x = self.X[P]; n = x.shape[0]
            train = int(n * g.train_ratio)
            if e < f:
                I["z"] = x.copy()
                if e != f - 1:
                    self.X[P] = np.maximum(x, 0)
                else:
                    m = x.max(axis=1, keepdims=True)
                    ex = np.exp(x - m); p = ex / ex.sum(axis=1, keepdims=True)
                    y = np.zeros_like(p); y[np.arange(n), o.states[P].labels] = 1.0
                    d = p - y; d[train:] = 0
                    I["p"] = p; self.X[P] = d
                    self.metrics.append(o._metrics(P, it, p, train, int(n * g.val_ratio)))
            else:
                first = ((e - f) % 2 == 0)
                if first:
                    if layer == f - 1:
                        I["g"] = x @ self.W[P][layer].T
                    else:
                        self.X[P] = x * (I["z"] > 0)
                        I["g"] = None
                else:
                    d = I["h"].T @ x
                    d = d * (1.0 / train if train > 0 else 0.0)
                    self.W[P][layer] = self.W[P][layer] - g.learning_rate * d
                    if o.variant == "optimize-gcn-inference":
                        self.W[P][layer] = self.W[P][layer] / self.k
                    self.X[P] = I["g"] if I["g"] is not None else np.zeros((n, 0))
        if e >= f and (e - f) % 2 == 1:
            avg = sum(self.W[P][layer] for P in range(self.k))
            if o.variant == "optimize-gcn":
                avg = avg / self.k
            for P in range(self.k):
                self.W[P][layer] = avg.copy()


# ----------------------------------------------------------------------------------------
# Synthetic inputs: symmetric random graph, vid % k partition, Bernoulli features
# ----------------------------------------------------------------------------------------
def synth_graph(num_vertices, num_undirected, seed):
    """num_undirected distinct undirected pairs without self loops, emitted in both directions
    (reference datasets are symmetric: tools/data_transform.py:40)."""
    rng = np.random.default_rng(seed)
    keys = np.empty(0, dtype=np.int64)
    while len(keys) < num_undirected:
        m = num_undirected - len(keys)
        a = rng.integers(0, num_vertices, size=m + m // 8 + 16, dtype=np.int64)
        b = rng.integers(0, num_vertices, size=m + m // 8 + 16, dtype=np.int64)
        ok = a != b
        lo = np.minimum(a, b)[ok]; hi = np.maximum(a, b)[ok]
        cand = lo * np.int64(num_vertices) + hi
        fresh = np.setdiff1d(cand, keys)                     # sorted, unique
        if len(fresh) > m:
            fresh = rng.permutation(fresh)[:m]
        keys = np.union1d(keys, fresh)
    lo = keys // num_vertices; hi = keys % num_vertices
    src = np.concatenate([lo, hi]).astype(np.int64)
    dst = np.concatenate([hi, lo]).astype(np.int64)
    return src, dst


def synth_features(num_vertices, input_dim, num_labels, seed, density=0.01):
    rng = np.random.default_rng(seed)
    feats = (rng.random((num_vertices, input_dim)) < density).astype(np.float64)
    labels = rng.integers(0, num_labels, size=num_vertices)
    return feats, labels


def synth_planted(num_vertices, num_undirected, input_dim, num_labels, seed, p_intra=0.85, p_on=0.06, p_off=0.004):
    """A LEARNABLE synthetic stand-in for a citation dataset (the Planetoid files are not available offline): every vertex
    gets a class; an edge joins two vertices of the same class with probability p_intra; the bag-of-words features are
    Bernoulli(p_on) inside the class's block of input_dim/num_labels words and Bernoulli(p_off) elsewhere.  Returns
    (src, dst, features, labels) with the edge list symmetric like synth_graph's."""
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, num_labels, size=num_vertices)
    by_class = [np.flatnonzero(labels == c) for c in range(num_labels)]
    keys = np.empty(0, dtype=np.int64)
    while len(keys) < num_undirected:
        m = num_undirected - len(keys) + 64
        a = rng.integers(0, num_vertices, size=m, dtype=np.int64)
        b = rng.integers(0, num_vertices, size=m, dtype=np.int64)
        intra = rng.random(m) < p_intra
        for c in range(num_labels):                           # redraw the second endpoint inside the first one's class
            sel = intra & (labels[a] == c)
            if sel.any() and len(by_class[c]):
                b[sel] = by_class[c][rng.integers(0, len(by_class[c]), size=int(sel.sum()))]
        ok = a != b
        lo = np.minimum(a, b)[ok]; hi = np.maximum(a, b)[ok]
        fresh = np.setdiff1d(lo * np.int64(num_vertices) + hi, keys)
        if len(fresh) > num_undirected - len(keys):
            fresh = rng.permutation(fresh)[:num_undirected - len(keys)]
        keys = np.union1d(keys, fresh)
    lo = keys // num_vertices; hi = keys % num_vertices
    src = np.concatenate([lo, hi]).astype(np.int64); dst = np.concatenate([hi, lo]).astype(np.int64)
    block = max(1, input_dim // num_labels)
    prob = np.full((num_vertices, input_dim), p_off)
    for c in range(num_labels):
        prob[np.ix_(by_class[c], np.arange(c * block, min(input_dim, (c + 1) * block)))] = p_on
    feats = (rng.random((num_vertices, input_dim)) < prob).astype(np.float64)
    return src, dst, feats, labels
