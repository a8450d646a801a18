"""cognn_pair_chain_u64: both share-holders' protocol steps in one kernel with the exchange in registers.  Every flag
combination the engine uses (and the opened-input form) against the oracle's two-party functions, bit for bit, including odd
widths (a thread's two elements fall into different rows) and odd lengths (single-element tail)."""
import ctypes

import numpy as np
import pytest

import cognn_oracle as co
from gpu_util import dev, dev_empty, host, ptr, rand_u64, U64

pytestmark = pytest.mark.gpu

TRUNC_IN, SCALE, RELU, OPENED, NO_C, OPEN_SUM = 1, 2, 4, 8, 16, 32


@pytest.fixture(scope="module")
def ctx():
    from cognn_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _keys(seed, owner, it, op):
    from cognn_amd import capi
    return capi.make_keys(seed, owner, it, op), (lambda slot: co.stream_key(seed, owner, it, op, slot))


def _expect(flags, x0, x1, c1, s0, s1, kf):
    """The oracle's per-step functions chained the way the flags say."""
    with np.errstate(over="ignore"):
        v0, v1 = x0.copy(), x1.copy()
        pos = None
        if flags & TRUNC_IN:
            if not flags & NO_C:
                v0 = v0 + co.prng_shape(kf["gemm"](co.SL_C0), v0.shape); v1 = v1 + c1
            v0, v1 = co.trunc_pair(v0, v1, kf["tin"])
        if flags & SCALE:
            if flags & OPENED:      # the inputs already are V_p - a_p: recover V_p for the oracle's function
                v0 = v0 + co.prng_shape(kf["scale"](co.SL_A0), v0.shape); v1 = v1 + co.prng_shape(kf["scale"](co.SL_A1), v1.shape)
            z0, z1 = co.beaver_rowscale_pair(v0, v1, s0, s1, kf["scale"])
            v0, v1 = co.trunc_pair(z0, z1, kf["strunc"])
        if flags & RELU:
            v0, v1, pos = co.relu_pair(v0, v1, kf["relu"])
    return v0, v1, pos


@pytest.mark.parametrize("flags", [TRUNC_IN, TRUNC_IN | NO_C, TRUNC_IN | SCALE, SCALE, SCALE | OPENED, SCALE | RELU, RELU,
                                   TRUNC_IN | SCALE | RELU, SCALE | OPENED | RELU])
@pytest.mark.parametrize("rows,F", [(257, 16), (301, 7), (64, 64), (1, 1), (33, 3)])
def test_pair_chain_matches_the_two_party_oracle(ctx, flags, rows, F):
    from cognn_amd import capi
    rng = np.random.default_rng(rows * 131 + F * 7 + flags)
    shape = (rows, F)
    # fixed-point-sized values so that truncation never meets its 2^61 bound; shares are uniform
    val = rng.integers(-(1 << 40), 1 << 40, size=shape).astype(np.int64).astype(U64)
    x1 = rand_u64(rng, shape)
    with np.errstate(over="ignore"):
        x0 = val - x1
    c1 = rand_u64(rng, shape) >> U64(30)
    s0 = co.normalizer(rng.integers(0, 9, size=rows)); s1 = np.zeros(rows, dtype=U64)
    ks = {n: _keys(5, 3, 9, op) for n, op in (("gemm", co.OP_PS_GEMM), ("tin", co.OP_PS_GEMM_TRUNC), ("scale", co.OP_GA_SCALE),
                                                ("strunc", co.OP_GA_SCALE_TRUNC), ("relu", co.OP_AP_RELU))}
    kf = {n: v[1] for n, v in ks.items()}
    ok0, ok1 = co.stream_key(1, 2, 3, 4, 0), co.stream_key(1, 2, 3, 4, 1)
    out0, out1, op0, op1 = (dev_empty(shape) for _ in range(4))
    mask = dev_empty(shape, "u8")
    c = capi.PairChain()
    c.x[0] = dev(x0).data_ptr(); c.x[1] = dev(x1).data_ptr(); c.c1 = dev(c1).data_ptr()
    c.scale[0] = dev(s0).data_ptr(); c.scale[1] = dev(s1).data_ptr()
    c.out[0] = out0.data_ptr(); c.out[1] = out1.data_ptr(); c.open[0] = op0.data_ptr(); c.open[1] = op1.data_ptr()
    c.mask = mask.data_ptr(); c.open_key[0] = ok0; c.open_key[1] = ok1
    c.gemm_keys = ks["gemm"][0]; c.trunc_in_keys = ks["tin"][0]; c.scale_keys = ks["scale"][0]
    c.scale_trunc_keys = ks["strunc"][0]; c.relu_keys = ks["relu"][0]
    c.rows = rows; c.F = F; c.flags = flags
    ctx.call("cognn_pair_chain_u64", ctypes.byref(c), 1)
    e0, e1, pos = _expect(flags, x0, x1, c1, s0, s1, kf)
    assert np.array_equal(host(out0), e0) and np.array_equal(host(out1), e1)
    with np.errstate(over="ignore"):
        assert np.array_equal(host(op0), e0 - co.prng_shape(ok0, shape))
        assert np.array_equal(host(op1), e1 - co.prng_shape(ok1, shape))
    # the dealt form (COGNN_OPT_DEALER_STREAMS): the same chain with its dealer values read from a slab instead of regenerated
    if not flags & OPENED:
        lib = capi.load()
        slots = lib.cognn_pair_chain_dealt_slots(flags, 1)
        assert slots == (5 if flags & TRUNC_IN else 0) + (8 if flags & SCALE else 0) + (7 if flags & RELU else 0) + 2
        slab = dev_empty(slots * rows * F)
        ctx.call("cognn_pair_chain_deal_u64", ctypes.byref(c), ptr(slab))
        d0, d1, dp0, dp1 = (dev_empty(shape) for _ in range(4))
        dmask = dev_empty(shape, "u8")
        c.out[0] = d0.data_ptr(); c.out[1] = d1.data_ptr(); c.open[0] = dp0.data_ptr(); c.open[1] = dp1.data_ptr(); c.mask = dmask.data_ptr()
        c.dealt = slab.data_ptr()
        ctx.call("cognn_pair_chain_u64", ctypes.byref(c), 1)
        assert np.array_equal(host(d0), e0) and np.array_equal(host(d1), e1)
        assert np.array_equal(host(dp0), host(op0)) and np.array_equal(host(dp1), host(op1))
        if flags & RELU:
            assert np.array_equal(host(dmask, np.uint8), host(mask, np.uint8))
        # the corrections-only form (COGNN_PC_DEALT_MINIMAL): only party 1's c_1 / r_1 / r'_1 and the ReLU's g are read - every other slot
        # of the slab is overwritten with garbage first, the chain must not notice
        sl = host(slab).reshape(slots, rows * F).copy()
        keep_rows, base = set(), 0
        if flags & TRUNC_IN:
            keep_rows |= {base + 2, base + 4}; base += 5              # r_1, r'_1
        if flags & SCALE:
            keep_rows |= {base + 3, base + 5, base + 7}; base += 8     # c_1, r_1, r'_1
        if flags & RELU:
            keep_rows |= {base + 5, base + 6}; base += 7              # c_1, g
        for r in range(slots):
            if r not in keep_rows:
                sl[r] = 0xDEADBEEFDEADBEEF
        slab2 = dev(sl.reshape(-1))
        if F % 2 == 0:
            for t in (d0, d1, dp0, dp1):
                t.zero_()
            c.dealt = slab2.data_ptr(); c.flags = flags | 512
            ctx.call("cognn_pair_chain_u64", ctypes.byref(c), 1)
            assert np.array_equal(host(d0), e0) and np.array_equal(host(d1), e1)
            assert np.array_equal(host(dp0), host(op0)) and np.array_equal(host(dp1), host(op1))
            c.flags = flags
        c.dealt = None
        c.out[0] = out0.data_ptr(); c.out[1] = out1.data_ptr(); c.open[0] = op0.data_ptr(); c.open[1] = op1.data_ptr(); c.mask = mask.data_ptr()
    with np.errstate(over="ignore"):
        # COGNN_PC_OPEN_SUM: one tensor, the opening as both parties hold it after the exchange
        c.open[0] = op0.data_ptr(); c.open[1] = None; c.flags = flags | OPEN_SUM
        c.out[0] = None; c.out[1] = None
        ctx.call("cognn_pair_chain_u64", ctypes.byref(c), 1)
        assert np.array_equal(host(op0), (e0 - co.prng_shape(ok0, shape)) + (e1 - co.prng_shape(ok1, shape)))
        if flags == TRUNC_IN | NO_C:             # sanity of the expectation itself: a truncation is floor(x / 2^16) or that + 1
            got = (host(out0) + host(out1)).astype(np.int64)
            d = got - (val.astype(np.int64) >> 16)
            assert d.min() >= -1 and d.max() <= 1
    if flags & RELU:
        assert np.array_equal(host(mask, np.uint8).astype(bool), pos)


@pytest.mark.parametrize("rows,F", [(130, 16), (33, 3)])
def test_pair_chain_with_input_mask(ctx, rows, F):
    """mask_in: the backward ReLU' selection (cognn_mask_select_u64 with the public sign mask) folded into the row-scale chain."""
    from cognn_amd import capi
    rng = np.random.default_rng(rows + F)
    shape = (rows, F)
    val = rng.integers(-(1 << 40), 1 << 40, size=shape).astype(np.int64).astype(U64)
    x1 = rand_u64(rng, shape)
    with np.errstate(over="ignore"):
        x0 = val - x1
    m = rng.random(shape) < 0.5
    s0 = co.normalizer(rng.integers(0, 9, size=rows)); s1 = np.zeros(rows, dtype=U64)
    ks = {n: _keys(6, 2, 11, op) for n, op in (("scale", co.OP_PS_SCALE), ("strunc", co.OP_PS_SCALE_TRUNC))}
    kf = {n: v[1] for n, v in ks.items()}
    out0, out1 = dev_empty(shape), dev_empty(shape)
    c = capi.PairChain()
    c.x[0] = dev(x0).data_ptr(); c.x[1] = dev(x1).data_ptr()
    c.scale[0] = dev(s0).data_ptr(); c.scale[1] = dev(s1).data_ptr()
    c.out[0] = out0.data_ptr(); c.out[1] = out1.data_ptr()
    c.mask_in = dev(m.astype(np.uint8)).data_ptr()
    c.scale_keys = ks["scale"][0]; c.scale_trunc_keys = ks["strunc"][0]
    c.rows = rows; c.F = F; c.flags = SCALE
    ctx.call("cognn_pair_chain_u64", ctypes.byref(c), 1)
    e0, e1, _ = _expect(SCALE, np.where(m, x0, U64(0)), np.where(m, x1, U64(0)), None, s0, s1, kf)
    assert np.array_equal(host(out0), e0) and np.array_equal(host(out1), e1)
    c.flags = SCALE | OPENED
    with pytest.raises(capi.CognnError, match="mask_in"):
        ctx.call("cognn_pair_chain_u64", ctypes.byref(c), 1)


@pytest.mark.parametrize("rows,F,scale", [(130, 64, True), (33, 3, True), (70, 16, False)])
def test_pair_chain_with_the_mask_between_truncation_and_scale(ctx, rows, F, scale):
    """COGNN_PC_MASK_AFTER_TRUNC: g = trunc(product + C), the backward ReLU' selection on g, then (optionally) the PreScatter row scale of
    the iteration that consumes it - one chain (the engine's whole-epoch path) against the oracle's steps in that order."""
    from cognn_amd import capi
    MASK_AFTER_TRUNC = 256
    rng = np.random.default_rng(7 * rows + F)
    shape = (rows, F)
    val = rng.integers(-(1 << 44), 1 << 44, size=shape).astype(np.int64).astype(U64)     # the untruncated product (Q32)
    ks = {n: _keys(6, 2, 11, op) for n, op in (("gemm", co.OP_AP_GEMM), ("tin", co.OP_AP_GEMM_TRUNC), ("scale", co.OP_PS_SCALE), ("strunc", co.OP_PS_SCALE_TRUNC))}
    kf = {n: v[1] for n, v in ks.items()}
    c1 = rand_u64(rng, shape)
    with np.errstate(over="ignore"):                         # raw product shares: x0 + C_0 + x1 + c1 = val
        x1 = rand_u64(rng, shape)
        x0 = val - x1 - c1 - co.prng_shape(kf["gemm"](co.SL_C0), shape)
    m = rng.random(shape) < 0.5
    s0 = co.normalizer(rng.integers(0, 9, size=rows)); s1 = np.zeros(rows, dtype=U64)
    out0, out1 = dev_empty(shape), dev_empty(shape)
    c = capi.PairChain()
    c.x[0] = dev(x0).data_ptr(); c.x[1] = dev(x1).data_ptr(); c.c1 = dev(c1).data_ptr()
    c.scale[0] = dev(s0).data_ptr(); c.scale[1] = dev(s1).data_ptr()
    c.out[0] = out0.data_ptr(); c.out[1] = out1.data_ptr()
    c.mask_in = dev(m.astype(np.uint8)).data_ptr()
    c.gemm_keys = ks["gemm"][0]; c.trunc_in_keys = ks["tin"][0]; c.scale_keys = ks["scale"][0]; c.scale_trunc_keys = ks["strunc"][0]
    c.rows = rows; c.F = F; c.flags = TRUNC_IN | MASK_AFTER_TRUNC | (SCALE if scale else 0)
    ctx.call("cognn_pair_chain_u64", ctypes.byref(c), 1)
    g0, g1, _ = _expect(TRUNC_IN, x0, x1, c1, s0, s1, kf)
    g0, g1 = np.where(m, g0, U64(0)), np.where(m, g1, U64(0))
    if scale:
        g0, g1, _ = _expect(SCALE, g0, g1, None, s0, s1, kf)
    assert np.array_equal(host(out0), g0) and np.array_equal(host(out1), g1)
    c.flags = SCALE | MASK_AFTER_TRUNC                       # needs the truncation it follows
    with pytest.raises(capi.CognnError, match="MASK_AFTER_TRUNC"):
        ctx.call("cognn_pair_chain_u64", ctypes.byref(c), 1)


def test_pair_chain_batches_and_rejects_bad_chains(ctx):
    from cognn_amd import capi
    rng = np.random.default_rng(5)
    n_chains = 11                                     # more than one launch batch (8)
    arr = (capi.PairChain * n_chains)()
    exp = []
    k, kf = _keys(8, 1, 2, co.OP_AP_GEMM_TRUNC)
    for i in range(n_chains):
        rows, F = 50 + 13 * i, 4 + i
        x0 = rand_u64(rng, (rows, F)) >> U64(4); x1 = rand_u64(rng, (rows, F)) >> U64(4)
        o0, o1 = dev_empty((rows, F)), dev_empty((rows, F))
        arr[i].x[0] = dev(x0).data_ptr(); arr[i].x[1] = dev(x1).data_ptr(); arr[i].out[0] = o0.data_ptr(); arr[i].out[1] = o1.data_ptr()
        arr[i].trunc_in_keys = k; arr[i].rows = rows; arr[i].F = F; arr[i].flags = TRUNC_IN | NO_C
        exp.append((o0, o1, co.trunc_pair(x0, x1, kf)))
    ctx.call("cognn_pair_chain_u64", arr, n_chains)
    for o0, o1, (e0, e1) in exp:
        assert np.array_equal(host(o0), e0) and np.array_equal(host(o1), e1)
    bad = capi.PairChain()
    bad.rows = 4; bad.F = 4; bad.flags = 0
    with pytest.raises(capi.CognnError, match="no step"):
        ctx.call("cognn_pair_chain_u64", ctypes.byref(bad), 1)
    bad.flags = TRUNC_IN | OPENED | SCALE
    with pytest.raises(capi.CognnError, match="opened input"):
        ctx.call("cognn_pair_chain_u64", ctypes.byref(bad), 1)
    bad.flags = SCALE
    with pytest.raises(capi.CognnError, match="null"):
        ctx.call("cognn_pair_chain_u64", ctypes.byref(bad), 1)


@pytest.mark.parametrize("F,relu,forward_only", [(64, True, False), (16, False, False), (64, True, True), (6, True, False), (2, False, False),
                                                 (64, None, True), (16, None, False), (64, True, "sum"), (16, None, "sum"),
                                                 # odd widths (7 or 3 labels): 8-byte lanes
                                                 (7, False, False), (3, True, False), (7, None, "sum"), (1, False, False), (33, True, "sum")])
def test_gather_with_pair_chain_epilogue_equals_gather_then_chain(ctx, F, relu, forward_only):
    """cognn_gather_pair_chain_u64 (the aggregate never written) against cognn_gather_csr_u64 on both sides' row segments followed
    by cognn_pair_chain_u64, and against the oracle's two-party functions; two owners of different sizes (tiles of 32 vertices,
    one partial tile), empty CSR rows, a hub row longer than the staged slice.  relu=None: no step at all (the last backward
    Gather of an epoch): outputs / openings of the aggregate itself."""
    from cognn_amd import capi
    rng = np.random.default_rng(900 + F)
    sizes = [70, 33]
    # table: [A_0 | B_0 | A_1 | B_1] with even-row alignment like the engine's layout
    offs, off = [], 0
    for n in sizes:
        a = off; off = (off + n + 1) & ~1
        b = off; off = (off + n + 1) & ~1
        offs.append((a, b))
    T = off
    deg = rng.poisson(5, size=T); deg[rng.random(T) < 0.2] = 0
    deg[offs[0][0] + 3] = 2000                                 # longer than the 1536 staged entries of its tile
    rowptr = np.zeros(T + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, T, size=int(rowptr[-1]), dtype=np.uint32)
    val = rng.integers(-(1 << 30), 1 << 30, size=(T, F)).astype(np.int64).astype(U64)       # small values: sums stay in range
    dtab, drp, dcl = dev(val), dev(rowptr.view(np.int32)), dev(col.view(np.int32))
    pairs = (capi.GatherPair * len(sizes))()
    keep = []
    for i, (n, (a, b)) in enumerate(zip(sizes, offs)):
        ks = {nm: _keys(7, i, 3, op) for nm, op in (("scale", co.OP_GA_SCALE), ("strunc", co.OP_GA_SCALE_TRUNC), ("relu", co.OP_AP_RELU))}
        s0 = co.normalizer(rng.integers(0, 9, size=n)); s1 = np.zeros(n, dtype=U64)
        bufs = [dev_empty((n, F)) for _ in range(4)]; mask = dev_empty((n, F), "u8")
        ok = (co.stream_key(2, i, 3, 4, 0), co.stream_key(2, i, 3, 4, 1))
        p = pairs[i]
        p.a_row0 = a; p.b_row0 = b
        c = p.chain
        c.scale[0] = dev(s0).data_ptr(); c.scale[1] = dev(s1).data_ptr()
        if not forward_only:
            c.out[0] = bufs[0].data_ptr(); c.out[1] = bufs[1].data_ptr(); c.mask = mask.data_ptr()
        c.open[0] = bufs[2].data_ptr(); c.open[1] = None if forward_only == "sum" else bufs[3].data_ptr(); c.open_key[0] = ok[0]; c.open_key[1] = ok[1]
        c.scale_keys = ks["scale"][0]; c.scale_trunc_keys = ks["strunc"][0]; c.relu_keys = ks["relu"][0]
        c.rows = n; c.F = F; c.flags = (0 if relu is None else SCALE | (RELU if relu else 0)) | (OPEN_SUM if forward_only == "sum" else 0)
        keep.append((n, a, b, s0, s1, {k: v[1] for k, v in ks.items()}, bufs, mask, ok))
    ctx.call("cognn_gather_pair_chain_u64", ptr(dtab), ptr(drp), ptr(dcl), F, pairs, len(sizes))
    # the plain aggregate of every table row
    agg = val.copy()
    with np.errstate(over="ignore"):
        for r in range(T):
            for q in range(rowptr[r], rowptr[r + 1]):
                agg[r] += val[col[q]]
    for n, a, b, s0, s1, kf, bufs, mask, ok in keep:
        v0, v1 = agg[a:a + n], agg[b:b + n]
        if relu is None:
            e0, e1 = v0, v1
        else:
            z0, z1 = co.beaver_rowscale_pair(v0, v1, s0, s1, kf["scale"])
            e0, e1 = co.trunc_pair(z0, z1, kf["strunc"])
        if relu:
            e0, e1, pos = co.relu_pair(e0, e1, kf["relu"])
            if not forward_only:
                assert np.array_equal(host(mask, np.uint8).astype(bool), pos)
        with np.errstate(over="ignore"):
            if forward_only == "sum":                     # COGNN_PC_OPEN_SUM: the two openings summed into open[0]
                assert np.array_equal(host(bufs[2]), (e0 - co.prng_shape(ok[0], e0.shape)) + (e1 - co.prng_shape(ok[1], e1.shape)))
            else:
                assert np.array_equal(host(bufs[2]), e0 - co.prng_shape(ok[0], e0.shape))
                assert np.array_equal(host(bufs[3]), e1 - co.prng_shape(ok[1], e1.shape))
        if not forward_only:
            assert np.array_equal(host(bufs[0]), e0) and np.array_equal(host(bufs[1]), e1)


@pytest.mark.parametrize("F,scale,write_logits", [(16, True, False), (16, True, True), (7, True, False), (3, True, False), (6, True, False), (64, True, False),
                                                   (1, True, False), (2, False, False), (31, True, False), (16, False, True)])
def test_gather_with_the_prediction_layer_as_second_epilogue(ctx, F, scale, write_logits):
    """cognn_gather_pair::softmax: the label-wide Gather, its scale and the prediction layer (softmax - label, reveal, metrics;
    gcn.h:578-632) in one launch - against the oracle's two-party functions on the plain aggregate (the logits are only written when
    asked for), incl. odd label counts, one label, empty CSR rows, a hub row and a partial tile."""
    from cognn_amd import capi
    rng = np.random.default_rng(1200 + F)
    sizes = [70, 33, 1]
    offs, off = [], 0
    for n in sizes:
        a = off; off = (off + n + 1) & ~1
        b = off; off = (off + n + 1) & ~1
        offs.append((a, b))
    T = off
    deg = rng.poisson(5, size=T); deg[rng.random(T) < 0.2] = 0
    deg[offs[0][0] + 3] = 1700
    rowptr = np.zeros(T + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, T, size=int(rowptr[-1]), dtype=np.uint32)
    # small fixed-point values (a few units in Q16) on every table row: the two sides' aggregates are shares of logits a few units apart
    val = rng.integers(-(1 << 17), 1 << 17, size=(T, F)).astype(np.int64).astype(U64)
    dtab, drp, dcl = dev(val), dev(rowptr.view(np.int32)), dev(col.view(np.int32))
    pairs = (capi.GatherPair * len(sizes))()
    jobs = (capi.SoftmaxJob * (2 * len(sizes)))()
    keep = []
    for i, (n, (a, b)) in enumerate(zip(sizes, offs)):
        ks = {nm: _keys(7, i, 3, op) for nm, op in (("scale", co.OP_GA_SCALE), ("strunc", co.OP_GA_SCALE_TRUNC), ("smx", co.OP_AP_SOFTMAX))}
        s0 = co.normalizer(rng.integers(0, 9, size=n)); s1 = np.zeros(n, dtype=U64)
        bufs = [dev_empty((n, F)) for _ in range(4)]
        labels = rng.integers(0, F, size=n).astype(np.int32); border = (rng.random(n) < 0.3).astype(np.uint8)
        train, valr = n // 3, n // 4
        cnt, loss = dev_empty(6), dev_empty(1, "f64")
        cnt.fill_(77); loss.fill_(3.0)                       # (the call zeroes them)
        dl, db = dev(labels), dev(border)
        p = pairs[i]
        p.a_row0 = a; p.b_row0 = b
        c = p.chain
        c.scale[0] = dev(s0).data_ptr(); c.scale[1] = dev(s1).data_ptr()
        if write_logits:
            c.out[0] = bufs[0].data_ptr(); c.out[1] = bufs[1].data_ptr()
        c.scale_keys = ks["scale"][0]; c.scale_trunc_keys = ks["strunc"][0]
        c.rows = n; c.F = F; c.flags = SCALE if scale else 0
        for q in (0, 1):
            j = jobs[2 * i + q]
            j.d_out = bufs[2 + q].data_ptr(); j.keys = ks["smx"][0]; j.p = q; j.rows = n; j.train_rows = train; j.val_rows = valr
            if q == 0:
                j.labels = dl.data_ptr(); j.border = db.data_ptr(); j.counts6 = cnt.data_ptr(); j.loss = loss.data_ptr()
            p.softmax[q] = ctypes.addressof(j)
        keep.append((n, a, b, s0, s1, {k: v[1] for k, v in ks.items()}, bufs, labels, border, train, valr, cnt, loss, dl, db))
    assert capi.load().cognn_gather_pair_chain_takes_softmax(F) == 1
    ctx.call("cognn_gather_pair_chain_u64", ptr(dtab), ptr(drp), ptr(dcl), F, pairs, len(sizes))
    agg = val.copy()
    with np.errstate(over="ignore"):
        for r in range(T):
            for q in range(rowptr[r], rowptr[r + 1]):
                agg[r] += val[col[q]]
    for n, a, b, s0, s1, kf, bufs, labels, border, train, valr, cnt, loss, dl, db in keep:
        v0, v1 = agg[a:a + n], agg[b:b + n]
        if scale:
            z0, z1 = co.beaver_rowscale_pair(v0, v1, s0, s1, kf["scale"])
            v0, v1 = co.trunc_pair(z0, z1, kf["strunc"])
        if write_logits:
            assert np.array_equal(host(bufs[0]), v0) and np.array_equal(host(bufs[1]), v1)
        p0, p1, d0, d1, plainP = co.softmax_pair(v0, v1, labels, train, kf["smx"])
        assert np.array_equal(host(bufs[2]), d0) and np.array_equal(host(bufs[3]), d1)
        pp = np.where(plainP == 0, 0.001, plainP)
        ok = pp.argmax(1) == labels
        idx = np.arange(n); tr = idx < train; te = idx >= train + valr; bd = border.astype(bool)
        c = host(cnt, np.int64)
        assert list(c[:5]) == [ok.sum(), (ok & tr).sum(), (ok & tr & bd).sum(), (ok & te).sum(), (ok & te & bd).sum()]
        want_loss = -np.log(pp[idx, labels]).sum()
        assert abs(float(host(loss, np.float64)[0]) - want_loss) < 1e-9 * max(1.0, abs(want_loss))   # fp tolerance: atomics order
    # a ReLU or an opening in the same chain is refused
    pairs[0].chain.flags = SCALE | RELU
    with pytest.raises(capi.CognnError, match="softmax follows"):
        ctx.call("cognn_gather_pair_chain_u64", ptr(dtab), ptr(drp), ptr(dcl), F, pairs, len(sizes))


@pytest.mark.parametrize("F,flags,softmax", [(64, SCALE | RELU, False), (16, SCALE, True), (7, SCALE, True), (16, 0, False)])
def test_gather_pair_chain_from_a_base_equals_the_one_launch_form(ctx, F, flags, softmax):
    """cognn_gather_pair_chain_base_u64: the rows' entries split over two launches (a plain cognn_gather_csr_u64 over the first part
    leaves the sums, the launch over the rest starts from them and carries the epilogue - what a multi-rank run does with the
    entries that read received rows) gives exactly what one launch over all entries gives: outputs, opening, prediction layer."""
    from cognn_amd import capi
    rng = np.random.default_rng(77 + F + flags)
    sizes = [70, 33]
    offs, off = [], 0
    for n in sizes:
        a = off; off = (off + n + 1) & ~1
        b = off; off = (off + n + 1) & ~1
        offs.append((a, b))
    T = off
    deg = rng.poisson(6, size=T); deg[rng.random(T) < 0.2] = 0
    rowptr = np.zeros(T + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, T, size=int(rowptr[-1]), dtype=np.uint32)
    cut = np.array([rng.integers(0, d + 1) for d in deg])            # entries [0, cut) of a row are "local", the rest "received"
    rp1 = np.zeros(T + 1, dtype=np.uint32); rp1[1:] = np.cumsum(cut)
    rp2 = np.zeros(T + 1, dtype=np.uint32); rp2[1:] = np.cumsum(deg - cut)
    c1 = np.concatenate([col[rowptr[r]:rowptr[r] + cut[r]] for r in range(T)] + [np.zeros(0, np.uint32)]).astype(np.uint32)
    c2 = np.concatenate([col[rowptr[r] + cut[r]:rowptr[r + 1]] for r in range(T)] + [np.zeros(0, np.uint32)]).astype(np.uint32)
    val = rng.integers(-(1 << 17), 1 << 17, size=(T, F)).astype(np.int64).astype(U64)
    dtab = dev(val)
    results = []
    for split in (False, True):
        pairs = (capi.GatherPair * len(sizes))()
        jobs = (capi.SoftmaxJob * (2 * len(sizes)))()
        outs = []
        for i, (n, (a, b)) in enumerate(zip(sizes, offs)):
            ks = {nm: _keys(9, i, 4, op) for nm, op in (("scale", co.OP_GA_SCALE), ("strunc", co.OP_GA_SCALE_TRUNC), ("relu", co.OP_AP_RELU), ("smx", co.OP_AP_SOFTMAX))}
            s0 = co.normalizer(np.arange(n) % 9); s1 = np.zeros(n, dtype=U64)
            bufs = [dev_empty((n, F)) for _ in range(5)]; cnt, loss = dev_empty(6), dev_empty(1, "f64")
            labels = dev((np.arange(n) % F).astype(np.int32))
            p = pairs[i]
            p.a_row0 = a; p.b_row0 = b
            c = p.chain
            c.scale[0] = dev(s0).data_ptr(); c.scale[1] = dev(s1).data_ptr()
            c.scale_keys = ks["scale"][0]; c.scale_trunc_keys = ks["strunc"][0]; c.relu_keys = ks["relu"][0]
            c.out[0] = bufs[0].data_ptr(); c.out[1] = bufs[1].data_ptr()
            if not softmax:
                c.open[0] = bufs[2].data_ptr(); c.open_key[0] = 5; c.open_key[1] = 6; c.flags = flags | OPEN_SUM
            else:
                c.flags = flags
                for q in (0, 1):
                    j = jobs[2 * i + q]
                    j.d_out = bufs[3 + q].data_ptr(); j.keys = ks["smx"][0]; j.p = q; j.rows = n; j.train_rows = n // 2; j.val_rows = n // 4
                    if q == 0:
                        j.labels = labels.data_ptr(); j.counts6 = cnt.data_ptr(); j.loss = loss.data_ptr()
                    p.softmax[q] = ctypes.addressof(j)
            c.rows = n; c.F = F
            outs.append((bufs, cnt, labels))
        if split:
            base = dev_empty((T, F))
            ctx.call("cognn_gather_csr_u64", ptr(base), ptr(dtab), ptr(dtab), ptr(dev(rp1.view(np.int32))), ptr(dev(c1.view(np.int32))), T, F)
            ctx.call("cognn_gather_pair_chain_base_u64", ptr(dtab), ptr(base), ptr(dev(rp2.view(np.int32))), ptr(dev(c2.view(np.int32))), F, pairs, len(sizes))
        else:
            ctx.call("cognn_gather_pair_chain_u64", ptr(dtab), ptr(dev(rowptr.view(np.int32))), ptr(dev(col.view(np.int32))), F, pairs, len(sizes))
        results.append([[host(t) for t in bufs] + [host(cnt, np.int64)[:5]] for bufs, cnt, _ in outs])
    for one, two in zip(*results):
        for x, y in zip(one, two):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("n", [1, 7, 16 * 7, 1433 * 16 + 1])
@pytest.mark.parametrize("pairs,average,avg_scale,post,raw", [(1, 0, 0, 0, 1), (3, 0, 0, 1, 0), (2, 1, 1, 0, 1), (5, 1, 0, 1, 1), (16, 1, 1, 0, 0)])
def test_pair_weight_update_matches_the_two_party_oracle(ctx, n, pairs, average, avg_scale, post, raw):
    """cognn_pair_weight_update_u64 (product truncation, gradient scale, learning rate, update, optional 1/k scale, optional
    weight average with its scale) against the oracle's truncation pairs chained the way the reference's ApplyComp + weight
    average do (gcn.h:671-684, 710-736, 753-778)."""
    from cognn_amd import capi
    rng = np.random.default_rng(n * 31 + pairs * 7 + average + 2 * post + 4 * raw)
    seed, it = 77, 5
    jobs = (capi.PairWUpdate * pairs)()
    exp, Wd = [], []
    gs = [int(co.fx_encode_trunc(1.0 / (40 + 3 * o))) for o in range(pairs)]
    lr, ws = int(co.fx_encode_trunc(0.5)), int(co.fx_encode_trunc(1.0 / pairs))
    with np.errstate(over="ignore"):
        for o in range(pairs):
            val = rng.integers(-(1 << 44), 1 << 44, size=n).astype(np.int64).astype(U64)       # Q32 product of Q16 values
            z1 = rand_u64(rng, n); z0 = val - z1
            c1 = rand_u64(rng, n) >> U64(30)
            wv = rng.integers(-(1 << 20), 1 << 20, size=n).astype(np.int64).astype(U64)
            w1 = rand_u64(rng, n); w0 = wv - w1
            ks = [_keys(seed, o, it, op) for op in (co.OP_AP_GEMM, co.OP_AP_GEMM_TRUNC, co.OP_AP_GSCALE_TRUNC, co.OP_AP_LR_TRUNC, co.OP_WAVG_TRUNC)]
            J = jobs[o]
            W0d, W1d = dev(w0), dev(w1)
            Wd.append((W0d, W1d))
            J.z[0] = dev(z0).data_ptr(); J.z[1] = dev(z1).data_ptr(); J.c1 = dev(c1).data_ptr()
            J.W[0] = W0d.data_ptr(); J.W[1] = W1d.data_ptr()
            J.gemm_keys = ks[0][0]
            for t in range(4):
                J.trunc_keys[t] = ks[1 + t][0]
            J.mul[0] = gs[o]; J.mul[1] = lr; J.mul[2] = ws if post else 0
            J.n = n; J.flags = (0 if raw else NO_C) | (capi.WU_SWAP if o >= 1 else 0)
            d0, d1 = (z0 + co.prng_shape(ks[0][1](co.SL_C0), (n,)), z1 + c1) if raw else (z0, z1)
            d0, d1 = co.trunc_pair(d0, d1, ks[1][1])
            d0, d1 = co.const_scale_trunc_pair(d0, d1, gs[o], ks[2][1])
            u0, u1 = co.const_scale_trunc_pair(d0, d1, lr, ks[3][1])
            n0, n1 = w0 - u0, w1 - u1
            if post:
                n0, n1 = co.const_scale_trunc_pair(n0, n1, ws, ks[4][1])
            exp.append((n0, n1))
        ak = _keys(seed, co.OWNER_WAVG, it, co.OP_WAVG_TRUNC)
        if average:
            s0 = np.zeros(n, dtype=U64); s1 = np.zeros(n, dtype=U64)
            for o in range(pairs):
                s0 = s0 + exp[o][0 if o == 0 else 1]; s1 = s1 + exp[o][1 if o == 0 else 0]
            if avg_scale:
                s0, s1 = co.const_scale_trunc_pair(s0, s1, ws, ak[1])
            exp = [((s0, s1) if o == 0 else (s1, s0)) for o in range(pairs)]
    ctx.call("cognn_pair_weight_update_u64", jobs, pairs, ctypes.byref(ak[0]), ws if (average and avg_scale) else 0, average)
    ctx.sync()
    for o in range(pairs):
        assert np.array_equal(host(Wd[o][0]), exp[o][0]) and np.array_equal(host(Wd[o][1]), exp[o][1]), o


@pytest.mark.parametrize("rows,cols", [(7, 16), (16, 64), (3, 5), (1, 1)])
def test_mask_open_reads_a_transposed_weight_matrix(ctx, rows, cols):
    """transposed = 3: X is stored [cols x rows], the opening is written in logical [rows x cols] order (W1^T of gcn.h:648,665
    without the transposed copy)."""
    rng = np.random.default_rng(rows * 17 + cols)
    X = rand_u64(rng, (cols, rows))
    key = co.stream_key(3, 1, 4, co.OP_AP_GEMM, co.SL_B0)
    E = dev_empty((rows, cols))
    ctx.call("cognn_mask_open_u64", ptr(E), ptr(dev(X)), ctypes.c_uint64(key), rows, cols, 3)
    ctx.sync()
    with np.errstate(over="ignore"):
        assert np.array_equal(host(E), X.T - co.prng_shape(key, (rows, cols)))


def test_consumers_hand_the_product_buffers_back_clean(ctx):
    """COGNN_PC_CLEAR_INPUT / COGNN_WU_CLEAR_Z: the kernel that reads a raw product zeroes it behind the read (the next split-K
    product then skips its zeroing launch, cognn_gemm_job::Z_zeroed); results are those of the plain form."""
    from cognn_amd import capi
    rng = np.random.default_rng(12)
    rows, F = 301, 16
    n = rows * F
    x0 = rand_u64(rng, (rows, F)) >> U64(20); x1 = rand_u64(rng, (rows, F)); c1 = rand_u64(rng, (rows, F)) >> U64(30)
    ks = {nm: _keys(5, 3, 9, op) for nm, op in (("gemm", co.OP_PS_GEMM), ("tin", co.OP_PS_GEMM_TRUNC))}
    outs = []
    for clear in (0, capi.PC_CLEAR_INPUT):
        X0, X1 = dev(x0), dev(x1)
        out0, out1 = dev_empty((rows, F)), dev_empty((rows, F))
        c = capi.PairChain()
        c.x[0] = X0.data_ptr(); c.x[1] = X1.data_ptr(); c.c1 = dev(c1).data_ptr()
        c.out[0] = out0.data_ptr(); c.out[1] = out1.data_ptr()
        c.gemm_keys = ks["gemm"][0]; c.trunc_in_keys = ks["tin"][0]
        c.rows = rows; c.F = F; c.flags = TRUNC_IN | clear
        ctx.call("cognn_pair_chain_u64", ctypes.byref(c), 1)
        ctx.sync()
        outs.append((host(out0).copy(), host(out1).copy()))
        if clear:
            assert not host(X0).any() and not host(X1).any()
        else:
            assert np.array_equal(host(X0), x0) and np.array_equal(host(X1), x1)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    # an output that aliases the product buffer cannot be combined with the clear
    c.out[0] = c.x[0]
    with pytest.raises(capi.CognnError, match="CLEAR_INPUT"):
        ctx.call("cognn_pair_chain_u64", ctypes.byref(c), 1)
    # weight update
    res = []
    for clear in (0, capi.WU_CLEAR_Z):
        jobs = (capi.PairWUpdate * 1)()
        J = jobs[0]
        Z0, Z1, W0, W1 = dev(x0.reshape(-1)), dev(x1.reshape(-1)), dev(x1.reshape(-1) >> U64(3)), dev(x0.reshape(-1))
        J.z[0] = Z0.data_ptr(); J.z[1] = Z1.data_ptr(); J.c1 = dev(c1.reshape(-1)).data_ptr(); J.W[0] = W0.data_ptr(); J.W[1] = W1.data_ptr()
        J.gemm_keys = ks["gemm"][0]
        for t in range(4):
            J.trunc_keys[t] = _keys(5, 3, 9, co.OP_AP_GSCALE_TRUNC + t)[0]
        J.mul[0] = 700; J.mul[1] = 32768; J.mul[2] = 0; J.n = n; J.flags = clear
        ctx.call("cognn_pair_weight_update_u64", jobs, 1, None, 0, 0)
        ctx.sync()
        res.append((host(W0).copy(), host(W1).copy()))
        assert (not host(Z0).any() and not host(Z1).any()) if clear else np.array_equal(host(Z0), x0.reshape(-1))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
