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
            assert d.min() >= 0 and d.max() <= 1
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
                                                 (64, None, True), (16, None, False), (64, True, "sum"), (16, None, "sum")])
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
