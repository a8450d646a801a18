"""Parity of every C-ABI arithmetic entry point against the CPU oracle (bit-exact; integer ring).
The oracle is "parity unpinned" w.r.t. the reference (no golden vectors exist)."""
import ctypes
import numpy as np
import pytest

import cognn_oracle as co
from gpu_util import dev, dev_empty, host, ptr, rand_u64, U64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from cognn_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def keys_of(seed, owner, it, op):
    from cognn_amd import capi
    return capi.make_keys(seed, owner, it, op), (lambda slot: co.stream_key(seed, owner, it, op, slot))


def test_device_epoch_salt_belongs_to_one_context(ctx):
    """cognn_set_epoch_salt (the device-side salt of recorded epochs): while one context holds a non-zero salt every launch of
    another context is refused - it would evaluate its dealer streams under a salt it did not set - and so is a second non-zero
    salt; after the reset (which synchronises) both work again and read salt 0."""
    from cognn_amd import capi
    other = capi.Context(0)
    try:
        n, key = 4096, co.stream_key(7, 1, 2, 3, 4)
        out = dev_empty(n)
        ctx.call("cognn_set_epoch_salt", 5)
        ctx.call("cognn_prng_fill_u64", ptr(out), key, n)
        assert np.array_equal(host(out), co.prng(key + 5, n))
        with pytest.raises(capi.CognnError, match="epoch salt"):
            other.call("cognn_prng_fill_u64", ptr(out), key, n)
        with pytest.raises(capi.CognnError, match="epoch salt"):
            other.call("cognn_set_epoch_salt", 9)
        ctx.call("cognn_set_epoch_salt", 0)
        other.call("cognn_prng_fill_u64", ptr(out), key, n)
        assert np.array_equal(host(out), co.prng(key, n))
        other.call("cognn_set_epoch_salt", 9)                 # ... and now the other context may hold it
        with pytest.raises(capi.CognnError, match="epoch salt"):
            ctx.call("cognn_prng_fill_u64", ptr(out), key, n)
        other.call("cognn_set_epoch_salt", 0)
        ctx.call("cognn_prng_fill_u64", ptr(out), key, n)
        assert np.array_equal(host(out), co.prng(key, n))
    finally:
        other.call("cognn_set_epoch_salt", 0)
        other.close()


def test_prng_and_share_split(ctx):
    n = 100003
    key = co.stream_key(7, 1, 2, 3, 4)
    out = dev_empty(n)
    ctx.call("cognn_prng_fill_u64", ptr(out), ctypes.c_uint64(key), n)
    assert np.array_equal(host(out), co.prng(key, n))
    rng = np.random.default_rng(0)
    fx = rand_u64(rng, n)
    s0, s1 = dev_empty(n), dev_empty(n)
    ctx.call("cognn_share_split_u64", ptr(dev(fx)), ctypes.c_uint64(key), ptr(s0), ptr(s1), n)
    with np.errstate(over="ignore"):
        assert np.array_equal(host(s1), co.prng(key, n))
        assert np.array_equal(host(s0) + host(s1), fx)


def test_fx_encode(ctx):
    rng = np.random.default_rng(1)
    rows, cols = 301, 17
    x = rng.normal(size=(rows, cols)) * 3
    x[0, 0] = 0.5 / 65536; x[0, 1] = -0.5 / 65536; x[0, 2] = 1.5 / 65536
    sc = rng.random(rows) + 0.1
    out = dev_empty((rows, cols))
    ctx.call("cognn_fx_encode_f64", ptr(dev(x)), ptr(dev(sc)), ptr(out), rows, cols)
    assert np.array_equal(host(out), co.fx_encode(x * sc[:, None]))
    ctx.call("cognn_fx_encode_f64", ptr(dev(x)), None, ptr(out), rows, cols)
    assert np.array_equal(host(out), co.fx_encode(x))


def _random_csr(rng, n_rows, n_table, avg_deg, empty_frac=0.1):
    deg = rng.poisson(avg_deg, size=n_rows)
    deg[rng.random(n_rows) < empty_frac] = 0
    rowptr = np.zeros(n_rows + 1, dtype=np.uint32)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n_table, size=int(rowptr[-1]), dtype=np.uint32)
    return rowptr, col


def _csr_ref(base, table, rowptr, col):
    out = np.zeros((len(rowptr) - 1, table.shape[1]), dtype=U64) if base is None else base.copy()
    with np.errstate(over="ignore"):
        for r in range(len(rowptr) - 1):
            for e in range(rowptr[r], rowptr[r + 1]):
                out[r] += table[col[e]]
    return out


@pytest.mark.parametrize("F", [64, 16, 7, 1, 2, 130, 6])
@pytest.mark.parametrize("with_base", [True, False])
def test_gather_csr(ctx, F, with_base):
    rng = np.random.default_rng(F * 2 + with_base)
    n_rows, n_table = 777, 1500
    rowptr, col = _random_csr(rng, n_rows, n_table, 9)
    table = rand_u64(rng, (n_table, F))
    base = rand_u64(rng, (n_rows, F)) if with_base else None
    out = dev_empty((n_rows, F))
    ctx.call("cognn_gather_csr_u64", ptr(out), ptr(dev(base)) if with_base else None, ptr(dev(table)),
             ptr(dev(rowptr.view(np.int32))), ptr(dev(col.view(np.int32))), n_rows, F)
    assert np.array_equal(host(out), _csr_ref(base, table, rowptr, col))


def test_gather_csr_heavy_rows_and_empty(ctx):
    """Tiles whose edge count exceeds the LDS staging capacity, plus zero-row and all-empty inputs."""
    rng = np.random.default_rng(5)
    n_rows, n_table, F = 130, 400, 16
    deg = np.full(n_rows, 3); deg[7] = 5000; deg[64] = 4000
    rowptr = np.zeros(n_rows + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n_table, size=int(rowptr[-1]), dtype=np.uint32)
    table = rand_u64(rng, (n_table, F))
    out = dev_empty((n_rows, F))
    ctx.call("cognn_gather_csr_u64", ptr(out), None, ptr(dev(table)), ptr(dev(rowptr.view(np.int32))),
             ptr(dev(col.view(np.int32))), n_rows, F)
    assert np.array_equal(host(out), _csr_ref(None, table, rowptr, col))
    ctx.call("cognn_gather_csr_u64", ptr(out), None, ptr(dev(table)), ptr(dev(rowptr.view(np.int32))),
             ptr(dev(col.view(np.int32))), 0, F)
    rp0 = np.zeros(n_rows + 1, dtype=np.uint32)
    base = rand_u64(rng, (n_rows, F))
    ctx.call("cognn_gather_csr_u64", ptr(out), ptr(dev(base)), ptr(dev(table)), ptr(dev(rp0.view(np.int32))),
             ptr(dev(col.view(np.int32))), n_rows, F)
    assert np.array_equal(host(out), base)


def test_gather_matches_unfused_reference_chain(ctx):
    """Fused CSR gather == OEP -> copy -> prefix_network_aggregate -> OEP -> cond-add of the oracle."""
    rng = np.random.default_rng(11)
    n, F = 50, 8
    pos = np.arange(100, 100 + n)                       # localVertexPos
    srcs, dsts, dummy = [], [], []
    for v in pos:
        d = rng.integers(0, 5)
        if d == 0:
            srcs.append(v); dsts.append(v); dummy.append(True)
        else:
            s = rng.choice(pos, size=d)
            srcs.extend(s); dsts.extend([v] * d); dummy.append(False)
    x = rand_u64(rng, (n, F))
    upd = co.oep(pos, srcs, x)
    ext = co.oep(dsts, pos, co.prefix_network_aggregate(dsts, upd))
    with np.errstate(over="ignore"):
        want = x + np.where(~np.array(dummy)[:, None], ext, U64(0))
    rowptr = [0]; col = []
    i = 0
    for r, v in enumerate(pos):
        cnt = 0
        while i < len(dsts) and dsts[i] == v:
            if not dummy[r]:
                col.append(srcs[i] - 100); cnt += 1
            i += 1
        rowptr.append(rowptr[-1] + cnt)
    rowptr = np.array(rowptr, dtype=np.uint32); col = np.array(col, dtype=np.uint32)
    xd = dev(x); out = dev_empty((n, F))
    ctx.call("cognn_gather_csr_u64", ptr(out), ptr(xd), ptr(xd), ptr(dev(rowptr.view(np.int32))),
             ptr(dev(col.view(np.int32))), n, F)
    assert np.array_equal(host(out), want)


@pytest.mark.parametrize("F", [64, 16, 7])
def test_scatter_add_rows(ctx, F):
    rng = np.random.default_rng(F)
    n, q = 500, 211
    v = rand_u64(rng, (n, F)); part = rand_u64(rng, (q, F))
    idx = rng.permutation(n)[:q].astype(np.uint32)
    vd = dev(v)
    ctx.call("cognn_scatter_add_rows_u64", ptr(vd), ptr(dev(part)), ptr(dev(idx.view(np.int32))), q, F)
    want = v.copy()
    with np.errstate(over="ignore"):
        want[idx] += part
    assert np.array_equal(host(vd), want)


GEMM_SHAPES = [
    # M, N, K, transA
    (300, 64, 128, 0), (1000, 16, 64, 0), (257, 7, 33, 0), (4096, 64, 128, 0), (513, 48, 96, 0),
    (700, 128, 64, 0), (16, 7, 5000, 1), (128, 64, 3000, 1), (5, 3, 2, 0), (260, 16, 16, 0),
]


@pytest.mark.parametrize("M,N,K,transA", GEMM_SHAPES)
def test_ring_gemm(ctx, M, N, K, transA):
    rng = np.random.default_rng(M + N + K)
    A = rand_u64(rng, (M, K)); B = rand_u64(rng, (K, N)); C0 = rand_u64(rng, (M, N))
    Ad = dev(A.T.copy() if transA else A); Bd = dev(B); Cd = dev(C0)
    ctx.call("cognn_ring_gemm_u64", ptr(Cd), ptr(Ad), ptr(Bd), M, N, K, transA, 0)
    want = co.ring_matmul(A, B)
    assert np.array_equal(host(Cd), want)
    Cd = dev(C0)
    ctx.call("cognn_ring_gemm_u64", ptr(Cd), ptr(Ad), ptr(Bd), M, N, K, transA, 1)
    with np.errstate(over="ignore"):
        assert np.array_equal(host(Cd), want + C0)
    if not transA or K < 4000:
        A2 = rand_u64(rng, (M, K))
        A2d = dev(A2.T.copy() if transA else A2)
        Cd = dev(C0)
        ctx.call("cognn_ring_gemm2_u64", ptr(Cd), ptr(Ad), ptr(A2d), ptr(Bd), M, N, K, transA, 0)
        with np.errstate(over="ignore"):
            assert np.array_equal(host(Cd), co.ring_matmul(A + A2, B))


def test_ring_gemm_limb_edge_values(ctx):
    """Operands made of 0x80 / 0x7f / 0xff bytes exercise the signed-limb carry chain."""
    M, N, K = 256, 64, 64
    vals = np.array([0x8080808080808080, 0x7F7F7F7F7F7F7F7F, 0xFFFFFFFFFFFFFFFF, 0x0000000000000080,
                     0x8000000000000000, 0x00FF00FF00FF00FF, 1, 0], dtype=U64)
    rng = np.random.default_rng(3)
    A = vals[rng.integers(0, len(vals), size=(M, K))]
    B = vals[rng.integers(0, len(vals), size=(K, N))]
    Cd = dev_empty((M, N))
    ctx.call("cognn_ring_gemm_u64", ptr(Cd), ptr(dev(A)), ptr(dev(B)), M, N, K, 0, 0)
    assert np.array_equal(host(Cd), co.ring_matmul(A, B))


def test_trunc_pair(ctx):
    rng = np.random.default_rng(2)
    n = 4099
    x = (rng.normal(size=n) * 2 ** 36).astype(np.int64).astype(U64)       # Q32-ish magnitudes
    x0 = rand_u64(rng, n)
    with np.errstate(over="ignore"):
        x1 = x - x0
    k, kf = keys_of(9, 3, 5, co.OP_PS_GEMM_TRUNC)
    c0, c1, o0, o1 = (dev_empty(n) for _ in range(4))
    ctx.call("cognn_trunc_open_u64", ptr(c0), ptr(dev(x0)), ctypes.c_uint64(1), ctypes.byref(k), 0, n)
    ctx.call("cognn_trunc_open_u64", ptr(c1), ptr(dev(x1)), ctypes.c_uint64(1), ctypes.byref(k), 1, n)
    ctx.call("cognn_trunc_close_u64", ptr(o0), ptr(c0), ptr(c1), ctypes.byref(k), 0, 0, n)
    ctx.call("cognn_trunc_close_u64", ptr(o1), None, None, ctypes.byref(k), 1, 0, n)
    w0, w1 = co.trunc_pair(x0, x1, kf)
    assert np.array_equal(host(o0), w0) and np.array_equal(host(o1), w1)
    with np.errstate(over="ignore"):
        rec = (host(o0) + host(o1)).astype(np.int64)
    exact = x.astype(np.int64) >> 16
    assert np.all((rec - exact >= -1) & (rec - exact <= 1))              # floor + {-1, 0, +1}: the mask's rounding and the dropped carry of the 48-bit opening
    # close + opening of the consuming op in one pass: E = y - mask(key_open)
    ko = 0x1234567890ABCDEF
    for p, (cc0, cc1, w) in enumerate(((c0, c1, w0), (None, None, w1))):
        o = dev_empty(n); e = dev_empty(n)
        ctx.call("cognn_trunc_close_open_u64", ptr(o), ptr(e), ptr(cc0) if cc0 is not None else None,
                 ptr(cc1) if cc1 is not None else None, ctypes.byref(k), p, ctypes.c_uint64(ko), n)
        with np.errstate(over="ignore"):
            assert np.array_equal(host(o), w) and np.array_equal(host(e), w - co.prng_shape(ko, (n,)))
    # both parties hold both opened values: each derives the next opening itself, E = y_0 + y_1 - a_0 - a_1 (no second round),
    # or - reveal - the result y_0 + y_1; out may be NULL
    ko1 = 0x0FEDCBA987654321
    with np.errstate(over="ignore"):
        e_pub = w0 + w1 - co.prng_shape(ko, (n,)) - co.prng_shape(ko1, (n,))
        for p, w in ((0, w0), (1, w1)):
            o = dev_empty(n); e = dev_empty(n)
            ctx.call("cognn_trunc_close_pub_u64", ptr(o), ptr(e), ptr(c0), ptr(c1), ctypes.byref(k), p, ctypes.c_uint64(ko), ctypes.c_uint64(ko1), 0, n)
            assert np.array_equal(host(o), w) and np.array_equal(host(e), e_pub)
        e = dev_empty(n)
        ctx.call("cognn_trunc_close_pub_u64", None, ptr(e), ptr(c0), ptr(c1), ctypes.byref(k), 0, ctypes.c_uint64(0), ctypes.c_uint64(0), 1, n)
        assert np.array_equal(host(e), w0 + w1)
        # the same close as a party fed by a real dealer runs it: the published t and the party's r' share arrive as tensors
        # (cognn_dealer_trunc_pub_u64 fills them), no stream key on the party's side
        for reveal, want_e in ((0, e_pub), (1, w0 + w1)):
            t = dev_empty(n); rp = [dev_empty(n), dev_empty(n)]
            ctx.call("cognn_dealer_trunc_pub_u64", ptr(t), ptr(rp[0]), ptr(rp[1]), ctypes.byref(k), ctypes.c_uint64(ko), ctypes.c_uint64(ko1), reveal, n)
            for p, w in ((0, w0), (1, w1)):
                o = dev_empty(n); e = dev_empty(n)
                ctx.call("cognn_trunc_close_pub_dealt_u64", ptr(o), ptr(e), ptr(c0), ptr(c1), ptr(t), ptr(rp[p]), p, n)
                assert np.array_equal(host(o), w) and np.array_equal(host(e), want_e)
            e = dev_empty(n)
            ctx.call("cognn_trunc_close_pub_dealt_u64", None, ptr(e), ptr(c0), ptr(c1), ptr(t), None, 0, n)
            assert np.array_equal(host(e), want_e)
    from cognn_amd import capi
    with pytest.raises(capi.CognnError, match="both opened values"):
        ctx.call("cognn_trunc_close_pub_u64", None, ptr(e), ptr(c0), None, ctypes.byref(k), 1, ctypes.c_uint64(ko), ctypes.c_uint64(ko1), 0, n)
    # mode 1 (apply gradient): out -= y, with a public multiplier
    W = rand_u64(rng, n); Wd0 = dev(W); Wd1 = dev(W)
    mul = 32768
    ctx.call("cognn_trunc_open_u64", ptr(c0), ptr(dev(x0)), ctypes.c_uint64(mul), ctypes.byref(k), 0, n)
    ctx.call("cognn_trunc_open_u64", ptr(c1), ptr(dev(x1)), ctypes.c_uint64(mul), ctypes.byref(k), 1, n)
    ctx.call("cognn_trunc_close_u64", ptr(Wd0), ptr(c0), ptr(c1), ctypes.byref(k), 0, 1, n)
    ctx.call("cognn_trunc_close_u64", ptr(Wd1), None, None, ctypes.byref(k), 1, 1, n)
    u0, u1 = co.const_scale_trunc_pair(x0, x1, mul, kf)
    with np.errstate(over="ignore"):
        assert np.array_equal(host(Wd0), W - u0) and np.array_equal(host(Wd1), W - u1)


@pytest.mark.parametrize("F", [64, 7])
def test_rowscale_trunc_pair(ctx, F):
    rng = np.random.default_rng(F)
    rows = 333
    v = co.fx_encode(rng.normal(size=(rows, F)) * 4)
    v0 = rand_u64(rng, (rows, F))
    with np.errstate(over="ignore"):
        v1 = v - v0
    s0 = co.normalizer(rng.integers(0, 30, size=rows)); s1 = np.zeros(rows, dtype=U64)
    k, kf = keys_of(1, 2, 3, co.OP_GA_SCALE)
    tk, tkf = keys_of(1, 2, 3, co.OP_GA_SCALE_TRUNC)
    E = [dev_empty((rows, F)) for _ in range(2)]; G = [dev_empty(rows) for _ in range(2)]
    for p, (vp, sp) in enumerate(((v0, s0), (v1, s1))):
        ctx.call("cognn_rowscale_open_u64", ptr(E[p]), ptr(G[p]), ptr(dev(vp)), ptr(dev(sp)), ctypes.byref(k), p, rows, F)
    c = [dev_empty((rows, F)) for _ in range(2)]
    for p in range(2):          # each side sums its own and the peer's opened shares inside the kernel
        ctx.call("cognn_rowscale_close_u64", ptr(c[p]), ptr(E[p]), ptr(E[1 - p]), ptr(G[p]), ptr(G[1 - p]), ctypes.byref(k),
                 ctypes.byref(tk), p, rows, F)
    o = [dev_empty((rows, F)) for _ in range(2)]
    ctx.call("cognn_trunc_close_u64", ptr(o[0]), ptr(c[0]), ptr(c[1]), ctypes.byref(tk), 0, 0, rows * F)
    ctx.call("cognn_trunc_close_u64", ptr(o[1]), None, None, ctypes.byref(tk), 1, 0, rows * F)
    z0, z1 = co.beaver_rowscale_pair(v0, v1, s0, s1, kf)
    w0, w1 = co.trunc_pair(z0, z1, tkf)
    assert np.array_equal(host(o[0]), w0) and np.array_equal(host(o[1]), w1)


def test_relu_pair(ctx):
    rng = np.random.default_rng(8)
    n = 5001
    z = co.fx_encode(rng.normal(size=n) * 10); z[:5] = 0
    z0 = rand_u64(rng, n)
    with np.errstate(over="ignore"):
        z1 = z - z0
    k, kf = keys_of(4, 0, 6, co.OP_AP_RELU)
    E = [dev_empty(n) for _ in range(2)]; G = [dev_empty(n) for _ in range(2)]
    zd = [dev(z0), dev(z1)]
    for p in range(2):
        ctx.call("cognn_relu_open_u64", ptr(E[p]), ptr(G[p]), ptr(zd[p]), ctypes.byref(k), p, n)
    w = [dev_empty(n) for _ in range(2)]
    for p in range(2):
        ctx.call("cognn_relu_mul_u64", ptr(w[p]), ptr(E[p]), ptr(E[1 - p]), ptr(G[p]), ptr(G[1 - p]), ctypes.byref(k), p, n)
    h = [dev_empty(n) for _ in range(2)]; mask = dev_empty(n, "u8")
    for p in range(2):
        ctx.call("cognn_relu_close_u64", ptr(h[p]), ptr(mask) if p == 0 else None, ptr(zd[p]), ptr(w[0]), ptr(w[1]), n)
    h0, h1, pos = co.relu_pair(z0, z1, kf)
    assert np.array_equal(host(h[0]), h0) and np.array_equal(host(h[1]), h1)
    assert np.array_equal(host(mask, np.uint8).astype(bool), pos)
    assert np.array_equal(pos, z.astype(np.int64) > 0)       # (|z t| >= 2^17 for these inputs: the 48-bit reading of w decides as the full sum does)
    # dealer-published g (no G opening online): same product shares, bit for bit
    E2 = [dev_empty(n) for _ in range(2)]; w2 = [dev_empty(n) for _ in range(2)]
    for p in range(2):
        ctx.call("cognn_relu_open_u64", ptr(E2[p]), None, ptr(zd[p]), ctypes.byref(k), p, n)
    for p in range(2):
        ctx.call("cognn_relu_mul_u64", ptr(w2[p]), ptr(E2[p]), ptr(E2[1 - p]), None, None, ctypes.byref(k), p, n)
    for p in range(2):
        assert np.array_equal(host(E2[p]), host(E[p])) and np.array_equal(host(w2[p]), host(w[p]))
    sel = dev_empty(n)
    ctx.call("cognn_mask_select_u64", ptr(sel), ptr(zd[0]), ptr(mask), n)
    assert np.array_equal(host(sel), np.where(pos, z0, U64(0)))


@pytest.mark.parametrize("L", [7, 16, 3])
def test_softmax_pair_and_metrics(ctx, L):
    rng = np.random.default_rng(L)
    rows, train = 400, 80
    z = co.fx_encode(rng.normal(size=(rows, L)) * 3)
    z[0] = co.fx_encode(np.array([40.0] + [-40.0] * (L - 1)))            # saturating row
    z0 = rand_u64(rng, (rows, L))
    with np.errstate(over="ignore"):
        z1 = z - z0
    labels = rng.integers(0, L, size=rows).astype(np.int32)
    k, kf = keys_of(5, 1, 1, co.OP_AP_SOFTMAX)
    p = [dev_empty((rows, L)) for _ in range(2)]; d = [dev_empty((rows, L)) for _ in range(2)]; pfx = dev_empty((rows, L))
    ctx.call("cognn_softmax_u64", ptr(p[0]), ptr(d[0]), ptr(pfx), ptr(dev(z0)), ptr(dev(z1)), ptr(dev(labels)),
             ctypes.byref(k), 0, rows, L, train)
    ctx.call("cognn_softmax_u64", ptr(p[1]), ptr(d[1]), None, None, None, None, ctypes.byref(k), 1, rows, L, train)
    p0, p1, d0, d1, plainP = co.softmax_pair(z0, z1, labels, train, kf)
    assert np.array_equal(host(p[0]), p0) and np.array_equal(host(p[1]), p1)
    assert np.array_equal(host(d[0]), d0) and np.array_equal(host(d[1]), d1)
    assert np.array_equal(host(pfx).astype(np.float64) / 65536, plainP)
    # integer softmax tracks the float softmax
    zf = co.fx_decode(z); e = np.exp(zf - zf.max(1, keepdims=True)); pf = e / e.sum(1, keepdims=True)
    assert np.abs(plainP - pf).max() < 1e-4
    border = (rng.random(rows) < 0.3).astype(np.uint8)
    val = 80
    counts = dev_empty(6); loss = dev_empty(1, "f64")
    ctx.call("cognn_metrics_q16", ptr(pfx), ptr(dev(labels)), ptr(dev(border)), rows, L, train, val, ptr(counts), ptr(loss))
    pp = np.where(plainP == 0, 0.001, plainP)
    ok = pp.argmax(1) == labels
    idx = np.arange(rows); tr = idx < train; te = idx >= train + val; b = border.astype(bool)
    cnt = host(counts, np.int64)
    assert list(cnt[:5]) == [ok.sum(), (ok & tr).sum(), (ok & tr & b).sum(), (ok & te).sum(), (ok & te & b).sum()]
    want_loss = -np.log(pp[idx, labels]).sum()
    assert abs(float(host(loss, np.float64)[0]) - want_loss) < 1e-9 * max(1.0, abs(want_loss))   # fp tolerance: atomics order


@pytest.mark.parametrize("L", [3, 7, 16])
def test_softmax_jobs_batch_with_fused_metrics(ctx, L):
    """cognn_softmax_jobs_u64: every hosted side's prediction layer in one launch, the owners' metrics fused in (no pfx
    tensor); against the oracle's softmax_pair and the metric definitions of gcn.h:611-632."""
    from cognn_amd import capi
    rng = np.random.default_rng(40 + L)
    sides = []
    for owner, rows in enumerate((400, 1, 513)):
        train, val = rows // 5, rows // 5
        z = co.fx_encode(rng.normal(size=(rows, L)) * 3)
        z0 = rand_u64(rng, (rows, L))
        with np.errstate(over="ignore"):
            z1 = z - z0
        labels = rng.integers(0, L, size=rows).astype(np.int32)
        border = (rng.random(rows) < 0.3).astype(np.uint8)
        sides.append(dict(owner=owner, rows=rows, train=train, val=val, z0=z0, z1=z1, labels=labels, border=border))
    jobs = (capi.SoftmaxJob * (2 * len(sides)))()
    outs = []
    for i, sd in enumerate(sides):
        k, kf = keys_of(5, sd["owner"], 1, co.OP_AP_SOFTMAX)
        sd["kf"] = kf
        for p in (0, 1):
            j = jobs[2 * i + p]
            d = dev_empty((sd["rows"], L)); outs.append(d)
            j.d_out = d.data_ptr(); j.keys = k; j.p = p; j.rows = sd["rows"]; j.train_rows = sd["train"]; j.val_rows = sd["val"]
            if p == 0:
                cnt, loss = dev_empty(6), dev_empty(1, "f64")
                sd["cnt"], sd["loss"] = cnt, loss
                if i == 1:                                  # z already revealed to the owner: z0 = z, no second share
                    with np.errstate(over="ignore"):
                        j.z0 = dev(sd["z0"] + sd["z1"]).data_ptr()
                    j.z1 = None
                else:
                    j.z0 = dev(sd["z0"]).data_ptr(); j.z1 = dev(sd["z1"]).data_ptr()
                j.labels = dev(sd["labels"]).data_ptr()
                j.border = dev(sd["border"]).data_ptr(); j.counts6 = cnt.data_ptr(); j.loss = loss.data_ptr()
    ctx.call("cognn_softmax_jobs_u64", jobs, len(jobs), L)
    for i, sd in enumerate(sides):
        p0, p1, d0, d1, plainP = co.softmax_pair(sd["z0"], sd["z1"], sd["labels"], sd["train"], sd["kf"])
        assert np.array_equal(host(outs[2 * i]), d0) and np.array_equal(host(outs[2 * i + 1]), d1)
        pp = np.where(plainP == 0, 0.001, plainP)
        ok = pp.argmax(1) == sd["labels"]
        idx = np.arange(sd["rows"]); tr = idx < sd["train"]; te = idx >= sd["train"] + sd["val"]; b = sd["border"].astype(bool)
        cnt = host(sd["cnt"], np.int64)
        assert list(cnt[:5]) == [ok.sum(), (ok & tr).sum(), (ok & tr & b).sum(), (ok & te).sum(), (ok & te & b).sum()]
        want_loss = -np.log(pp[idx, sd["labels"]]).sum()
        assert abs(float(host(sd["loss"], np.float64)[0]) - want_loss) < 1e-9 * max(1.0, abs(want_loss))   # fp tolerance: atomics order


@pytest.mark.parametrize("M,N,K,transA", [(300, 64, 128, 0), (270, 7, 16, 0), (16, 7, 300, 1), (1433, 16, 270, 1),
                                             (300, 16, 77, 0), (1354, 16, 1433, 0), (260, 7, 7, 0), (64, 16, 2000, 1),
                                             (128, 64, 1000, 2), (24, 5, 300, 2),
                                             # the register-direct kernels (N <= 16: any M; 16 < N <= 64: >= 1024 row tiles):
                                             (2048, 16, 64, 0), (2051, 13, 130, 0), (515, 16, 33, 0), (16384, 64, 128, 0),
                                             (16390, 33, 66, 0), (16400, 48, 35, 0), (16384, 32, 64, 0),
                                             # the register-direct TN kernel (N <= 16, K >= 256): ragged M / N / K, both mask addressings
                                             (64, 16, 4099, 1), (70, 3, 1000, 2), (128, 16, 515, 2), (500, 16, 4929, 1), (17, 1, 257, 1),
                                             # ... and its 2-, 3- and 4-column-tile forms (16 < N <= 64)
                                             (128, 64, 4100, 2), (100, 33, 700, 1), (130, 48, 1029, 2), (64, 17, 300, 1)])
def test_beaver_gemm_pair(ctx, M, N, K, transA):
    """Full Beaver product: mask-open, exchange, dealer C1, close; vs oracle twoPartyGCNMatMul stand-in."""
    rng = np.random.default_rng(M * 3 + N)
    X = rand_u64(rng, (M, K)); W = rand_u64(rng, (K, N))
    X0 = rand_u64(rng, (M, K)); W0 = rand_u64(rng, (K, N))
    with np.errstate(over="ignore"):
        X1 = X - X0; W1 = W - W0
    k, kf = keys_of(11, 2, 4, co.OP_PS_GEMM)
    stor = (lambda a: a.T.copy()) if transA else (lambda a: a)
    E = [dev_empty(stor(X0).shape) for _ in range(2)]; Fm = [dev_empty((K, N)) for _ in range(2)]
    for p, (xp, wp) in enumerate(((X0, W0), (X1, W1))):
        ctx.call("cognn_mask_open_u64", ptr(E[p]), ptr(dev(stor(xp))), ctypes.c_uint64(kf(co.SL_A0 + p)), M, K, transA | 16)   # | COGNN_MASK_OPEN_LIMB: a product's A mask
        ctx.call("cognn_mask_open_u64", ptr(Fm[p]), ptr(dev(wp)), ctypes.c_uint64(kf(co.SL_B0 + p)), K, N, 0)
    Fs = dev_empty((K, N))
    ctx.call("cognn_add_u64", ptr(Fs), ptr(Fm[0]), ptr(Fm[1]), K * N)
    c1 = dev_empty((M, N)); sa = dev_empty(M * K + K * N)
    ctx.call("cognn_dealer_gemm_c1_u64", ptr(c1), ctypes.byref(k), M, N, K, transA, ptr(sa), ctypes.c_void_p(sa.data_ptr() + 8 * M * K))
    Z = [dev_empty((M, N)) for _ in range(2)]
    for p in range(2):
        ctx.call("cognn_beaver_gemm_close_u64", ptr(Z[p]), ptr(E[p]), ptr(E[1 - p]), ptr(Fs), ptr(c1) if p == 1 else None,
                 ctypes.byref(k), p, M, N, K, transA, ptr(sa))
    z0, z1 = co.beaver_gemm_pair(X0, X1, W0, W1, kf, a_of_transposed=(transA == 2))
    assert np.array_equal(host(Z[0]), z0) and np.array_equal(host(Z[1]), z1)
    with np.errstate(over="ignore"):
        assert np.array_equal(host(Z[0]) + host(Z[1]), co.ring_matmul(X, W))
    # the same close with the opening of X handed over pre-summed (one operand stream: the engine's form for co-located pairs,
    # public openings and the cached feature opening - and the form the register-direct TN kernel takes for four column tiles)
    Es = dev_empty(stor(X0).shape)
    ctx.call("cognn_add_u64", ptr(Es), ptr(E[0]), ptr(E[1]), M * K)
    for p in range(2):
        Zs = dev_empty((M, N))
        ctx.call("cognn_beaver_gemm_close_u64", ptr(Zs), ptr(Es), None, ptr(Fs), ptr(c1) if p == 1 else None, ctypes.byref(k), p, M, N, K, transA, ptr(sa))
        assert np.array_equal(host(Zs), (z0, z1)[p])


@pytest.mark.parametrize("N,K,Ms,two,presplit", [(64, 128, (16384, 16400, 4096), False, False), (64, 128, (16384, 16384), False, True),
                                                   (16, 64, (20000, 12000, 800), True, False), (16, 16, (32768,), False, False),
                                                   (33, 70, (16390, 16390), False, True), (48, 35, (17000, 16000), True, False),
                                                   (7, 16, (270, 300), False, False),
                                                   # few row tiles / long K: workgroups take K ranges (split K), dataset shapes
                                                   (16, 1433, (1354, 1354), False, False), (16, 3703, (1656,), True, False), (16, 7, (1354, 1300), False, False),
                                                   (64, 1432, (600, 40), False, True), (3, 5, (100,), True, False), (48, 500, (4930, 4929), False, False),
                                                   # the fragment-ordered operand with an odd K (Cora / CiteSeer feature counts), split K and whole K
                                                   (16, 1433, (1354, 1354), False, True), (16, 3703, (1656, 1656), False, True), (16, 33, (40000, 33000), False, True),
                                                   # very short K (g = (p - y) . W1^T with 3 labels), PubMed's layer-0 shape (K ranges although the image would fit)
                                                   (16, 3, (4929, 4930), False, False), (16, 1, (300,), True, False), (7, 2, (40000,), False, False),
                                                   (16, 500, (4929, 4930, 4929), False, True),
                                                   # both halves of the A fragment as images (cognn_gemm_job::A_presplit): N > 16 whole-K uses them, the others ignore them
                                                   (64, 128, (16384, 16400), False, "masks"), (48, 96, (20000, 20001), False, "masks"), (16, 64, (40000,), False, "masks"),
                                                   (64, 1432, (600, 40), False, "masks")])
def test_beaver_gemm_group(ctx, N, K, Ms, two, presplit):
    """cognn_beaver_gemm_close_group_u64: the products of a phase as one grouped launch (weight planes built in the kernel's
    prologue; optionally the left operand pre-split in fragment order) against the oracle's Beaver product, job by job - jobs
    of different row counts, both share indices, ragged shapes, and a set too small for the grouped kernel (per-job path)."""
    from cognn_amd import capi
    rng = np.random.default_rng(N * 7 + K)
    jobs = (capi.GemmJob * (2 * len(Ms)))()
    keep, want, c1s = [], [], []
    for q, M in enumerate(Ms):
        X0 = rand_u64(rng, (M, K)); X1 = rand_u64(rng, (M, K)); W0 = rand_u64(rng, (K, N)); W1 = rand_u64(rng, (K, N))
        k, kf = keys_of(11, q, 4, co.OP_PS_GEMM)
        E = [dev_empty((M, K)) for _ in range(2)]; Fm = [dev_empty((K, N)) for _ in range(2)]
        for p, (xp, wp) in enumerate(((X0, W0), (X1, W1))):
            ctx.call("cognn_mask_open_u64", ptr(E[p]), ptr(dev(xp)), ctypes.c_uint64(kf(co.SL_A0 + p)), M, K, 16)
            ctx.call("cognn_mask_open_u64", ptr(Fm[p]), ptr(dev(wp)), ctypes.c_uint64(kf(co.SL_B0 + p)), K, N, 0)
        Es = dev_empty((M, K)); Fs = dev_empty((K, N))
        ctx.call("cognn_add_u64", ptr(Es), ptr(E[0]), ptr(E[1]), M * K)
        ctx.call("cognn_add_u64", ptr(Fs), ptr(Fm[0]), ptr(Fm[1]), K * N)
        c1 = dev_empty((M, N)); sa = dev_empty(M * K + K * N)
        ctx.call("cognn_dealer_gemm_c1_u64", ptr(c1), ctypes.byref(k), M, N, K, 0, ptr(sa), ctypes.c_void_p(sa.data_ptr() + 8 * M * K))
        img = None
        mimg = [None, None]
        if presplit:
            img = dev_empty(capi.load().cognn_gemm_presplit_bytes(M, K) // 8)
            ctx.call("cognn_gemm_presplit_u64", ptr(img), ptr(E[0]), ptr(E[1]), M, K)
        if presplit == "masks":                              # ... and each side's mask A_p in the same order (a mask that is dealt once)
            for p in range(2):
                tmp = dev_empty((M, K)); mimg[p] = dev_empty(capi.load().cognn_gemm_presplit_bytes(M, K) // 8)
                ctx.call("cognn_gemm_mask_fill_u64", ptr(tmp), ctypes.c_uint64(kf(co.SL_A0 + p)), M * K)
                ctx.call("cognn_gemm_presplit_u64", ptr(mimg[p]), ptr(tmp), None, M, K)
                keep.append(tmp)
        z0, z1 = co.beaver_gemm_pair(X0, X1, W0, W1, kf)
        for p in range(2):
            J = jobs[2 * q + p]
            Z = dev_empty((M, N)); scr = dev_empty(M * K + K * N)
            J.Z = Z.data_ptr()
            if two:
                J.E0, J.E1, J.F0, J.F1 = E[p].data_ptr(), E[1 - p].data_ptr(), Fm[p].data_ptr(), Fm[1 - p].data_ptr()
            else:
                J.E0, J.E1, J.F0, J.F1 = Es.data_ptr(), None, Fs.data_ptr(), None
            J.c1 = c1.data_ptr() if p == 1 else None
            J.keys = k; J.p = p; J.M = M; J.scratch = scr.data_ptr()
            J.E_presplit = img.data_ptr() if img is not None else None
            J.A_presplit = mimg[p].data_ptr() if mimg[p] is not None else None
            keep += [Z, scr, mimg[p]]; want.append((Z, (z0, z1)[p]))
        keep += [E, Fm, Es, Fs, c1, sa, img]
        c1s.append(c1)
    ctx.call("cognn_beaver_gemm_close_group_u64", jobs, len(jobs), N, K, 0)
    for Z, w in want:
        assert np.array_equal(host(Z), w)
    # raw: the same without C_p (the engine's form: C_p joins in the truncation opening)
    ctx.call("cognn_beaver_gemm_close_group_u64", jobs, len(jobs), N, K, 1)
    for q, M in enumerate(Ms):
        k, kf = keys_of(11, q, 4, co.OP_PS_GEMM)
        with np.errstate(over="ignore"):
            assert np.array_equal(host(want[2 * q][0]) + co.prng_shape(kf(co.SL_C0), (M, N)), want[2 * q][1])
            assert np.array_equal(host(want[2 * q + 1][0]) + host(c1s[q]), want[2 * q + 1][1])
    # Z_zeroed: the caller vouches for a clean output (its last reader cleared it: COGNN_PC_CLEAR_INPUT / COGNN_WU_CLEAR_Z).  The split-K
    # form then adds into Z without zeroing it first - shown by planting a marker: it survives in the result exactly when the launch
    # took K ranges (few row tiles / long K), and a clean Z gives the plain raw product either way
    raw = [host(Z).copy() for Z, _ in want]
    for j in range(len(jobs)):
        jobs[j].Z_zeroed = 1
        want[j][0].zero_()
    ctx.call("cognn_beaver_gemm_close_group_u64", jobs, len(jobs), N, K, 1)
    for j, (Z, _) in enumerate(want):
        assert np.array_equal(host(Z), raw[j])
    for Z, _ in want:
        Z.fill_(5)
    ctx.call("cognn_beaver_gemm_close_group_u64", jobs, len(jobs), N, K, 1)
    with np.errstate(over="ignore"):
        kept = [np.array_equal(host(Z), raw[j] + U64(5)) for j, (Z, _) in enumerate(want)]
        plain = [np.array_equal(host(Z), raw[j]) for j, (Z, _) in enumerate(want)]
    assert all(kept) or all(plain), (kept, plain)            # (split K: added onto the marker; whole K or per-job path: plain stores)


@pytest.mark.parametrize("M,N,Ks,stor_mask,images", [(128, 64, (4100, 4096, 0), 1, True), (130, 48, (1029, 515), 1, True), (100, 33, (700,), 0, True),
                                                        (128, 16, (5000, 4099), 1, True), (128, 64, (4100, 300), 1, False), (64, 17, (300, 257), 0, False)])
def test_beaver_gemm_group_tn(ctx, M, N, Ks, stor_mask, images):
    """cognn_beaver_gemm_close_group_tn_u64: the weight-gradient products d = h_t^T . in of a phase as one launch (per-job K, raw
    products), job by job against the oracle's Beaver product - and with both halves of the left operand handed over as
    fragment-ordered images (cognn_gemm_presplit_tn_u64: the constant feature tensor, whose mask is dealt once), which N > 16
    reads instead of the operand and the mask stream and N <= 16 ignores: bit-identical either way."""
    from cognn_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(M * 5 + N)
    jobs = (capi.GemmJob * (2 * len(Ks)))()
    keep, want = [], []
    for q, K in enumerate(Ks):
        k, kf = keys_of(13, q, 5, co.OP_AP_GEMM)
        Kc = max(K, 1)
        X0 = rand_u64(rng, (M, Kc)); X1 = rand_u64(rng, (M, Kc)); W0 = rand_u64(rng, (Kc, N)); W1 = rand_u64(rng, (Kc, N))
        E = [dev_empty((Kc, M)) for _ in range(2)]; Fm = [dev_empty((Kc, N)) for _ in range(2)]
        for p, (xp, wp) in enumerate(((X0, W0), (X1, W1))):     # the operand is stored [K x M]; transA 1: mask addressed (m, k), 2: (k, m)
            ctx.call("cognn_mask_open_u64", ptr(E[p]), ptr(dev(xp.T.copy())), ctypes.c_uint64(kf(co.SL_A0 + p)), M, Kc, (2 if stor_mask else 1) | 16)
            ctx.call("cognn_mask_open_u64", ptr(Fm[p]), ptr(dev(wp)), ctypes.c_uint64(kf(co.SL_B0 + p)), Kc, N, 0)
        Es = dev_empty((Kc, M)); Fs = dev_empty((Kc, N))
        ctx.call("cognn_add_u64", ptr(Es), ptr(E[0]), ptr(E[1]), M * Kc)
        ctx.call("cognn_add_u64", ptr(Fs), ptr(Fm[0]), ptr(Fm[1]), Kc * N)
        img = None
        if images and K > 0:
            nb = lib.cognn_gemm_presplit_tn_bytes(M, K)
            assert nb == ((M + 15) // 16) * ((K + 31) // 32) * 4096
            img = dev_empty(nb // 8)
            ctx.call("cognn_gemm_presplit_tn_u64", ptr(img), ptr(E[0]), ptr(E[1]), ctypes.c_uint64(0), stor_mask, M, K)
        if K > 0:
            z0, z1 = co.beaver_gemm_pair(X0, X1, W0, W1, kf, a_of_transposed=bool(stor_mask))
            with np.errstate(over="ignore"):
                raw = (z0 - co.prng_shape(kf(co.SL_C0), (M, N)), None)
        for p in range(2):
            J = jobs[2 * q + p]
            Z = dev_empty((M, N)); Z.fill_(3)                    # (the call zeroes it)
            J.Z = Z.data_ptr(); J.E0, J.E1, J.F0, J.F1 = Es.data_ptr(), None, Fs.data_ptr(), None
            J.keys = k; J.p = p; J.K = K
            mimg = None
            if img is not None:
                mimg = dev_empty(nb // 8)
                ctx.call("cognn_gemm_presplit_tn_u64", ptr(mimg), None, None, ctypes.c_uint64(kf(co.SL_A0 + p)), stor_mask, M, K)
                J.E_presplit = img.data_ptr(); J.A_presplit = mimg.data_ptr()
            keep += [Z, mimg]
            want.append((Z, K, raw[0] if (K > 0 and p == 0) else None, z1 if K > 0 else None, kf))
        keep += [E, Fm, Es, Fs, img]
    ctx.call("cognn_beaver_gemm_close_group_tn_u64", jobs, len(jobs), M, N, stor_mask)
    got = [host(Z) for Z, *_ in want]
    for j, (Z, K, r0, z1, kf) in enumerate(want):
        if K == 0:
            assert not got[j].any()
        elif j % 2 == 0:
            assert np.array_equal(got[j], r0)
        else:                                                # the p = 1 side's raw product + its dealt share C_1 = the oracle's share
            c1 = dev_empty((M, N)); sa = dev_empty(M * K + K * N)
            k, _ = keys_of(13, j // 2, 5, co.OP_AP_GEMM)
            ctx.call("cognn_dealer_gemm_c1_u64", ptr(c1), ctypes.byref(k), M, N, K, 2 if stor_mask else 1, ptr(sa), ctypes.c_void_p(sa.data_ptr() + 8 * M * K))
            with np.errstate(over="ignore"):
                assert np.array_equal(got[j] + host(c1), z1)
    if images:                                               # ... and the same launch without the images
        for j in range(len(jobs)):
            jobs[j].E_presplit = None; jobs[j].A_presplit = None
        ctx.call("cognn_beaver_gemm_close_group_tn_u64", jobs, len(jobs), M, N, stor_mask)
        for j, (Z, *_r) in enumerate(want):
            assert np.array_equal(host(Z), got[j])


def test_gather_csr_open_epilogue(ctx):
    """Gather whose output rows inside given segments are the Beaver opening V - prng(key, local index)."""
    rng = np.random.default_rng(77)
    n_rows, n_table, F = 300, 500, 16
    rowptr, col = _random_csr(rng, n_rows, n_table, 6)
    table = rand_u64(rng, (n_table, F)); base = rand_u64(rng, (n_rows, F))
    sb = np.array([0, 130], dtype=np.int64); se = np.array([100, 300], dtype=np.int64)
    sk = np.array([co.stream_key(1, 2, 3, 4, 0), co.stream_key(1, 5, 3, 4, 1)], dtype=np.uint64)
    out = dev_empty((n_rows, F))
    ctx.call("cognn_gather_csr_open_u64", ptr(out), ptr(dev(base)), ptr(dev(table)), ptr(dev(rowptr.view(np.int32))),
             ptr(dev(col.view(np.int32))), n_rows, F, 2, ctypes.c_void_p(sb.ctypes.data), ctypes.c_void_p(se.ctypes.data),
             ctypes.c_void_p(sk.ctypes.data))
    want = _csr_ref(base, table, rowptr, col)
    with np.errstate(over="ignore"):
        for b, e, k in zip(sb, se, sk):
            want[b:e] -= co.prng(int(k), (e - b) * F).reshape(e - b, F)
    assert np.array_equal(host(out), want)


def test_relu_close_open(ctx):
    rng = np.random.default_rng(78)
    n = 3001
    z = rand_u64(rng, n); w0 = rand_u64(rng, n); w1 = rand_u64(rng, n)
    key = co.stream_key(9, 9, 9, 9, 1)
    h, E, mask = dev_empty(n), dev_empty(n), dev_empty(n, "u8")
    ctx.call("cognn_relu_close_open_u64", ptr(h), ptr(E), ptr(mask), ptr(dev(z)), ptr(dev(w0)), ptr(dev(w1)), ctypes.c_uint64(key), n)
    with np.errstate(over="ignore"):
        pos = (co.open_hi48(w0, w1) << U64(16)).astype(np.int64) > 0      # the opened product read through its top 48 bits
        hw = np.where(pos, z, U64(0))
        assert np.array_equal(host(h), hw) and np.array_equal(host(E), hw - co.limb_value(co.prng(key, n)))   # the next product's A mask
    assert np.array_equal(host(mask, np.uint8).astype(bool), pos)


def test_launch_lanes_fork_and_join(ctx):
    """cognn_lane_begin / _select / _end: work issued on the lanes starts after what the context's stream held before the fork,
    and the stream continues only after every lane has finished (two dependent chains on two lanes, combined afterwards)."""
    from cognn_amd import capi
    n = 1 << 22
    keys = [co.stream_key(3, 1, 4, 1, s) for s in range(3)]
    base, a, b, out = dev_empty(n), dev_empty(n), dev_empty(n), dev_empty(n)
    ctx.call("cognn_prng_fill_u64", ptr(base), ctypes.c_uint64(keys[0]), n)          # before the fork: both lanes read it
    ctx.call("cognn_lane_begin", 2)
    for lane, (t, k) in enumerate(((a, keys[1]), (b, keys[2]))):
        ctx.call("cognn_lane_select", lane)
        ctx.call("cognn_prng_fill_u64", ptr(t), ctypes.c_uint64(k), n)
        for _ in range(3):
            ctx.call("cognn_add_u64", ptr(t), ptr(t), ptr(base), n)
    ctx.call("cognn_lane_end")
    ctx.call("cognn_add_u64", ptr(out), ptr(a), ptr(b), n)                           # after the join: reads both lanes' results
    with np.errstate(over="ignore"):
        want = co.prng(keys[1], n) + co.prng(keys[2], n) + U64(6) * co.prng(keys[0], n)
    assert np.array_equal(host(out), want)
    with pytest.raises(capi.CognnError, match="lane"):
        ctx.call("cognn_lane_select", 0)                                             # no lanes open
    with pytest.raises(capi.CognnError, match="lane"):
        ctx.call("cognn_lane_end")
    ctx.call("cognn_lane_begin", 1)
    with pytest.raises(capi.CognnError, match="lane"):
        ctx.call("cognn_lane_begin", 2)                                              # already open
    with pytest.raises(capi.CognnError, match="lane"):
        ctx.call("cognn_lane_select", 1)
    ctx.call("cognn_lane_end")


@pytest.mark.parametrize("count,n", [(1, 7), (3, 8192), (16, 100001)])
def test_sum_and_fanout(ctx, count, n):
    """cognn_sum_u64 / cognn_fanout_u64 (weight averaging: all hosted weight shares summed in one launch, the average handed
    back in one launch); in-place accumulation (output among the inputs) and the argument checks."""
    from cognn_amd import capi
    rng = np.random.default_rng(count * 31 + n)
    xs = [rand_u64(rng, n) for _ in range(count)]
    dx = [dev(x) for x in xs]
    ins = (ctypes.c_void_p * count)(*[t.data_ptr() for t in dx])
    out = dev_empty(n)
    ctx.call("cognn_sum_u64", ptr(out), ins, count, n)
    with np.errstate(over="ignore"):
        want = np.sum(np.stack(xs), axis=0, dtype=U64)
        assert np.array_equal(host(out), want)
        ins2 = (ctypes.c_void_p * 2)(out.data_ptr(), dx[0].data_ptr())               # running sum: out += x0
        ctx.call("cognn_sum_u64", ptr(out), ins2, 2, n)
        assert np.array_equal(host(out), want + xs[0])
    outs = [dev_empty(n) for _ in range(count)]
    op = (ctypes.c_void_p * count)(*[t.data_ptr() for t in outs])
    ctx.call("cognn_fanout_u64", op, count, ptr(dx[0]), n)
    for t in outs:
        assert np.array_equal(host(t), xs[0])
    with pytest.raises(capi.CognnError, match="1..16"):
        ctx.call("cognn_sum_u64", ptr(out), ins, 17, n)
    with pytest.raises(capi.CognnError, match="1..16"):
        ctx.call("cognn_fanout_u64", op, 0, ptr(dx[0]), n)


@pytest.mark.parametrize("n,C", [(100003, 3), (7, 5), (2, 2), (1 << 20, 8), (65, 64)])
def test_chunk_window_of_the_element_wise_entry_points(ctx, n, C):
    """cognn_ctx_set_chunk: the C windows of an element-wise call together do exactly what the unwindowed call does (dealer
    streams addressed by the absolute element index), single and batched launches; elements outside the window stay untouched;
    cognn_chunk_range is the partition both the kernels and the engine's messages use; every other entry point refuses to run
    under a window."""
    from cognn_amd import capi
    rng = np.random.default_rng(n + C)
    x = rand_u64(rng, n)
    ks, kf = keys_of(3, 1, 4, co.OP_GA_SCALE_TRUNC)
    whole = dev_empty(n)
    ctx.call("cognn_trunc_open_u64", ptr(whole), ptr(dev(x)), ctypes.c_uint64(5), ctypes.byref(ks), 1, n)
    ctx.sync()
    want = host(whole).copy()
    marker = np.full(n, 0xDEADBEEF, dtype=U64)
    out = dev(marker.copy()); out2 = dev(marker.copy())
    xd = dev(x)
    covered = 0
    lo = ctypes.c_int64(); hi = ctypes.c_int64()
    for c in range(C):
        ctx.lib.cognn_chunk_range(n, c, C, ctypes.byref(lo), ctypes.byref(hi))
        assert lo.value == covered and lo.value % 2 == 0 and hi.value >= lo.value
        covered = hi.value
        ctx.call("cognn_ctx_set_chunk", c, C)
        ctx.call("cognn_batch_begin")
        ctx.call("cognn_trunc_open_u64", ptr(out), ptr(xd), ctypes.c_uint64(5), ctypes.byref(ks), 1, n)
        ctx.call("cognn_trunc_open_u64", ptr(out2), ptr(xd), ctypes.c_uint64(5), ctypes.byref(ks), 1, n)
        ctx.call("cognn_batch_end")
        if c == 0:
            with pytest.raises(capi.CognnError, match="chunk window"):
                ctx.call("cognn_transpose_u64", ptr(out), ptr(xd), 1, n)
        ctx.call("cognn_ctx_set_chunk", 0, 1)
        ctx.sync()
        got = host(out)
        assert np.array_equal(got[:hi.value], want[:hi.value]) and np.all(got[hi.value:] == 0xDEADBEEF)
    assert covered == n
    assert np.array_equal(host(out), want) and np.array_equal(host(out2), want)
    single = dev(marker.copy())
    for c in reversed(range(C)):                             # unbatched launches, any order
        ctx.call("cognn_ctx_set_chunk", c, C)
        ctx.call("cognn_trunc_open_u64", ptr(single), ptr(xd), ctypes.c_uint64(5), ctypes.byref(ks), 1, n)
    ctx.call("cognn_ctx_set_chunk", 0, 1)
    ctx.sync()
    assert np.array_equal(host(single), want)
    with pytest.raises(capi.CognnError):
        ctx.call("cognn_ctx_set_chunk", C, C)


@pytest.mark.parametrize("N,K,Ms,scale,presplit", [(64, 128, (16400, 16384), False, True), (32, 64, (32768, 8200), True, False), (48, 96, (20000, 20001), True, False)])
def test_beaver_gemm_group_with_the_pair_chain_as_epilogue(ctx, N, K, Ms, scale, presplit):
    """cognn_gemm_job::epilogue: the p = 1 side's grouped product runs the pair's truncation (+ row scale) chain on its accumulator
    tiles, reading the p = 0 side's raw product written by an earlier call - against the plain sequence (both raw products, then
    cognn_pair_chain_u64), bit for bit; shapes outside the whole-K form are refused."""
    from cognn_amd import capi
    lib = capi.load()
    assert lib.cognn_beaver_gemm_group_takes_epilogue(N, K, sum((M + 15) // 16 for M in Ms)) == 1
    rng = np.random.default_rng(N + K)
    P = len(Ms)
    jobs0 = (capi.GemmJob * P)(); jobs1 = (capi.GemmJob * P)(); chains = (capi.PairChain * P)(); chains_ref = (capi.PairChain * P)()
    keep, outs, refs = [], [], []
    for q, M in enumerate(Ms):
        E = dev(rand_u64(rng, (M, K))); F = dev(rand_u64(rng, (K, N))); c1 = dev(rand_u64(rng, (M, N)) >> U64(8))
        s0 = dev(co.normalizer(rng.integers(0, 9, size=M))); s1 = dev(np.zeros(M, dtype=U64))
        k = capi.make_keys(11, q, 4, co.OP_PS_GEMM); tk = capi.make_keys(11, q, 4, co.OP_PS_GEMM_TRUNC)
        sk = capi.make_keys(11, q, 4, co.OP_PS_SCALE); stk = capi.make_keys(11, q, 4, co.OP_PS_SCALE_TRUNC)
        img = None
        if presplit:
            img = dev_empty(lib.cognn_gemm_presplit_bytes(M, K) // 8)
            ctx.call("cognn_gemm_presplit_u64", ptr(img), ptr(E), None, M, K)
        Z = [dev_empty((M, N)), dev_empty((M, N))]; scr = dev_empty(M * K + K * N)
        for p, J in ((0, jobs0[q]), (1, jobs1[q])):
            J.Z = Z[p].data_ptr(); J.E0 = E.data_ptr(); J.F0 = F.data_ptr(); J.keys = k; J.p = p; J.M = M; J.scratch = scr.data_ptr()
            J.E_presplit = img.data_ptr() if img is not None else None
        o = [dev_empty((M, N)) for _ in range(4)]
        for c, (o0, o1) in ((chains[q], o[:2]), (chains_ref[q], o[2:])):
            c.x[0] = Z[0].data_ptr(); c.x[1] = Z[1].data_ptr(); c.c1 = c1.data_ptr(); c.out[0] = o0.data_ptr(); c.out[1] = o1.data_ptr()
            c.scale[0] = s0.data_ptr(); c.scale[1] = s1.data_ptr()
            c.gemm_keys = k; c.trunc_in_keys = tk; c.scale_keys = sk; c.scale_trunc_keys = stk
            c.rows = M; c.F = N; c.flags = 1 | (2 if scale else 0)
        keep += [E, F, c1, s0, s1, img, Z, scr, o]
        outs.append(o[:2]); refs.append(o[2:])
    # reference: both raw products, then the chain launch
    ctx.call("cognn_beaver_gemm_close_group_u64", jobs0, P, N, K, 1)
    ctx.call("cognn_beaver_gemm_close_group_u64", jobs1, P, N, K, 1)
    ctx.call("cognn_pair_chain_u64", chains_ref, P)
    ctx.sync()
    z1_ref = [host(dev_z).copy() for dev_z in (keep[9 * q + 6][1] for q in range(P))]
    # epilogue form: the p = 1 products are never stored
    for q in range(P):
        keep[9 * q + 6][1].fill_(7)
        jobs1[q].epilogue = ctypes.addressof(chains[q])
    ctx.call("cognn_beaver_gemm_close_group_u64", jobs1, P, N, K, 1)
    ctx.sync()
    for q in range(P):
        assert np.array_equal(host(outs[q][0]), host(refs[q][0])) and np.array_equal(host(outs[q][1]), host(refs[q][1])), q
        assert np.all(host(keep[9 * q + 6][1]) == 7) and z1_ref[q].any()            # (Z of the p = 1 job was not written)
    jobs1[0].epilogue = None
    with pytest.raises(capi.CognnError, match="epilogue"):                          # on some jobs only
        ctx.call("cognn_beaver_gemm_close_group_u64", jobs1, P, N, K, 1)
    assert lib.cognn_beaver_gemm_group_takes_epilogue(32, 1433, 170) == 0           # the split-K shapes take none
    assert lib.cognn_beaver_gemm_group_takes_epilogue(16, 64, 65536) == 0           # ... nor one-column-tile products (VALU-bound: the chain is cheaper alone)


@pytest.mark.parametrize("N,K,Ms", [(16, 128, (40000, 39999)), (16, 1433, (1354, 1354)), (64, 128, (16384, 16400, 5)), (7, 16, (300,)), (3, 500, (4929, 4930, 4929, 4929)),
                                    (16, 3703, (1656,)), (64, 33, tuple([257] * 17)), (1, 1, (1, 0))])
def test_dealer_product_shares_as_one_grouped_launch(ctx, N, K, Ms):
    """cognn_dealer_gemm_c1_group_u64 (offline phase): C_1 = (A_0 + A_1) . (B_0 + B_1) - C_0 of several triples of one (N, K) in one
    launch of the grouped MFMA kernel - every operand generated in registers (the A masks in limb form), K ranges split over
    workgroups for the dataset shapes (Cora's 1354 x 1433, CiteSeer's 1656 x 3703) - against numpy and against the per-triple
    entry point; more than 16 triples are rejected (the engine launches 16 at a time)."""
    from cognn_amd import capi
    lib = capi.load()
    jobs = (capi.DealerJob * len(Ms))()
    outs, want = [], []
    for q, M in enumerate(Ms):
        k, kf = keys_of(23, q, 1, co.OP_PS_GEMM)
        c1 = dev_empty((M, N))
        if M:
            c1.fill_(0x5555)                                 # (the split-K form adds into the output: it must clear it itself)
        jobs[q].C1 = c1.data_ptr(); jobs[q].keys = k; jobs[q].M = M
        outs.append(c1)
        with np.errstate(over="ignore"):
            a = co.gemm_mask_shape(kf(co.SL_A0), (M, K)) + co.gemm_mask_shape(kf(co.SL_A1), (M, K))
            b = co.prng_shape(kf(co.SL_B0), (K, N)) + co.prng_shape(kf(co.SL_B1), (K, N))
            want.append(co.ring_matmul(a, b) - co.prng_shape(kf(co.SL_C0), (M, N)))
    if len(Ms) > 16:
        with pytest.raises(capi.CognnError):
            ctx.call("cognn_dealer_gemm_c1_group_u64", jobs, len(jobs), N, K)
        return
    assert lib.cognn_dealer_gemm_c1_groupable(N, K) == 1
    ctx.call("cognn_dealer_gemm_c1_group_u64", jobs, len(jobs), N, K)
    for q, M in enumerate(Ms):
        assert np.array_equal(host(outs[q]), want[q]), q
        if M:                                                # the per-triple entry point deals the same share
            one = dev_empty((M, N)); sa = dev_empty(M * K + K * N)
            ctx.call("cognn_dealer_gemm_c1_u64", ptr(one), ctypes.byref(jobs[q].keys), M, N, K, 0, ptr(sa), ctypes.c_void_p(sa.data_ptr() + 8 * M * K))
            assert np.array_equal(host(one), want[q]), q


@pytest.mark.parametrize("shapes", [[(128, 16, 4929, 2), (128, 16, 4930, 2), (16, 3, 4929, 1)], [(64, 7, 1354, 1)], [(33, 5, 300, 2), (5, 1, 1, 1), (8, 8, 0, 2)],
                                    [(500, 16, 4930, 2), (16, 3, 4930, 1), (500, 16, 4929, 2), (16, 3, 4929, 1), (500, 16, 255, 2), (500, 16, 256, 1)],
                                    [(128, 64, 9000, 2), (64, 16, 9000, 1), (128, 64, 8191, 2), (40, 33, 700, 1), (40, 48, 700, 2), (2, 2, 70000, 1)],
                                    [(24, 16, 3000 + 7 * q, 1 + q % 2) for q in range(16)]])
def test_dealer_product_shares_with_a_transposed_operand(ctx, shapes):
    """cognn_dealer_gemm_c1_tn_group_u64: the weight-gradient triples (A used transposed: transA 1 = logical mask index, 2 = the mask
    in storage order) - C_1 = -C_0 + (A_0 + A_1)^T-form product; the jobs of one (M, N) the register-direct kernel serves in ONE launch
    of its dealer form (every operand generated in registers), the others (K < 256) after a fill - against numpy and the per-triple call."""
    from cognn_amd import capi
    jobs = (capi.DealerTnJob * len(shapes))()
    keep, want = [], []
    for q, (M, N, K, tA) in enumerate(shapes):
        k, kf = keys_of(31, q, 2, co.OP_AP_GEMM)
        c1 = dev_empty((M, N)); sa = dev_empty(M * K + 2); sb = dev_empty(K * N + 2)
        jobs[q].C1 = c1.data_ptr(); jobs[q].keys = k; jobs[q].M = M; jobs[q].N = N; jobs[q].K = K; jobs[q].transA = tA
        jobs[q].scratchA = sa.data_ptr(); jobs[q].scratchB = sb.data_ptr()
        keep += [c1, sa, sb]
        with np.errstate(over="ignore"):
            if tA == 2:                                      # the mask of the [K x M] tensor, indexed in storage order
                a = (co.gemm_mask_shape(kf(co.SL_A0), (K, M)) + co.gemm_mask_shape(kf(co.SL_A1), (K, M))).T
            else:
                a = co.gemm_mask_shape(kf(co.SL_A0), (M, K)) + co.gemm_mask_shape(kf(co.SL_A1), (M, K))
            b = co.prng_shape(kf(co.SL_B0), (K, N)) + co.prng_shape(kf(co.SL_B1), (K, N))
            want.append(co.ring_matmul(a, b) - co.prng_shape(kf(co.SL_C0), (M, N)))
    ctx.call("cognn_dealer_gemm_c1_tn_group_u64", jobs, len(jobs))
    for q, (M, N, K, tA) in enumerate(shapes):
        assert np.array_equal(host(keep[3 * q]), want[q]), q
        if M * N:
            one = dev_empty((M, N)); sa = dev_empty(M * K + K * N + 2)
            ctx.call("cognn_dealer_gemm_c1_u64", ptr(one), ctypes.byref(jobs[q].keys), M, N, K, tA, ptr(sa), ctypes.c_void_p(sa.data_ptr() + 8 * M * K))
            assert np.array_equal(host(one), want[q]), q
