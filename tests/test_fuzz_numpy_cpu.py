"""The randomised differential suite's CPU side (tests/test_fuzz_gpu.py compares libcognn_hip.so with the plain-C++ backend
oracle/cpu_backend.cpp, which includes the same cognn_amd/csrc/cognn_spec.h as the kernels: a wrong constant there - PRNG
multipliers, key derivation, the softmax polynomial, the truncation offsets - would pass on both sides).  Here the C++ backend
is checked, on the same kind of ragged random shapes and for every family of that suite, against the numpy oracle
(oracle/cognn_oracle.py), which restates those definitions independently.  HIP == C++ backend (GPU suite) and
C++ backend == numpy (this file, no GPU needed) close the chain for the fuzzed shapes."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import cognn_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U64 = np.uint64
SEEDS = range(12)


@pytest.fixture(scope="module")
def cpu():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    from cognn_amd import capi
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libcognn_engine_cpu.so"))
    for name, (res, args) in capi._SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    return lib


def hp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def rand_u64(rng, shape):
    return rng.integers(0, 1 << 63, size=shape, dtype=np.int64).astype(U64) * U64(2) + rng.integers(0, 2, size=shape, dtype=np.int64).astype(U64)


def keys_of(seed, op):
    from cognn_amd import capi
    owner, it = seed % 5, seed % 7
    return capi.make_keys(seed, owner, it, op), (lambda slot: co.stream_key(seed, owner, it, op, slot))


def call(lib, name, *args):
    assert getattr(lib, name)(None, *args) == 0, name


@pytest.mark.parametrize("seed", SEEDS)
def test_prng_sharing_and_mask_streams(cpu, seed):
    rng = np.random.default_rng(100 + seed)
    rows, F = int(rng.integers(1, 200)), int(rng.choice([1, 2, 5, 16, 33]))
    n = rows * F
    key = int(rand_u64(rng, 1)[0])
    out = np.zeros(n, dtype=U64)
    call(cpu, "cognn_prng_fill_u64", hp(out), ctypes.c_uint64(key), n)
    assert np.array_equal(out, co.prng(key, n))
    fx = rand_u64(rng, n); s0 = np.zeros(n, dtype=U64); s1 = np.zeros(n, dtype=U64)
    call(cpu, "cognn_share_split_u64", hp(fx), ctypes.c_uint64(key), hp(s0), hp(s1), n)
    with np.errstate(over="ignore"):
        assert np.array_equal(s1, co.prng(key, n)) and np.array_equal(s0 + s1, fx)
    x = rand_u64(rng, (rows, F)); e = np.zeros((rows, F), dtype=U64)
    call(cpu, "cognn_mask_open_u64", hp(e), hp(x), ctypes.c_uint64(key), rows, F, 0)
    with np.errstate(over="ignore"):
        assert np.array_equal(e, x - co.prng_shape(key, (rows, F)))
        xt = np.ascontiguousarray(x.T); et = np.zeros((F, rows), dtype=U64)
        call(cpu, "cognn_mask_open_u64", hp(et), hp(xt), ctypes.c_uint64(key), rows, F, 1)     # stored transposed, logical mask index
        assert np.array_equal(et.T, x - co.prng_shape(key, (rows, F)))
        call(cpu, "cognn_mask_open_u64", hp(et), hp(xt), ctypes.c_uint64(key), rows, F, 2)     # storage-order mask index
        assert np.array_equal(et, xt - co.prng_shape(key, (F, rows)))
        # COGNN_MASK_OPEN_LIMB (16): the stream as a product's A mask - the signed-digit reading of every PRNG word
        call(cpu, "cognn_mask_open_u64", hp(e), hp(x), ctypes.c_uint64(key), rows, F, 16)
        assert np.array_equal(e, x - co.gemm_mask_shape(key, (rows, F)))
        call(cpu, "cognn_mask_open_u64", hp(et), hp(xt), ctypes.c_uint64(key), rows, F, 1 | 16)
        assert np.array_equal(et.T, x - co.gemm_mask_shape(key, (rows, F)))
        m = np.zeros(n, dtype=U64)
        call(cpu, "cognn_gemm_mask_fill_u64", hp(m), ctypes.c_uint64(key), n)
        w = co.prng(key, n)
        digits = w.view(np.int8).reshape(n, 8).astype(np.int64)            # little-endian bytes of w read as signed limbs
        want = np.zeros(n, dtype=U64)
        for i in range(8):
            want += (digits[:, i].astype(U64)) << U64(8 * i)
        assert np.array_equal(m, want) and np.array_equal(m, co.limb_value(w))
    # key derivation
    from cognn_amd import capi
    k = capi.make_keys(seed + 1, 3, 9, co.OP_AP_GEMM)
    for s in range(capi.NUM_SLOTS):
        assert k.k[s] == co.stream_key(seed + 1, 3, 9, co.OP_AP_GEMM, s)


@pytest.mark.parametrize("seed", SEEDS)
def test_gather_family(cpu, seed):
    rng = np.random.default_rng(1000 + seed)
    rows, table_rows = int(rng.integers(1, 300)), int(rng.integers(1, 400))
    F = int(rng.choice([1, 2, 3, 7, 16, 17, 64, 65]))
    deg = rng.poisson(rng.choice([0.5, 3, 12]), size=rows)
    rowptr = np.zeros(rows + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, table_rows, size=int(rowptr[-1]), dtype=np.uint32)
    table = rand_u64(rng, (table_rows, F)); base = rand_u64(rng, (rows, F))
    want = base.copy()
    with np.errstate(over="ignore"):
        for r in range(rows):
            for q in range(rowptr[r], rowptr[r + 1]):
                want[r] += table[col[q]]
    out = np.zeros((rows, F), dtype=U64)
    call(cpu, "cognn_gather_csr_u64", hp(out), hp(base), hp(table), hp(rowptr), hp(col), rows, F)
    assert np.array_equal(out, want)
    nseg = int(rng.integers(1, 4))
    cuts = np.sort(rng.integers(0, rows + 1, size=2 * nseg)).astype(np.int64)
    sb, se = cuts[0::2].copy(), cuts[1::2].copy()
    sk = rand_u64(rng, nseg)
    call(cpu, "cognn_gather_csr_open_u64", hp(out), hp(base), hp(table), hp(rowptr), hp(col), rows, F, ctypes.c_int32(nseg), hp(sb), hp(se), hp(sk))
    w2 = want.copy()
    with np.errstate(over="ignore"):
        for b, e, k in zip(sb, se, sk):
            if e > b:
                w2[b:e] -= co.prng(int(k), int(e - b) * F).reshape(int(e - b), F)
    assert np.array_equal(out, w2)


@pytest.mark.parametrize("seed", SEEDS)
def test_gemm_family(cpu, seed):
    rng = np.random.default_rng(2000 + seed)
    M = int(rng.choice([1, 5, 63, 64, 257, 300])); N = int(rng.choice([1, 3, 7, 16, 33, 64])); K = int(rng.choice([1, 4, 15, 16, 33, 77, 128]))
    tA = [0, 1, 2][seed % 3]
    X0, X1 = rand_u64(rng, (M, K)), rand_u64(rng, (M, K)); W0, W1 = rand_u64(rng, (K, N)), rand_u64(rng, (K, N))
    k, kf = keys_of(seed, co.OP_PS_GEMM)
    stor = (lambda a: np.ascontiguousarray(a.T)) if tA else (lambda a: a)
    E = [np.zeros(stor(X0).shape, dtype=U64) for _ in range(2)]; Fm = [np.zeros((K, N), dtype=U64) for _ in range(2)]
    for p, (xp, wp) in enumerate(((X0, W0), (X1, W1))):
        call(cpu, "cognn_mask_open_u64", hp(E[p]), hp(stor(xp)), ctypes.c_uint64(kf(co.SL_A0 + p)), M, K, tA | 16)     # | COGNN_MASK_OPEN_LIMB: a product's A mask
        call(cpu, "cognn_mask_open_u64", hp(Fm[p]), hp(wp), ctypes.c_uint64(kf(co.SL_B0 + p)), K, N, 0)
    c1 = np.zeros((M, N), dtype=U64); sa = np.zeros(M * K + K * N + 16, dtype=U64)
    call(cpu, "cognn_dealer_gemm_c1_u64", hp(c1), ctypes.byref(k), M, N, K, tA, hp(sa), ctypes.c_void_p(sa.ctypes.data + 8 * M * K))
    z0, z1 = co.beaver_gemm_pair(X0, X1, W0, W1, kf, a_of_transposed=(tA == 2))
    for p in range(2):
        Z = np.zeros((M, N), dtype=U64)
        call(cpu, "cognn_beaver_gemm_close2_u64", hp(Z), hp(E[p]), hp(E[1 - p]), hp(Fm[p]), hp(Fm[1 - p]), hp(c1) if p == 1 else None, ctypes.byref(k), p,
             M, N, K, tA, hp(sa), 0)
        assert np.array_equal(Z, (z0, z1)[p])
    with np.errstate(over="ignore"):
        assert np.array_equal(z0 + z1, co.ring_matmul(X0 + X1, W0 + W1))
    # plain ring product
    A = rand_u64(rng, (M, K)); B = rand_u64(rng, (K, N)); C = np.zeros((M, N), dtype=U64)
    call(cpu, "cognn_ring_gemm_u64", hp(C), hp(stor(A)), hp(B), M, N, K, 1 if tA else 0, 0)
    assert np.array_equal(C, co.ring_matmul(A, B))


@pytest.mark.parametrize("seed", SEEDS)
def test_elementwise_family(cpu, seed):
    rng = np.random.default_rng(3000 + seed)
    rows, F = int(rng.integers(1, 200)), int(rng.choice([1, 2, 5, 16, 33, 64]))
    n = rows * F
    # truncation (|x| < 2^61: the bounded-mask protocol's domain) with a public multiplier
    x = (rng.integers(-(1 << 40), 1 << 40, size=n)).astype(np.int64).astype(U64)
    x0 = rand_u64(rng, n)
    with np.errstate(over="ignore"):
        x1 = x - x0
    mul = int(rng.integers(1, 1 << 16))
    tk, tkf = keys_of(seed, co.OP_PS_GEMM_TRUNC)
    c = [np.zeros(n, dtype=U64) for _ in range(2)]
    for p, xp in enumerate((x0, x1)):
        call(cpu, "cognn_trunc_open_u64", hp(c[p]), hp(xp), ctypes.c_uint64(mul), ctypes.byref(tk), p, n)
    y = [np.zeros(n, dtype=U64) for _ in range(2)]
    call(cpu, "cognn_trunc_close_u64", hp(y[0]), hp(c[0]), hp(c[1]), ctypes.byref(tk), 0, 0, n)
    call(cpu, "cognn_trunc_close_u64", hp(y[1]), None, None, ctypes.byref(tk), 1, 0, n)
    w0, w1 = co.const_scale_trunc_pair(x0, x1, mul, tkf)
    assert np.array_equal(y[0], w0) and np.array_equal(y[1], w1)
    # row scale + truncation
    V = (rng.integers(-(1 << 30), 1 << 30, size=(rows, F))).astype(np.int64).astype(U64)
    V0 = rand_u64(rng, (rows, F))
    s = co.fx_encode(rng.random(rows))
    with np.errstate(over="ignore"):
        V1 = V - V0
    sk, skf = keys_of(seed + 1, co.OP_GA_SCALE)
    E = [np.zeros((rows, F), dtype=U64) for _ in range(2)]; G = [np.zeros(rows, dtype=U64) for _ in range(2)]
    zeros = np.zeros(rows, dtype=U64)
    for p, (vp, sp) in enumerate(((V0, s), (V1, zeros))):
        call(cpu, "cognn_rowscale_open_u64", hp(E[p]), hp(G[p]), hp(vp), hp(sp), ctypes.byref(sk), p, rows, F)
    cc = [np.zeros((rows, F), dtype=U64) for _ in range(2)]
    for p in range(2):
        call(cpu, "cognn_rowscale_close_u64", hp(cc[p]), hp(E[p]), hp(E[1 - p]), hp(G[p]), hp(G[1 - p]), ctypes.byref(sk), ctypes.byref(tk), p, rows, F)
    yy = [np.zeros((rows, F), dtype=U64) for _ in range(2)]
    call(cpu, "cognn_trunc_close_u64", hp(yy[0]), hp(cc[0]), hp(cc[1]), ctypes.byref(tk), 0, 0, n)
    call(cpu, "cognn_trunc_close_u64", hp(yy[1]), None, None, ctypes.byref(tk), 1, 0, n)
    z0, z1 = co.beaver_rowscale_pair(V0, V1, s, zeros, skf)
    w0, w1 = co.trunc_pair(z0, z1, tkf)
    assert np.array_equal(yy[0], w0) and np.array_equal(yy[1], w1)
    # masked-sign ReLU
    z = (rng.normal(size=n) * (1 << 18)).astype(np.int64).astype(U64)
    zz0 = rand_u64(rng, n)
    with np.errstate(over="ignore"):
        zz1 = z - zz0
    rk, rkf = keys_of(seed + 2, co.OP_AP_RELU)
    Er = [np.zeros(n, dtype=U64) for _ in range(2)]
    for p, zp in enumerate((zz0, zz1)):
        call(cpu, "cognn_relu_open_u64", hp(Er[p]), None, hp(zp), ctypes.byref(rk), p, n)
    w = [np.zeros(n, dtype=U64) for _ in range(2)]
    for p in range(2):
        call(cpu, "cognn_relu_mul_u64", hp(w[p]), hp(Er[p]), hp(Er[1 - p]), None, None, ctypes.byref(rk), p, n)
    h0w, h1w, pos = co.relu_pair(zz0, zz1, rkf)
    for p, (zp, hw) in enumerate(((zz0, h0w), (zz1, h1w))):
        h = np.zeros(n, dtype=U64); m = np.zeros(n, dtype=np.uint8)
        call(cpu, "cognn_relu_close_u64", hp(h), hp(m), hp(zp), hp(w[0]), hp(w[1]), n)
        assert np.array_equal(h, hw) and np.array_equal(m.astype(bool), pos)
    assert np.array_equal(pos, z.astype(np.int64) > 0)


@pytest.mark.parametrize("seed", SEEDS)
def test_softmax_family(cpu, seed):
    rng = np.random.default_rng(4000 + seed)
    rows = int(rng.integers(1, 300)); L = int(rng.choice([2, 3, 7, 16, 33])); train = int(rng.integers(0, rows + 1))
    z = (rng.normal(size=(rows, L)) * (1 << 18)).astype(np.int64).astype(U64)
    z0 = rand_u64(rng, (rows, L))
    with np.errstate(over="ignore"):
        z1 = z - z0
    labels = rng.integers(0, L, size=rows, dtype=np.int32)
    k, kf = keys_of(seed, co.OP_AP_SOFTMAX)
    p0, p1, d0, d1, plain = co.softmax_pair(z0, z1, labels, train, kf)
    P = np.zeros((rows, L), dtype=U64); D = np.zeros((rows, L), dtype=U64); PF = np.zeros((rows, L), dtype=U64)
    call(cpu, "cognn_softmax_u64", hp(P), hp(D), hp(PF), hp(z0), hp(z1), hp(labels), ctypes.byref(k), 0, rows, L, train)
    assert np.array_equal(P, p0) and np.array_equal(D, d0) and np.array_equal(PF.astype(np.float64) / 65536.0, plain)
    call(cpu, "cognn_softmax_u64", hp(P), hp(D), None, None, None, None, ctypes.byref(k), 1, rows, L, train)
    assert np.array_equal(P, p1) and np.array_equal(D, d1)


@pytest.mark.parametrize("seed", SEEDS)
def test_gather_with_prediction_layer_family(cpu, seed):
    """cognn_gather_pair_chain_u64 with cognn_gather_pair::softmax on the C++ backend (the checker of tests/test_fuzz_gpu.py's family of
    the same name) against numpy: plain aggregate, the oracle's row scale + truncation, softmax_pair and the metric definitions."""
    from cognn_amd import capi
    rng = np.random.default_rng(9000 + seed)
    F = int(rng.choice([1, 2, 3, 6, 7, 16, 16, 31, 40, 64]))
    npairs = int(rng.integers(1, 4))
    sizes = [int(rng.integers(1, 150)) for _ in range(npairs)]
    offs, off = [], 0
    for n in sizes:
        a = off; off = (off + n + 1) & ~1
        b = off; off = (off + n + 1) & ~1
        offs.append((a, b))
    T = off
    deg = rng.poisson(rng.choice([0.5, 4, 12]), size=T)
    rowptr = np.zeros(T + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, T, size=int(rowptr[-1]), dtype=np.uint32)
    table = rng.integers(-(1 << 16), 1 << 16, size=(T, F)).astype(np.int64).astype(U64)
    scale = bool(seed % 4 != 3)
    pairs = (capi.GatherPair * npairs)()
    jobs = (capi.SoftmaxJob * (2 * npairs))()
    keep = []
    for i, (n, (a, b)) in enumerate(zip(sizes, offs)):
        s0 = rand_u64(rng, n) >> U64(44); s1 = np.zeros(n, dtype=U64)
        labels = rng.integers(0, F, size=n, dtype=np.int32); border = (rng.random(n) < 0.4).astype(np.uint8)
        train = int(rng.integers(0, n + 1)); val = int(rng.integers(0, n - train + 1))
        d0, d1 = np.zeros((n, F), U64), np.zeros((n, F), U64)
        cnt, loss = np.full(6, 9, np.int64), np.full(1, 2.5)
        ks, kfs = keys_of(seed + i, co.OP_GA_SCALE); kt, kft = keys_of(seed + i, co.OP_GA_SCALE_TRUNC); km, kfm = keys_of(seed + i, co.OP_AP_SOFTMAX)
        p = pairs[i]
        p.a_row0 = a; p.b_row0 = b
        c = p.chain
        c.rows = n; c.F = F; c.flags = 2 if scale else 0
        c.scale[0] = s0.ctypes.data; c.scale[1] = s1.ctypes.data
        c.scale_keys = ks; c.scale_trunc_keys = kt
        for q in (0, 1):
            j = jobs[2 * i + q]
            j.d_out = (d0 if q == 0 else d1).ctypes.data; j.keys = km; j.p = q; j.rows = n; j.train_rows = train; j.val_rows = val
            if q == 0:
                j.labels = labels.ctypes.data; j.border = border.ctypes.data; j.counts6 = cnt.ctypes.data; j.loss = loss.ctypes.data
            p.softmax[q] = ctypes.addressof(j)
        keep.append((n, a, b, s0, s1, labels, border, train, val, d0, d1, cnt, loss, kfs, kft, kfm))
    call(cpu, "cognn_gather_pair_chain_u64", hp(table), hp(rowptr), hp(col), F, pairs, npairs)
    agg = table.copy()
    with np.errstate(over="ignore"):
        for r in range(T):
            for q in range(rowptr[r], rowptr[r + 1]):
                agg[r] += table[col[q]]
    for n, a, b, s0, s1, labels, border, train, val, d0, d1, cnt, loss, kfs, kft, kfm in keep:
        v0, v1 = agg[a:a + n], agg[b:b + n]
        if scale:
            z0, z1 = co.beaver_rowscale_pair(v0, v1, s0, s1, kfs)
            v0, v1 = co.trunc_pair(z0, z1, kft)
        _, _, w0, w1, plain = co.softmax_pair(v0, v1, labels, train, kfm)
        assert np.array_equal(d0, w0) and np.array_equal(d1, w1)
        pp = np.where(plain == 0, 0.001, plain)
        ok = pp.argmax(1) == labels
        idx = np.arange(n); tr = idx < train; te = idx >= train + val; bd = border.astype(bool)
        assert list(cnt[:5]) == [ok.sum(), (ok & tr).sum(), (ok & tr & bd).sum(), (ok & te).sum(), (ok & te & bd).sum()]
        want = -np.log(pp[idx, labels]).sum()
        assert abs(float(loss[0]) - want) <= 1e-9 * max(1.0, abs(want))
