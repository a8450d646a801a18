"""Randomised differential test at the C ABI: every compute entry point of libcognn_hip.so against the plain-C++ reference
backend (oracle/libcognn_engine_cpu.so - test infrastructure) on random shapes, bit for bit.  Complements the fixed-shape
oracle tests: ragged widths, odd row counts, shapes on both sides of every kernel-selection threshold."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from gpu_util import dev, dev_empty, host, ptr, rand_u64, U64

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALE = int(os.environ.get("COGNN_FUZZ_SCALE", "1"))        # soak runs: COGNN_FUZZ_SCALE=20 multiplies the number of seeds


@pytest.fixture(scope="module")
def libs():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from cognn_amd import capi
    path = os.path.join(ROOT, "oracle", "libcognn_engine_cpu.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    cpu = ctypes.CDLL(path)
    for name, (res, args) in capi._SIGNATURES.items():
        if hasattr(cpu, name):
            fn = getattr(cpu, name)
            fn.restype = res
            fn.argtypes = args
    c = capi.Context(0)
    yield c, cpu
    c.close()


def hp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def keys(seed, op=10):
    from cognn_amd import capi
    return capi.make_keys(seed, seed % 5, seed % 7, op)


def both(libs, name, outs, args):
    """args: list of values; numpy arrays are inputs (uploaded for the GPU call), ("out", i) refers to outs[i] (shape, dtype)."""
    ctx, cpu = libs
    gouts = [dev_empty(o[0].shape, "u8" if o[0].dtype == np.uint8 else "u64") if o[1] is None else dev(o[1]) for o in outs]
    couts = [np.zeros(o[0].shape, dtype=o[0].dtype) if o[1] is None else o[1].copy() for o in outs]
    ga, ca = [], []
    for a in args:
        if isinstance(a, tuple) and a[0] == "out":
            ga.append(ptr(gouts[a[1]])); ca.append(hp(couts[a[1]]))
        elif isinstance(a, np.ndarray):
            ga.append(ptr(dev(a))); ca.append(hp(a))
        else:
            ga.append(a); ca.append(a)
    ctx.call(name, *ga)
    rc = getattr(cpu, name)(None, *ca)
    assert rc == 0, name
    for g, c, o in zip(gouts, couts, outs):
        if not o[2]:
            continue                                        # scratch: contents are implementation-defined
        got = host(g, np.uint8) if c.dtype == np.uint8 else host(g)
        assert np.array_equal(got.reshape(c.shape), c), name
    return couts


def O(shape, dtype=U64, init=None, check=True):
    return (np.zeros(shape, dtype=dtype), init, check)


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_fuzz_gather(libs, seed):
    rng = np.random.default_rng(1000 + seed)
    rows = int(rng.integers(1, 700)); table_rows = int(rng.integers(1, 900))
    F = int(rng.choice([1, 2, 3, 6, 7, 8, 16, 17, 32, 62, 64, 65, 128, 130]))
    deg = rng.poisson(rng.choice([0.5, 3, 20]), size=rows)
    if seed % 4 == 0:
        deg[int(rng.integers(0, rows))] = 3000                      # one power-law row beyond the LDS slice
    rowptr = np.zeros(rows + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, table_rows, size=int(rowptr[-1]), dtype=np.uint32)
    table = rand_u64(rng, (table_rows, F)); base = rand_u64(rng, (rows, F))
    both(libs, "cognn_gather_csr_u64", [O((rows, F))], [("out", 0), base if seed % 2 else None, table, rowptr, col, rows, F])
    nseg = int(rng.integers(1, 5))
    cuts = np.sort(rng.integers(0, rows + 1, size=2 * nseg)).astype(np.int64)
    sb, se = cuts[0::2].copy(), cuts[1::2].copy()
    sk = rand_u64(rng, nseg)
    both(libs, "cognn_gather_csr_open_u64", [O((rows, F))],
         [("out", 0), base, table, rowptr, col, rows, F, ctypes.c_int32(nseg), sb, se, sk])
    n = int(rng.integers(1, 300)); idx = rng.permutation(rows)[: min(n, rows)].astype(np.uint32)
    part = rand_u64(rng, (len(idx), F))
    both(libs, "cognn_scatter_add_rows_u64", [O((rows, F), init=base)], [("out", 0), part, idx, len(idx), F])


@pytest.mark.parametrize("seed", range(40 * SCALE))
def test_fuzz_gemm(libs, seed):
    rng = np.random.default_rng(2000 + seed)
    M = int(rng.choice([1, 5, 63, 64, 255, 256, 257, 300, 1000, 2048, 4097]))
    N = int(rng.choice([1, 3, 7, 16, 31, 32, 33, 64, 65, 100]))
    K = int(rng.choice([1, 4, 7, 15, 16, 17, 32, 64, 77, 128, 300, 513]))
    tA = [1, 0, 2, 0][seed % 4]                                        # 2: transposed storage, mask streams in storage order
    A = rand_u64(rng, (K, M) if tA else (M, K)); A2 = rand_u64(rng, A.shape); B = rand_u64(rng, (K, N)); C0 = rand_u64(rng, (M, N))
    both(libs, "cognn_ring_gemm_u64", [O((M, N), init=C0)], [("out", 0), A, B, M, N, K, tA, 1])
    both(libs, "cognn_ring_gemm2_u64", [O((M, N))], [("out", 0), A, A2, B, M, N, K, tA, 0])
    k = keys(seed)
    sa = np.zeros(M * K + K * N + 16, dtype=U64)
    # dealer product share: the two scratch operands are distinct regions of one buffer
    ctx, cpu = libs
    gs = dev_empty(sa.shape); gc1 = dev_empty((M, N)); cc1 = np.zeros((M, N), dtype=U64); cs = np.zeros_like(sa)
    ctx.call("cognn_dealer_gemm_c1_u64", ptr(gc1), ctypes.byref(k), M, N, K, tA, ptr(gs), ctypes.c_void_p(gs.data_ptr() + 8 * M * K))
    assert cpu.cognn_dealer_gemm_c1_u64(None, hp(cc1), ctypes.byref(k), M, N, K, tA, hp(cs), ctypes.c_void_p(cs.ctypes.data + 8 * M * K)) == 0
    assert np.array_equal(host(gc1), cc1)
    E1 = rand_u64(rng, A.shape)
    for p in (0, 1):
        both(libs, "cognn_beaver_gemm_close_u64", [O((M, N)), O(sa.shape, check=False)],
             [("out", 0), A, E1 if seed % 2 else None, B, cc1 if p == 1 else None, ctypes.byref(k), p, M, N, K, tA, ("out", 1)])
        B2 = rand_u64(rng, (K, N))                            # F given as two shares
        both(libs, "cognn_beaver_gemm_close2_u64", [O((M, N)), O(sa.shape, check=False)],
             [("out", 0), A, E1, B, B2 if seed % 3 else None, cc1 if p == 1 else None, ctypes.byref(k), p, M, N, K, tA, ("out", 1), 0])
        if ctx.lib.cognn_beaver_gemm_fusable(M, N, K, tA):
            both(libs, "cognn_beaver_gemm_close2_u64", [O((M, N)), O(sa.shape, check=False)],
                 [("out", 0), A, E1, B, B2, None, ctypes.byref(k), p, M, N, K, tA, ("out", 1), 1])
            both(libs, "cognn_beaver_gemm_close_raw_u64", [O((M, N)), O(sa.shape, check=False)],
                 [("out", 0), A, E1, B, ctypes.byref(k), p, M, N, K, ("out", 1)])


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_fuzz_elementwise(libs, seed):
    rng = np.random.default_rng(3000 + seed)
    rows = int(rng.integers(1, 400)); F = int(rng.choice([1, 2, 5, 16, 33, 64]))
    n = rows * F
    k, tk = keys(seed, 12), keys(seed + 1, 13)
    x = rand_u64(rng, n); y = rand_u64(rng, n); c1 = rand_u64(rng, n)
    for p in (0, 1):
        c = both(libs, "cognn_trunc_open_u64", [O(n)], [("out", 0), x, ctypes.c_uint64(int(rng.integers(1, 1 << 20))), ctypes.byref(k), p, n])[0]
        both(libs, "cognn_trunc_open_add_u64", [O(n)], [("out", 0), x, c1 if p == 1 else None, ctypes.byref(k), ctypes.byref(tk), p, n])
        both(libs, "cognn_trunc_close_u64", [O(n, init=y)], [("out", 0), c if p == 0 else None, x if p == 0 else None, ctypes.byref(tk), p, seed % 2, n])
        both(libs, "cognn_trunc_close_open_u64", [O(n), O(n)],
             [("out", 0), ("out", 1), c if p == 0 else None, x if p == 0 else None, ctypes.byref(tk), p, ctypes.c_uint64(0x9E3779B97F4A7C15 + seed), n])
        V = rand_u64(rng, (rows, F)); s = rand_u64(rng, rows)
        eg = both(libs, "cognn_rowscale_open_u64", [O((rows, F)), O(rows)], [("out", 0), ("out", 1), V, s, ctypes.byref(k), p, rows, F])
        E1 = rand_u64(rng, (rows, F)); G1 = rand_u64(rng, rows)
        both(libs, "cognn_rowscale_close_u64", [O((rows, F))],
             [("out", 0), eg[0], E1, eg[1], G1 if seed % 2 else None, ctypes.byref(k), ctypes.byref(tk), p, rows, F])
        e = both(libs, "cognn_relu_open_u64", [O(n), O(n)], [("out", 0), ("out", 1), x, ctypes.byref(k), p, n])
        both(libs, "cognn_relu_mul_u64", [O(n)], [("out", 0), e[0], y, e[1], c1, ctypes.byref(k), p, n])
        both(libs, "cognn_relu_mul_u64", [O(n)], [("out", 0), e[0], y, None, None, ctypes.byref(k), p, n])
    both(libs, "cognn_relu_close_u64", [O(n), O(n, np.uint8)], [("out", 0), ("out", 1), x, y, c1, n])
    both(libs, "cognn_relu_close_open_u64", [O(n), O(n), O(n, np.uint8)], [("out", 0), ("out", 1), ("out", 2), x, y, c1, ctypes.c_uint64(77 + seed), n])
    mask = rng.integers(0, 2, size=n, dtype=np.uint8)
    both(libs, "cognn_mask_select_u64", [O(n)], [("out", 0), x, mask, n])
    both(libs, "cognn_mask_open_u64", [O((rows, F))], [("out", 0), x.reshape(rows, F), ctypes.c_uint64(99 + seed), rows, F, 0])
    both(libs, "cognn_mask_open_u64", [O((F, rows))], [("out", 0), x.reshape(F, rows), ctypes.c_uint64(99 + seed), rows, F, 1])
    both(libs, "cognn_transpose_u64", [O((F, rows))], [("out", 0), x.reshape(rows, F), rows, F])
    both(libs, "cognn_add_u64", [O(n)], [("out", 0), x, y, n])
    both(libs, "cognn_sub_u64", [O(n)], [("out", 0), x, y, n])
    both(libs, "cognn_prng_fill_u64", [O(n)], [("out", 0), ctypes.c_uint64(5 + seed), n])


@pytest.mark.parametrize("seed", range(6 * SCALE))
def test_fuzz_softmax(libs, seed):
    rng = np.random.default_rng(4000 + seed)
    rows = int(rng.integers(1, 500)); L = int(rng.choice([2, 3, 7, 16, 33, 40])); train = int(rng.integers(0, rows + 1))
    z = (rng.normal(size=(rows, L)) * (1 << 18)).astype(np.int64).astype(U64)
    z0 = rand_u64(rng, (rows, L))
    with np.errstate(over="ignore"):
        z1 = z - z0
    labels = rng.integers(0, L, size=rows, dtype=np.int32)
    k = keys(seed, 17)
    both(libs, "cognn_softmax_u64", [O((rows, L)), O((rows, L)), O((rows, L))],
         [("out", 0), ("out", 1), ("out", 2), z0, z1, labels, ctypes.byref(k), 0, rows, L, train])
    both(libs, "cognn_softmax_u64", [O((rows, L)), O((rows, L))], [("out", 0), ("out", 1), None, None, None, None, ctypes.byref(k), 1, rows, L, train])


@pytest.mark.parametrize("seed", range(10 * SCALE))
def test_fuzz_gather_with_prediction_layer(libs, seed):
    """cognn_gather_pair_chain_u64 with cognn_gather_pair::softmax (Gather + scale + prediction layer in one launch) against the
    reference backend, which runs the plain sequence (per-side gathers, per-side chain steps, cognn_softmax_u64, cognn_metrics_q16)."""
    from cognn_amd import capi
    ctx, cpu = libs
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
    if seed % 3 == 0:
        deg[offs[0][0]] = 1800                                    # beyond the staged slice of its tile
    rowptr = np.zeros(T + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, T, size=int(rowptr[-1]), dtype=np.uint32)
    table = rng.integers(-(1 << 16), 1 << 16, size=(T, F)).astype(np.int64).astype(U64)   # aggregates: logits a few units apart
    scale = bool(seed % 4 != 3)
    write_logits = bool(seed % 2)
    gp, cp = (capi.GatherPair * npairs)(), (capi.GatherPair * npairs)()
    gj, cj = (capi.SoftmaxJob * (2 * npairs))(), (capi.SoftmaxJob * (2 * npairs))()
    keep, outs = [], []
    for i, (n, (a, b)) in enumerate(zip(sizes, offs)):
        s0 = rand_u64(rng, n) >> U64(44); s1 = np.zeros(n, dtype=U64)      # a row scale below 2^20 (Q16), held by the owner
        labels = rng.integers(0, F, size=n, dtype=np.int32); border = (rng.random(n) < 0.4).astype(np.uint8)
        train = int(rng.integers(0, n + 1)); val = int(rng.integers(0, n - train + 1))
        host_bufs = dict(out0=np.zeros((n, F), U64), out1=np.zeros((n, F), U64), d0=np.zeros((n, F), U64), d1=np.zeros((n, F), U64),
                         cnt=np.full(6, 9, np.int64), loss=np.full(1, 2.5))
        dev_bufs = dict(out0=dev_empty((n, F)), out1=dev_empty((n, F)), d0=dev_empty((n, F)), d1=dev_empty((n, F)), cnt=dev(host_bufs["cnt"]),
                        loss=dev(host_bufs["loss"]))
        dS0, dS1, dL, dB = dev(s0), dev(s1), dev(labels), dev(border)
        for P, J, addr, ins in ((gp, gj, (lambda t: t.data_ptr()), dict(s0=dS0, s1=dS1, lab=dL, bd=dB, **dev_bufs)),
                                (cp, cj, (lambda t: t.ctypes.data), dict(s0=s0, s1=s1, lab=labels, bd=border, **host_bufs))):
            p = P[i]
            p.a_row0 = a; p.b_row0 = b
            c = p.chain
            c.rows = n; c.F = F; c.flags = 2 if scale else 0
            c.scale[0] = addr(ins["s0"]); c.scale[1] = addr(ins["s1"])
            c.scale_keys = keys(seed + i, 11); c.scale_trunc_keys = keys(seed + i, 12)
            if write_logits:
                c.out[0] = addr(ins["out0"]); c.out[1] = addr(ins["out1"])
            for q in (0, 1):
                j = J[2 * i + q]
                j.d_out = addr(ins["d0" if q == 0 else "d1"]); j.keys = keys(seed + i, 17); j.p = q; j.rows = n; j.train_rows = train; j.val_rows = val
                if q == 0:
                    j.labels = addr(ins["lab"]); j.border = addr(ins["bd"]); j.counts6 = addr(ins["cnt"]); j.loss = addr(ins["loss"])
                p.softmax[q] = ctypes.addressof(j)
        keep.append((s0, s1, labels, border, dS0, dS1, dL, dB))
        outs.append((host_bufs, dev_bufs))
    dt, drp, dcl = dev(table), dev(rowptr.view(np.int32)), dev(col.view(np.int32))
    ctx.call("cognn_gather_pair_chain_u64", ptr(dt), ptr(drp), ptr(dcl), F, gp, npairs)
    assert cpu.cognn_gather_pair_chain_u64(None, hp(table), hp(rowptr), hp(col), F, cp, npairs) == 0
    for hb, db in outs:
        assert np.array_equal(host(db["d0"]), hb["d0"]) and np.array_equal(host(db["d1"]), hb["d1"])
        if write_logits:
            assert np.array_equal(host(db["out0"]), hb["out0"]) and np.array_equal(host(db["out1"]), hb["out1"])
        assert np.array_equal(host(db["cnt"], np.int64)[:5], hb["cnt"][:5])
        got, want = float(host(db["loss"], np.float64)[0]), float(hb["loss"][0])
        assert abs(got - want) <= 1e-9 * max(1.0, abs(want))        # fp tolerance: order of the atomic additions
