#!/usr/bin/env python3
"""Kernel micro-benchmarks on cuda:0 (HIP events via torch): the fused Beaver GEMM close and the CSR gather at
config5 shapes.  Usage: python tools/microbench.py [gemm|gather|all] [--iters N]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cognn_amd import capi  # noqa: E402


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def bench_gemm(ctx, iters, M=1 << 17, K=128, N=64):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    E0 = torch.randint(-2**62, 2**62, (M, K), dtype=torch.int64, device="cuda", generator=g)
    E1 = torch.randint(-2**62, 2**62, (M, K), dtype=torch.int64, device="cuda", generator=g)
    F = torch.randint(-2**62, 2**62, (K, N), dtype=torch.int64, device="cuda", generator=g)
    c1 = torch.randint(-2**62, 2**62, (M, N), dtype=torch.int64, device="cuda", generator=g)
    Z = torch.empty((M, N), dtype=torch.int64, device="cuda")
    scratch = torch.empty(M * K + K * N, dtype=torch.int64, device="cuda")
    k = capi.make_keys(1, 2, 3, capi.OP_PS_GEMM)
    for p in (1, 0):
        ms = timeit(lambda: ctx.call("cognn_beaver_gemm_close_u64", P(Z), P(E0), P(E1), P(F), P(c1) if p == 1 else None,
                                     ctypes.byref(k), p, M, N, K, 0, P(scratch)), iters)
        ops = 2.0 * 36 * 2 * M * K * N
        print("beaver_gemm_close p=%d M=%d K=%d N=%d: %.3f ms  %.1f i8-TOP/s (%.1f%% of 5000)  E-stream %.0f GB/s"
              % (p, M, K, N, ms, ops / ms / 1e9, ops / ms / 1e9 / 50.0, 2 * M * K * 8 / ms / 1e6))
    for p in (0,):
        ms = timeit(lambda: ctx.call("cognn_beaver_gemm_close_raw_u64", P(Z), P(E0), P(E1), P(F), ctypes.byref(k), p, M, N, K, P(scratch)), iters)
        ops = 2.0 * 36 * 2 * M * K * N
        print("beaver_gemm_close_raw p=%d M=%d K=%d N=%d: %.3f ms  %.1f i8-TOP/s (%.1f%% of 5000)" % (p, M, K, N, ms, ops / ms / 1e9, ops / ms / 1e9 / 50.0))
        # the layer-0 product of the engine: the opened feature tensor is kept pre-summed (one stream)
        ms = timeit(lambda: ctx.call("cognn_beaver_gemm_close_raw_u64", P(Z), P(E0), None, P(F), ctypes.byref(k), p, M, N, K, P(scratch)), iters)
        print("beaver_gemm_close_raw one E stream p=%d M=%d K=%d N=%d: %.3f ms  %.1f i8-TOP/s (%.1f%% of 5000)  %.0f GB/s"
              % (p, M, K, N, ms, ops / ms / 1e9, ops / ms / 1e9 / 50.0, (M * K + M * N) * 8 / ms / 1e6))
    A = E0; B = F; C = Z
    ms = timeit(lambda: ctx.call("cognn_ring_gemm_u64", P(C), P(A), P(B), M, N, K, 0, 0), iters)
    print("ring_gemm (single product) M=%d K=%d N=%d: %.3f ms  %.1f i8-TOP/s" % (M, K, N, ms, 36 * 2.0 * M * K * N / ms / 1e9))


def bench_gemm_tn(ctx, iters, M=128, N=64, K=1 << 17):
    """weight-gradient product d = h_t^T . in (gcn.h:671,710): logical A [M x K] stored [K x M], K = #vertices"""
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    E0 = torch.randint(-2**62, 2**62, (K, M), dtype=torch.int64, device="cuda", generator=g)
    E1 = torch.randint(-2**62, 2**62, (K, M), dtype=torch.int64, device="cuda", generator=g)
    F = torch.randint(-2**62, 2**62, (K, N), dtype=torch.int64, device="cuda", generator=g)
    c1 = torch.randint(-2**62, 2**62, (M, N), dtype=torch.int64, device="cuda", generator=g)
    Z = torch.empty((M, N), dtype=torch.int64, device="cuda")
    scratch = torch.empty(M * K + K * N, dtype=torch.int64, device="cuda")
    k = capi.make_keys(1, 2, 3, capi.OP_AP_GEMM)
    for p in (1, 0):
        ms = timeit(lambda: ctx.call("cognn_beaver_gemm_close_u64", P(Z), P(E0), P(E1), P(F), P(c1) if p == 1 else None,
                                     ctypes.byref(k), p, M, N, K, 1, P(scratch)), iters)
        ops = 2.0 * 36 * 2 * M * K * N
        gb = 8.0 * (2 * M * K + K * N) / 1e9
        print("beaver_gemm_close TN p=%d M=%d K=%d N=%d: %.3f ms  %.1f i8-TOP/s  operand streams %.0f GB/s"
              % (p, M, K, N, ms, ops / ms / 1e9, gb / (ms / 1e3)))
    ms = timeit(lambda: ctx.call("cognn_beaver_gemm_close_u64", P(Z), P(E0), None, P(F), None, ctypes.byref(k), 0, M, N, K, 1, P(scratch)), iters)
    print("beaver_gemm_close TN one E stream p=0 M=%d K=%d N=%d: %.3f ms  %.1f i8-TOP/s" % (M, K, N, ms, 2.0 * 36 * 2 * M * K * N / ms / 1e9))


def bench_gather(ctx, iters, rows=1 << 21, table_rows=1 << 21, deg=12, F=64):
    rng = np.random.default_rng(0)
    d = rng.poisson(deg, size=rows)
    rowptr = np.zeros(rows + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(d)
    col = rng.integers(0, table_rows, size=int(rowptr[-1]), dtype=np.uint32)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    table = torch.randint(-2**62, 2**62, (table_rows, F), dtype=torch.int64, device="cuda", generator=g)
    out = torch.empty((rows, F), dtype=torch.int64, device="cuda")
    base = table if table_rows >= rows else torch.zeros((rows, F), dtype=torch.int64, device="cuda")   # (base has `rows` rows)
    rp = torch.from_numpy(rowptr.view(np.int32)).cuda(); cl = torch.from_numpy(col.view(np.int32)).cuda()
    assert int(col.max()) < table_rows and len(rowptr) == rows + 1 and base.shape[0] >= rows
    ms = timeit(lambda: ctx.call("cognn_gather_csr_u64", P(out), P(base), P(table), P(rp), P(cl), rows, F), iters)
    E = int(rowptr[-1])
    by = 8.0 * F * (E + 2 * rows) + 4.0 * E + 4.0 * (rows + 1)
    print("gather_csr rows=%d edges=%d F=%d: %.3f ms  %.0f GB/s algorithmic (%.1f%% of 8000)" % (rows, E, F, ms, by / ms / 1e6, by / ms / 1e6 / 80.0))


def bench_overlap(ctx, iters, rows=1 << 21, deg=16, F=64, M=1 << 17, K=64, N=16, n_gemm=16):
    """Does the HBM-bound gather hide the VALU-bound N <= 16 products of other owners?  Gather on lane 0, n_gemm Beaver closes
    on lane 1 (cognn_lane_*), against each alone."""
    rng = np.random.default_rng(0)
    d = rng.poisson(deg, size=rows)
    rowptr = np.zeros(rows + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(d)
    col = rng.integers(0, rows, size=int(rowptr[-1]), dtype=np.uint32)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    table = torch.randint(-2**62, 2**62, (rows, F), dtype=torch.int64, device="cuda", generator=g)
    out = torch.empty((rows, F), dtype=torch.int64, device="cuda")
    rp = torch.from_numpy(rowptr.view(np.int32)).cuda(); cl = torch.from_numpy(col.view(np.int32)).cuda()
    E0 = torch.randint(-2**62, 2**62, (M, K), dtype=torch.int64, device="cuda", generator=g)
    E1 = torch.randint(-2**62, 2**62, (M, K), dtype=torch.int64, device="cuda", generator=g)
    Fm = torch.randint(-2**62, 2**62, (K, N), dtype=torch.int64, device="cuda", generator=g)
    Z = [torch.empty((M, N), dtype=torch.int64, device="cuda") for _ in range(n_gemm)]
    scratch = [torch.empty(M * K + K * N, dtype=torch.int64, device="cuda") for _ in range(n_gemm)]
    k = capi.make_keys(1, 2, 3, capi.OP_PS_GEMM)

    def gather():
        ctx.call("cognn_gather_csr_u64", P(out), P(table), P(table), P(rp), P(cl), rows, F)

    def gemms():
        for i in range(n_gemm):
            ctx.call("cognn_beaver_gemm_close_raw_u64", P(Z[i]), P(E0), P(E1), P(Fm), ctypes.byref(k), i & 1, M, N, K, P(scratch[i]))

    def both():
        ctx.call("cognn_lane_begin", 2)
        ctx.call("cognn_lane_select", 0); gather()
        ctx.call("cognn_lane_select", 1); gemms()
        ctx.call("cognn_lane_end")

    tg, tm, tb = timeit(gather, iters), timeit(gemms, iters), timeit(both, iters)
    print("overlap: gather F=%d alone %.3f ms, %d products (M=%d K=%d N=%d) alone %.3f ms, together on two lanes %.3f ms (sum %.3f)"
          % (F, tg, n_gemm, M, K, N, tm, tb, tg + tm))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    ctx = capi.Context(0)
    if a.what in ("gemm", "all"):
        bench_gemm(ctx, a.iters)
        bench_gemm(ctx, a.iters, K=64, N=16)
        bench_gemm(ctx, a.iters, K=128, N=16)
        bench_gemm(ctx, a.iters, K=16, N=16)
        bench_gemm(ctx, a.iters, K=64, N=32)
    if a.what in ("gemm_tn", "all"):
        bench_gemm_tn(ctx, a.iters)
        bench_gemm_tn(ctx, a.iters, M=64, N=16)
    if a.what in ("overlap",):
        bench_overlap(ctx, a.iters)
        bench_overlap(ctx, a.iters, K=128, N=64, n_gemm=8)
    if a.what in ("gather_sizes",):                          # how fast do indexed row fetches go when the table fits the Infinity Cache / L2?
        for lt in (21, 20, 19, 18, 17, 16, 14):
            print("table %d MiB:" % ((1 << lt) * 64 * 8 >> 20), end=" ")
            bench_gather(ctx, a.iters, table_rows=1 << lt, deg=16, F=64)
    if a.what in ("gather", "all"):
        bench_gather(ctx, a.iters, F=64)
        bench_gather(ctx, a.iters, F=16)
