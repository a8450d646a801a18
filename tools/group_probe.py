#!/usr/bin/env python3
"""Timing probe of cognn_beaver_gemm_close_group_u64 at dataset shapes (raw products of one co-located pair, fragment-ordered E):
  python tools/group_probe.py M K N [pairs]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cognn_amd import capi  # noqa: E402


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def main():
    M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    pairs = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    ctx = capi.Context(0)
    lib = capi.load()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    rnd = lambda *s: torch.randint(-2**62, 2**62, s, dtype=torch.int64, device="cuda", generator=g)
    jobs = (capi.GemmJob * (2 * pairs))()
    keep = []
    for q in range(pairs):
        E, F = rnd(M, K), rnd(K, N)
        img = torch.empty(lib.cognn_gemm_presplit_bytes(M, K) // 8, dtype=torch.int64, device="cuda")
        ctx.call("cognn_gemm_presplit_u64", P(img), P(E), None, M, K)
        k = capi.make_keys(1, q, 3, capi.OP_PS_GEMM)
        for p in (0, 1):
            J = jobs[2 * q + p]
            Z = torch.zeros((M, N), dtype=torch.int64, device="cuda"); scr = torch.empty(M * K + K * N, dtype=torch.int64, device="cuda")
            J.Z = Z.data_ptr(); J.E0 = E.data_ptr(); J.F0 = F.data_ptr(); J.keys = k; J.p = p; J.M = M; J.scratch = scr.data_ptr()
            J.E_presplit = img.data_ptr()
            keep += [Z, scr]
        keep += [E, F, img]
    run = lambda: ctx.call("cognn_beaver_gemm_close_group_u64", jobs, len(jobs), N, K, 1)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        run()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    ops = 2.0 * 36 * 2 * M * K * N * 2 * pairs
    print("M=%d K=%d N=%d pairs=%d whole_k=%d: %.1f us per call (incl. zeroing launch if split K)  %.0f i8-TOP/s  E image %.0f MB"
          % (M, K, N, pairs, lib.cognn_beaver_gemm_group_is_whole_k(N, K, 2 * pairs * ((M + 15) // 16)), us, ops / us / 1e6, pairs * M * K * 8 / 1e6))


if __name__ == "__main__":
    main()
