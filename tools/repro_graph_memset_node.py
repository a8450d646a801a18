"""Reproducer (ROCm 7.2, gfx950): a recorded launch sequence  fill kernel -> d2d copy -> zeroing -> add -> add  on a private non-blocking
stream, replayed six times, with one operation of ANOTHER context on the legacy default stream + a wait for it between replays.
With the zeroing recorded as a hipMemsetAsync node (COGNN_ZERO_WITH_MEMSET=1) the replays after the first foreign operation are
wrong - half of the elements after a kernel of this library, all of them after a torch kernel: the zeroing is no longer ordered before
the node that follows it; with the copy node alone, or with the zeroing done by a kernel (the library's default, cg_zero), every
replay is exact.  This is why no entry point of the library issues hipMemsetAsync (csrc/common.h) - found through
tests/test_engine_gpu.py::test_recorded_and_eager_engines_interleaved.
    COGNN_ZERO_WITH_MEMSET=1 python tools/repro_graph_memset_node.py      # shows the wrong replays
    python tools/repro_graph_memset_node.py                               # all zeros"""
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import cognn_oracle as co
from cognn_amd import capi

def P(t, off=0): return ctypes.c_void_p(t.data_ptr() + 8 * off)

def run(name, n, foreign, use_copy=True, use_memset=True, reps=6):
    a = capi.Context(0)
    a.call("cognn_ctx_use_private_stream")
    other = capi.Context(0)
    x = torch.zeros(n, dtype=torch.int64, device="cuda"); y = torch.zeros(n, dtype=torch.int64, device="cuda")
    z = torch.zeros(n, dtype=torch.int64, device="cuda"); acc = torch.zeros(n, dtype=torch.int64, device="cuda")
    fb = torch.zeros(1 << 16, dtype=torch.int64, device="cuda"); fb2 = torch.ones(1 << 16, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    key = 0x1234567
    zero = torch.zeros(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    def body():
        a.call("cognn_prng_fill_u64", P(x), key, n)             # x = prng(salted)
        if use_copy: a.call("cognn_memcpy_d2d", P(y), P(x), n * 8)   # y = x
        else: a.call("cognn_add_u64", P(y), P(x), P(zero), n)    # y = x + 0
        if use_memset: a.call("cognn_memset0", P(z), n * 8)      # z = 0
        else: a.call("cognn_sub_u64", P(z), P(z), P(z), n)       # z = z - z
        a.call("cognn_add_u64", P(z), P(z), P(y), n)             # z += y  (= x)
        a.call("cognn_add_u64", P(acc), P(acc), P(z), n)         # acc += z
    body(); a.call("cognn_ctx_sync")
    a.call("cognn_graph_capture_begin"); body()
    ex = ctypes.c_void_p(); a.call("cognn_graph_capture_end", ctypes.byref(ex))
    want = co.prng(key, n).copy()                                # acc after the warm pass (salt 0)
    res = []
    for e in range(1, reps + 1):
        salt = e * 0x9E3779B97F4A7C15 % 2**64
        a.call("cognn_set_epoch_salt", salt)
        a.call("cognn_graph_launch", ex)
        a.call("cognn_set_epoch_salt", 0)
        with np.errstate(over="ignore"):
            want = want + co.prng((key + salt) % 2**64, n)
        got = acc.cpu().numpy().view(np.uint64)
        res.append(int((got != want).sum()))
        want = got.copy()                                        # (errors do not carry over to the next comparison)
        if foreign == "add": other.call("cognn_add_u64", P(fb), P(fb), P(fb2), 1 << 16); other.call("cognn_ctx_sync")
        elif foreign == "memset": other.call("cognn_memset0", P(fb), 8 << 16); other.call("cognn_ctx_sync")
        elif foreign == "torch": fb.add_(1); other.call("cognn_ctx_sync")
    print(name, "wrong elements per replay:", res, flush=True)
    a.call("cognn_graph_destroy", ex); a.close(); other.close()

for n in (4096, 1 << 20):
    for uc, um in ((True, True), (True, False), (False, True), (False, False)):
        for fo in ("add", "torch"):
            run("n=%d copy node %s, memset node %s, foreign %s" % (n, uc, um, fo), n, fo, use_copy=uc, use_memset=um)
