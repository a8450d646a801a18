"""Helpers for the -m gpu parity tests: numpy <-> device tensors (torch is only the allocator)."""
import ctypes
import numpy as np

U64 = np.uint64
_KEEP = []      # device tensors stay alive until the next test starts (pointers are passed raw)


def release():
    del _KEEP[:]


def dev(arr):
    import torch
    a = np.ascontiguousarray(arr)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    t = torch.from_numpy(a).cuda()
    _KEEP.append(t)
    return t


def dev_empty(shape, dtype="u64"):
    import torch
    td = {"u64": torch.int64, "u8": torch.uint8, "i32": torch.int32, "f64": torch.float64, "u32": torch.int32}[dtype]
    t = torch.zeros(shape, dtype=td, device="cuda")
    _KEEP.append(t)
    return t


def host(t, dtype=U64):
    a = t.detach().cpu().numpy()
    if dtype == U64:
        return a.view(np.uint64)
    if dtype == np.uint32:
        return a.view(np.uint32)
    return a


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def rand_u64(rng, shape):
    return rng.integers(0, 1 << 64, size=shape, dtype=np.uint64)
