"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/*.h declares.  No compute call is made here (there is no GPU and no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for hdr in os.listdir(os.path.join(ROOT, "include")):
        if not hdr.endswith(".h"):
            continue
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(cognn_[a-z0-9_]+)\s*\(", text))
    return names


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from cognn_amd import capi
    return capi.load()


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in sorted(_declared_symbols()) if not hasattr(lib, n)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing


def test_binding_covers_every_declared_symbol(lib):
    from cognn_amd import capi, engine_api
    bound = set(capi.exported_names()) | set(engine_api.exported_names())
    assert _declared_symbols() <= bound


def test_abi_version_and_loud_failure_without_gpu(lib):
    assert lib.cognn_abi_version() == 3
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = lib.cognn_ctx_create(0, None, ctypes.byref(h))
    assert rc != 0
    assert b"no HIP device" in lib.cognn_last_error()


def test_key_derivation_matches_oracle(lib):
    import cognn_oracle as co
    from cognn_amd import capi
    k = capi.make_keys(0xC06A11, 3, 7, co.OP_GA_SCALE)
    for s in range(capi.NUM_SLOTS):
        assert k.k[s] == co.stream_key(0xC06A11, 3, 7, co.OP_GA_SCALE, s)


def test_chunk_range_partitions_every_length(lib):
    """cognn_chunk_range (host function, no GPU): the C chunks of any length are contiguous, even-aligned, ordered and cover
    [0, n) exactly - the partition the element-wise kernels' chunk window and the engine's chunked messages share."""
    import ctypes
    lo = ctypes.c_int64(); hi = ctypes.c_int64()
    for n in list(range(0, 40)) + [1433 * 16, (1 << 20) * 64 + 3, 7 * 19717]:
        for C in (1, 2, 3, 4, 5, 8, 64):
            end = 0
            for c in range(C):
                lib.cognn_chunk_range(n, c, C, ctypes.byref(lo), ctypes.byref(hi))
                assert lo.value == end and hi.value >= lo.value and (lo.value % 2 == 0 or C == 1), (n, C, c, lo.value, hi.value)
                end = hi.value
            assert end == n, (n, C)
