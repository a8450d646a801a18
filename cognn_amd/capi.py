"""ctypes binding of libcognn_hip.so (include/cognn_hip.h, include/cognn_engine.h).

Thin plumbing only: every compute call goes straight to the HIP library.  There is no CPU
fallback; a missing library or a missing GPU is an error the caller sees.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcognn_hip.so")

NUM_SLOTS = 11

# dealer op ids (cognn_amd/csrc/cognn_spec.h)
OP_SHARE_FEAT, OP_SHARE_W = 1, 2
(OP_PS_GEMM, OP_PS_GEMM_TRUNC, OP_PS_SCALE, OP_PS_SCALE_TRUNC, OP_GA_SCALE, OP_GA_SCALE_TRUNC,
 OP_AP_RELU, OP_AP_SOFTMAX, OP_AP_GEMM, OP_AP_GEMM_TRUNC, OP_AP_GSCALE_TRUNC, OP_AP_LR_TRUNC,
 OP_WAVG_TRUNC) = range(10, 23)
(SL_A0, SL_A1, SL_B0, SL_B1, SL_C0, SL_R, SL_R0, SL_RP0, SL_T, SL_T0, SL_RHO) = range(11)


class Keys(ctypes.Structure):
    _fields_ = [("k", ctypes.c_uint64 * NUM_SLOTS)]


class PairChain(ctypes.Structure):
    """cognn_pair_chain (include/cognn_hip.h)."""
    _fields_ = [("x", ctypes.c_void_p * 2), ("c1", ctypes.c_void_p), ("scale", ctypes.c_void_p * 2), ("out", ctypes.c_void_p * 2),
                ("open", ctypes.c_void_p * 2), ("mask", ctypes.c_void_p), ("open_key", ctypes.c_uint64 * 2),
                ("gemm_keys", Keys), ("trunc_in_keys", Keys), ("scale_keys", Keys), ("scale_trunc_keys", Keys), ("relu_keys", Keys),
                ("rows", ctypes.c_int64), ("F", ctypes.c_int64), ("flags", ctypes.c_int32), ("mask_in", ctypes.c_void_p), ("dealt", ctypes.c_void_p)]


PC_TRUNC_IN, PC_SCALE, PC_RELU, PC_INPUT_OPENED, PC_NO_C, PC_OPEN_SUM, WU_SWAP, PC_CLEAR_INPUT, WU_CLEAR_Z = 1, 2, 4, 8, 16, 32, 64, 128, 128


class PairWUpdate(ctypes.Structure):
    """cognn_pair_wupdate (include/cognn_hip.h)."""
    _fields_ = [("z", ctypes.c_void_p * 2), ("c1", ctypes.c_void_p), ("W", ctypes.c_void_p * 2), ("gemm_keys", Keys),
                ("trunc_keys", Keys * 4), ("mul", ctypes.c_uint64 * 3), ("n", ctypes.c_int64), ("flags", ctypes.c_int32)]


class ScatterPair(ctypes.Structure):
    """cognn_scatter_pair (include/cognn_hip.h)."""
    _fields_ = [("srcA", ctypes.c_void_p), ("srcB", ctypes.c_void_p), ("n0", ctypes.c_void_p), ("n1", ctypes.c_void_p),
                ("scale0", Keys), ("trunc0", Keys), ("scale1", Keys), ("trunc1", Keys), ("n1_from_server", ctypes.c_int32),
                ("crossed", ctypes.c_int32)]


class GatherPair(ctypes.Structure):
    """cognn_gather_pair (include/cognn_hip.h)."""
    _fields_ = [("a_row0", ctypes.c_int64), ("b_row0", ctypes.c_int64), ("chain", PairChain), ("softmax", ctypes.c_void_p * 2)]


class SoftmaxJob(ctypes.Structure):
    """cognn_softmax_job (include/cognn_hip.h)."""
    _fields_ = [("d_out", ctypes.c_void_p), ("z0", ctypes.c_void_p), ("z1", ctypes.c_void_p), ("labels", ctypes.c_void_p),
                ("border", ctypes.c_void_p), ("keys", Keys), ("p", ctypes.c_int32), ("rows", ctypes.c_int64),
                ("train_rows", ctypes.c_int64), ("val_rows", ctypes.c_int64), ("counts6", ctypes.c_void_p), ("loss", ctypes.c_void_p)]


class GemmJob(ctypes.Structure):
    """cognn_gemm_job (include/cognn_hip.h)."""
    _fields_ = [("Z", ctypes.c_void_p), ("E0", ctypes.c_void_p), ("E1", ctypes.c_void_p), ("F0", ctypes.c_void_p), ("F1", ctypes.c_void_p),
                ("c1", ctypes.c_void_p), ("keys", Keys), ("p", ctypes.c_int32), ("M", ctypes.c_int64), ("scratch", ctypes.c_void_p),
                ("E_presplit", ctypes.c_void_p), ("A_presplit", ctypes.c_void_p), ("A_dealt", ctypes.c_void_p), ("K", ctypes.c_int64), ("epilogue", ctypes.c_void_p), ("Z_zeroed", ctypes.c_int32)]


class DealerJob(ctypes.Structure):
    """cognn_dealer_job (include/cognn_hip.h)."""
    _fields_ = [("C1", ctypes.c_void_p), ("keys", Keys), ("M", ctypes.c_int64)]


class DealerTnJob(ctypes.Structure):
    """cognn_dealer_tn_job (include/cognn_hip.h)."""
    _fields_ = [("C1", ctypes.c_void_p), ("keys", Keys), ("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64), ("transA", ctypes.c_int32),
                ("scratchA", ctypes.c_void_p), ("scratchB", ctypes.c_void_p)]


class CognnError(RuntimeError):
    pass


_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_U = ctypes.c_uint64
_KP = ctypes.POINTER(Keys)

_SIGNATURES = {
    "cognn_abi_version": (_I, []),
    "cognn_last_error": (ctypes.c_char_p, []),
    "cognn_ctx_create": (_I, [_I, _P, ctypes.POINTER(_P)]),
    "cognn_ctx_create_private": (_I, [_I, ctypes.POINTER(_P)]),
    "cognn_ctx_destroy": (_I, [_P]),
    "cognn_ctx_sync": (_I, [_P]),
    "cognn_batch_begin": (_I, [_P]),
    "cognn_batch_end": (_I, [_P]),
    "cognn_ctx_set_chunk": (_I, [_P, ctypes.c_int32, ctypes.c_int32]),
    "cognn_chunk_range": (None, [_L, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_L), ctypes.POINTER(_L)]),
    "cognn_lane_begin": (_I, [_P, ctypes.c_int32]),
    "cognn_lane_select": (_I, [_P, ctypes.c_int32]),
    "cognn_lane_end": (_I, [_P]),
    "cognn_set_epoch_salt": (_I, [_P, _U]),
    "cognn_ctx_use_private_stream": (_I, [_P]),
    "cognn_graph_capture_begin": (_I, [_P]),
    "cognn_graph_capture_end": (_I, [_P, ctypes.POINTER(_P)]),
    "cognn_graph_launch": (_I, [_P, _P]),
    "cognn_graph_destroy": (_I, [_P, _P]),
    "cognn_malloc": (_I, [_P, ctypes.POINTER(_P), ctypes.c_size_t]),
    "cognn_free": (_I, [_P, _P]),
    "cognn_memcpy_h2d": (_I, [_P, _P, _P, ctypes.c_size_t]),
    "cognn_memcpy_d2h": (_I, [_P, _P, _P, ctypes.c_size_t]),
    "cognn_memcpy_d2d": (_I, [_P, _P, _P, ctypes.c_size_t]),
    "cognn_memset0": (_I, [_P, _P, ctypes.c_size_t]),
    "cognn_make_keys": (None, [_U, _U, _U, _U, _KP]),
    "cognn_fx_encode_f64": (_I, [_P, _P, _P, _P, _L, _L]),
    "cognn_share_split_u64": (_I, [_P, _P, _U, _P, _P, _L]),
    "cognn_prng_fill_u64": (_I, [_P, _P, _U, _L]),
    "cognn_gemm_mask_fill_u64": (_I, [_P, _P, _U, _L]),
    "cognn_pack48_u64": (_I, [_P, _P, _P, _L]),
    "cognn_unpack48_u64": (_I, [_P, _P, _P, _L]),
    "cognn_gather_csr_u64": (_I, [_P, _P, _P, _P, _P, _P, _L, _L]),
    "cognn_gather_csr_open_u64": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, ctypes.c_int32, _P, _P, _P]),
    "cognn_relu_close_open_u64": (_I, [_P, _P, _P, _P, _P, _P, _P, _U, _L]),
    "cognn_scatter_add_rows_u64": (_I, [_P, _P, _P, _P, _L, _L]),
    "cognn_ring_gemm_u64": (_I, [_P, _P, _P, _P, _L, _L, _L, _I, _I]),
    "cognn_ring_gemm2_u64": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _I, _I]),
    "cognn_mask_open_u64": (_I, [_P, _P, _P, _U, _L, _L, _I]),
    "cognn_add_u64": (_I, [_P, _P, _P, _P, _L]),
    "cognn_sub_u64": (_I, [_P, _P, _P, _P, _L]),
    "cognn_sum_u64": (_I, [_P, _P, _P, ctypes.c_int32, _L]),
    "cognn_fanout_u64": (_I, [_P, _P, ctypes.c_int32, _P, _L]),
    "cognn_dealer_gemm_c1_u64": (_I, [_P, _P, _KP, _L, _L, _L, _I, _P, _P]),
    "cognn_dealer_gemm_c1_groupable": (_I, [_L, _L]),
    "cognn_dealer_gemm_c1_group_u64": (_I, [_P, _P, ctypes.c_int32, _L, _L]),
    "cognn_dealer_gemm_c1_tn_group_u64": (_I, [_P, _P, ctypes.c_int32]),
    "cognn_beaver_gemm_close_u64": (_I, [_P, _P, _P, _P, _P, _P, _KP, _I, _L, _L, _L, _I, _P]),
    "cognn_trunc_open_u64": (_I, [_P, _P, _P, _U, _KP, _I, _L]),
    "cognn_trunc_open_add_u64": (_I, [_P, _P, _P, _P, _KP, _KP, _I, _L]),
    "cognn_beaver_gemm_fusable": (_I, [_L, _L, _L, _I]),
    "cognn_beaver_gemm_close_raw_u64": (_I, [_P, _P, _P, _P, _P, _KP, _I, _L, _L, _L, _P]),
    "cognn_beaver_gemm_close2_u64": (_I, [_P, _P, _P, _P, _P, _P, _P, _KP, _I, _L, _L, _L, _I, _P, _I]),
    "cognn_beaver_gemm_close_group_u64": (_I, [_P, ctypes.POINTER(GemmJob), ctypes.c_int32, _L, _L, _I]),
    "cognn_beaver_gemm_tn_groupable": (_I, [_L, _L, _L, _I]),
    "cognn_beaver_gemm_close_group_tn_u64": (_I, [_P, ctypes.POINTER(GemmJob), ctypes.c_int32, _L, _L, _I]),
    "cognn_beaver_gemm_group_takes_epilogue": (_I, [_L, _L, _L]),
    "cognn_beaver_gemm_group_is_whole_k": (_I, [_L, _L, _L]),
    "cognn_gemm_presplit_bytes": (_L, [_L, _L]),
    "cognn_gemm_presplit_u64": (_I, [_P, _P, _P, _P, _L, _L]),
    "cognn_gemm_presplit_tn_bytes": (_L, [_L, _L]),
    "cognn_gemm_presplit_tn_u64": (_I, [_P, _P, _P, _P, _U, _I, _L, _L]),
    "cognn_trunc_close_u64": (_I, [_P, _P, _P, _P, _KP, _I, _I, _L]),
    "cognn_trunc_close_open_u64": (_I, [_P, _P, _P, _P, _P, _KP, _I, _U, _L]),
    "cognn_trunc_close_pub_u64": (_I, [_P, _P, _P, _P, _P, _KP, _I, _U, _U, _I, _L]),
    "cognn_trunc_close_pub_dealt_u64": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L]),
    "cognn_dealer_trunc_pub_u64": (_I, [_P, _P, _P, _P, _KP, _U, _U, _I, _L]),
    "cognn_rowscale_open_u64": (_I, [_P, _P, _P, _P, _P, _KP, _I, _L, _L]),
    "cognn_rowscale_close_u64": (_I, [_P, _P, _P, _P, _P, _P, _KP, _KP, _I, _L, _L]),
    "cognn_relu_open_u64": (_I, [_P, _P, _P, _P, _KP, _I, _L]),
    "cognn_relu_mul_u64": (_I, [_P, _P, _P, _P, _P, _P, _KP, _I, _L]),
    "cognn_relu_close_u64": (_I, [_P, _P, _P, _P, _P, _P, _L]),
    "cognn_mask_select_u64": (_I, [_P, _P, _P, _P, _L]),
    "cognn_softmax_u64": (_I, [_P, _P, _P, _P, _P, _P, _P, _KP, _I, _L, _L, _L]),
    "cognn_metrics_q16": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _P, _P]),
    "cognn_pair_chain_u64": (_I, [_P, ctypes.POINTER(PairChain), ctypes.c_int32]),
    "cognn_pair_weight_update_u64": (_I, [_P, ctypes.POINTER(PairWUpdate), ctypes.c_int32, ctypes.POINTER(Keys), ctypes.c_uint64, ctypes.c_int32]),
    "cognn_pair_chain_dealt_slots": (_L, [ctypes.c_int32, ctypes.c_int32]),
    "cognn_pair_chain_deal_u64": (_I, [_P, ctypes.POINTER(PairChain), _P]),
    "cognn_softmax_jobs_u64": (_I, [_P, ctypes.POINTER(SoftmaxJob), ctypes.c_int32, _L]),
    "cognn_scatter_gather_original_u64": (_I, [_P, _P, _P, _P, _P, _P, _P, _KP, _KP, _L, _L, _P, _P, _P, _P, ctypes.POINTER(ScatterPair), ctypes.c_int32]),
    "cognn_gather_pair_chain_u64": (_I, [_P, _P, _P, _P, _L, ctypes.POINTER(GatherPair), ctypes.c_int32]),
    "cognn_gather_pair_chain_takes_softmax": (_I, [_L]),
    "cognn_gather_pair_chain_base_u64": (_I, [_P, _P, _P, _P, _P, _L, ctypes.POINTER(GatherPair), ctypes.c_int32]),
    "cognn_graph_build_colocated": (_I, [_P, _L, _L, ctypes.c_int32] + [_P] * 6 + [_L] + [_P] * 8),
    "cognn_transpose_u64": (_I, [_P, _P, _P, _L, _L]),
    "cognn_timer_begin": (_I, [_P, _I]),
    "cognn_timer_end": (_I, [_P, _I]),
    "cognn_timer_read": (_I, [_P, _I, ctypes.POINTER(_L), ctypes.POINTER(ctypes.c_double)]),
    "cognn_timer_reset": (_I, [_P]),
}

_lib = None


def exported_names():
    return list(_SIGNATURES.keys())


def load():
    """Load libcognn_hip.so (LIB_PATH) once and declare every entry point of the C ABI."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    try:
        # Load order matters: torch ships its own libamdhip64.so.7; importing it first makes this library bind to
        # the same HIP runtime instance instead of bringing a second one into the process.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise CognnError("%s not found: build it with `make` (or __graft_entry__.build()); "
                         "the engine has no CPU fallback" % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    try:
        from . import engine_api
        engine_api.declare(lib)
    except ImportError:
        pass
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise CognnError(load().cognn_last_error().decode())


def make_keys(seed, owner, it, op):
    k = Keys()
    load().cognn_make_keys(seed, owner, it, op, ctypes.byref(k))
    return k


class Context:
    """cognn_ctx bound to a HIP stream (by default torch's current stream on `device`)."""

    def __init__(self, device=0, stream=None):
        lib = load()
        h = _P()
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(device).cuda_stream
        check(lib.cognn_ctx_create(device, _P(stream), ctypes.byref(h)))
        self.lib = lib
        self.h = h
        self.device = device

    def sync(self):
        check(self.lib.cognn_ctx_sync(self.h))

    def close(self):
        if self.h:
            self.lib.cognn_ctx_destroy(self.h)
            self.h = None

    def call(self, name, *args):
        check(getattr(self.lib, name)(self.h, *args))
