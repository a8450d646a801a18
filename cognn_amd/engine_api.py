"""ctypes declarations for include/cognn_engine.h (the C++ GAS engine inside libcognn_hip.so)."""
import ctypes

_P = ctypes.c_void_p
_I = ctypes.c_int32
_L = ctypes.c_int64
_U = ctypes.c_uint64
_D = ctypes.c_double


class EngineConfig(ctypes.Structure):
    _fields_ = [("num_parties", _I), ("rank", _I), ("world", _I), ("variant", _I),
                ("num_layers", _I), ("num_labels", _I), ("input_dim", _I), ("hidden_dim", _I),
                ("learning_rate", _D), ("train_ratio", _D), ("val_ratio", _D), ("test_ratio", _D),
                ("seed", _U), ("device", _I), ("stream", _P), ("undirected", _I), ("verbose", _I), ("placement", _I)]


class Xfer(ctypes.Structure):
    _fields_ = [("peer", _I), ("is_send", _I), ("ptr", _P), ("bytes", _L)]


EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, _P, ctypes.POINTER(Xfer), _I)
EXCHANGE_WAIT_FN = ctypes.CFUNCTYPE(ctypes.c_int, _P)
EXCHANGE_WAIT_ROUND_FN = ctypes.CFUNCTYPE(ctypes.c_int, _P, _L)

_SIGNATURES = {
    "cognn_engine_last_error": (ctypes.c_char_p, []),
    "cognn_engine_create": (ctypes.c_int, [ctypes.POINTER(EngineConfig), _L, _L, _P, _P, _P, ctypes.POINTER(_P)]),
    "cognn_engine_destroy": (ctypes.c_int, [_P]),
    "cognn_engine_set_exchange": (ctypes.c_int, [_P, EXCHANGE_FN, _P]),
    "cognn_engine_set_exchange_async": (ctypes.c_int, [_P, EXCHANGE_FN, EXCHANGE_WAIT_FN, _P]),
    "cognn_engine_set_exchange_async2": (ctypes.c_int, [_P, EXCHANGE_FN, EXCHANGE_WAIT_FN, EXCHANGE_WAIT_ROUND_FN, _P]),
    "cognn_engine_party_rows": (ctypes.c_int, [_P, _I, ctypes.POINTER(_L)]),
    "cognn_engine_party_vids": (ctypes.c_int, [_P, _I, _P]),
    "cognn_engine_party_degrees": (ctypes.c_int, [_P, _I, _P, _P, _P]),
    "cognn_engine_set_party_data": (ctypes.c_int, [_P, _I, _P, _P]),
    "cognn_engine_set_weights": (ctypes.c_int, [_P, _P, _P]),
    "cognn_engine_start": (ctypes.c_int, [_P]),
    "cognn_engine_offline": (ctypes.c_int, [_P, _L, _L]),
    "cognn_engine_run": (ctypes.c_int, [_P, _L, _L]),
    "cognn_engine_offline_discard": (ctypes.c_int, [_P, _L, _L, ctypes.POINTER(_L)]),
    "cognn_engine_offline_save": (ctypes.c_int, [_P, ctypes.c_char_p]),
    "cognn_engine_offline_load": (ctypes.c_int, [_P, ctypes.c_char_p, _L, _L, ctypes.POINTER(_L)]),
    "cognn_engine_get_shares": (ctypes.c_int, [_P, _I, _I, _P, ctypes.POINTER(_L), ctypes.POINTER(_L)]),
    "cognn_engine_get_weight": (ctypes.c_int, [_P, _I, _I, _I, _P]),
    "cognn_engine_get_metrics": (ctypes.c_int, [_P, _I, _P]),
    "cognn_engine_enable_timing": (ctypes.c_int, [_P, _I]),
    "cognn_engine_get_timing": (ctypes.c_int, [_P, _I, ctypes.POINTER(_L), ctypes.POINTER(_D), ctypes.POINTER(_D)]),
    "cognn_engine_get_workload": (ctypes.c_int, [_P, _P]),
    "cognn_engine_sync": (ctypes.c_int, [_P]),
    "cognn_engine_get_phase_seconds": (ctypes.c_int, [_P, _P]),
    "cognn_engine_set_option": (ctypes.c_int, [_P, _I, _L]),
    "cognn_engine_get_memory": (ctypes.c_int, [_P, ctypes.POINTER(_L), ctypes.POINTER(_L)]),
}


# include/cognn_exchange.h: the native RCCL transport (only in the HIP library)
_EXCHANGE_SIGNATURES = {
    "cognn_exchange_last_error": (ctypes.c_char_p, []),
    "cognn_rccl_unique_id": (ctypes.c_int, [_P]),
    "cognn_rccl_rendezvous_tcp": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _D, _P]),
    "cognn_rccl_exchange_create": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, ctypes.POINTER(_P)]),
    "cognn_rccl_exchange_destroy": (ctypes.c_int, [_P]),
    "cognn_rccl_exchange_begin": (ctypes.c_int, [_P, ctypes.POINTER(Xfer), _I]),
    "cognn_rccl_exchange_wait": (ctypes.c_int, [_P]),
    "cognn_rccl_exchange_wait_round": (ctypes.c_int, [_P, _L]),
    "cognn_engine_set_exchange_rccl": (ctypes.c_int, [_P, _P]),
    "cognn_rccl_exchange_stats": (ctypes.c_int, [_P, ctypes.POINTER(_L), ctypes.POINTER(_L), ctypes.POINTER(_L)]),
    "cognn_rccl_exchange_time": (ctypes.c_int, [_P, ctypes.POINTER(_D)]),
    "cognn_rccl_exchange_barrier": (ctypes.c_int, [_P]),
    "cognn_rccl_exchange_ranks": (ctypes.c_int, [_P, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_L)]),
}
RCCL_ID_BYTES = 128


def exported_names():
    return list(_SIGNATURES.keys()) + list(_EXCHANGE_SIGNATURES.keys())


def declare(lib):
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _EXCHANGE_SIGNATURES.items():
        fn = getattr(lib, name, None)          # absent from a library built without the transport
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
