"""ctypes declarations for include/cognn_engine.h (the C++ GAS engine inside libcognn_hip.so)."""
import ctypes

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_U = ctypes.c_uint64

_SIGNATURES = {}


def exported_names():
    return list(_SIGNATURES.keys())


def declare(lib):
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
