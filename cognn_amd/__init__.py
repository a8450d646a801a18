"""cognn_amd — MI355X-native engine for CoGNN's secret-shared GCN hot path.

Python here is plumbing (ctypes binding of the C ABI, torch for device memory / streams /
torch.distributed); the product is libcognn_hip.so (HIP kernels + C++ host engine).
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
