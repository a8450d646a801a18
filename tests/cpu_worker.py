"""Test infrastructure: cognn_amd.worker with the engine's host code bound to the plain-C++ reference backend
(oracle/libcognn_engine_cpu.so) instead of the HIP library, so that the launcher / log contract / gloo transport can be
exercised on machines without a GPU.  Started by tools/run_cluster.py --worker tests/cpu_worker.py in tests/test_launcher_cpu.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cognn_amd import capi  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "oracle", "libcognn_engine_cpu.so")

from cognn_amd import worker  # noqa: E402

if __name__ == "__main__":
    worker.main()
