"""tests/test_hostlogic_cpu.py's checks on the HIP library (bounded device memory, offline cache validation, replay)."""
import pytest

from test_hostlogic_cpu import check, run_worker

pytestmark = pytest.mark.gpu


def test_host_logic_hip_backend(tmp_path):
    check(run_worker("hip", tmp_path))
