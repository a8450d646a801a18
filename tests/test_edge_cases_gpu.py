"""The degenerate inputs of tests/test_edge_cases_cpu.py on the HIP kernels (zero-row launches, empty CSR segments, ...)."""
import pytest

from test_edge_cases_cpu import CASES, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_edge_case_hip(name):
    run_case(name, "hip")
