"""The degenerate inputs of tests/test_edge_cases_cpu.py on the HIP kernels (zero-row launches, empty CSR segments, ...)."""
import pytest

from test_edge_cases_cpu import CASES, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_edge_case_hip(name):
    run_case(name, "hip")


@pytest.mark.parametrize("name", ["duplicate-edges", "skewed-partition", "star"])
def test_edge_case_hip_host_graph_builder(name):
    """world-1 runs normally build degrees and CSR on the device; the host builder (what multi-rank runs use) gives the same shares"""
    run_case(name, "hip", host_graph_build=True)
