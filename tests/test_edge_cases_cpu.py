"""Degenerate inputs through the engine's host logic (CPU reference backend, one subprocess per case): no edges, an empty
party, one vertex per party, a star, duplicate edges, everything in one party, a skewed partition."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["no-edges", "empty-party", "one-vertex-each", "star", "duplicate-edges", "all-in-one-party", "skewed-partition"]


@pytest.fixture(scope="module", autouse=True)
def _build_cpu_engine():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def run_case(name, backend, host_graph_build=False):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    if host_graph_build:                                  # the engine's host CSR builder instead of the device one (world 1)
        env["COGNN_HOST_GRAPH_BUILD"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "edge_worker.py"), name, backend], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("name", CASES)
def test_edge_case_cpu_backend(name):
    run_case(name, "cpu")


@pytest.mark.parametrize("name", ["duplicate-edges", "skewed-partition", "empty-party"])
def test_edge_case_cpu_backend_host_graph_builder(name):
    run_case(name, "cpu", host_graph_build=True)
