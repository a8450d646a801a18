"""bench.py's N > 1 branch end to end on the one-GPU box: two ranks share cuda:0, torch.distributed runs over gloo and the
share exchange is staged through host memory (COGNN_BENCH_BACKEND=gloo); the driver's multi-GPU runs use RCCL instead.
Checks the contract of the printed line (one JSON object from rank 0, whole-job value, n_gpus, scaling)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("chunks,placement", [(1, "vertex-set"), (3, "party"), (1, "party")])
def test_bench_two_ranks_over_gloo(chunks, placement):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, COGNN_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "small", "--chunks", str(chunks), "--placement", placement], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["config"]["parties"] == 8 and d["config"]["exchange"] == "gloo-host-staged"
    assert d["config"]["exchange_chunks"] == chunks and d["config"]["placement"] == placement
    assert d["cpu_baseline"] is None or isinstance(d["cpu_baseline"], dict)
    # the N-rank run ends in the single-process bits (rank 0 re-runs the job as one process and compares every rank's share digests)
    assert d["check"]["cross_path_identical"] is True and d["check"]["shares_compared"] == 16, d["check"]
    if placement == "party":                                 # the other placement is measured (and checked) beside the headline one
        v = d["vertex_set_placement"]
        assert v["placement"] == "vertex-set" and v["ms_per_step"] > 0 and v["check"]["cross_path_identical"] is True, v
    else:
        assert "vertex_set_placement" not in d


def test_bench_started_bare_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run (no WORLD_SIZE in the environment) - how the driver starts the N = 1
    bench: the process becomes the launcher before it touches a GPU, starts the two ranks as child processes and relays rank 0's
    single JSON line; the record shows how many ranks the transport's own all-reduce saw."""
    env = {key: val for key, val in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["COGNN_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "small",
                        "--no-placement-leg"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["check"]["cross_path_identical"] is True
    assert d["exchange"]["ranks"] == 2 and d["exchange"]["allreduce_of_ones"] == 2
    assert d["switches"].get("COGNN_BENCH_BACKEND") == "gloo"


def test_bench_started_bare_reports_a_failing_rank():
    """... and a rank that dies takes the launcher's exit code with it (no JSON line, the other rank is stopped)."""
    env = {key: val for key, val in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["COGNN_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0", "--workload", "small"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)       # 8 parties do not divide over 3 ranks
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
