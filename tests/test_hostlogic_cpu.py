"""Engine host logic on the CPU reference backend: device memory bounded over epochs (a consumed product share is recycled),
the offline cache only accepts files written for exactly this product of exactly this run, replays with and without
retention agree.  The same worker runs on the HIP library in tests/test_hostlogic_gpu.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_worker(backend, tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostlogic_worker.py"), backend, str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def check(out):
    mem = out["mem"]
    assert mem[1] == mem[-1], "device allocations grow with the number of epochs: %s" % mem       # epoch 0 may add the first pool entries
    assert mem[0][0] <= mem[1][0] <= mem[0][0] + 8
    assert out["files"] == 3 * 5                       # 3 co-party sides x (2 forward + 3 backward) products of one epoch
    assert out["loaded_same"] == 15
    for key in ("loaded_other_hidden", "loaded_other_graph", "loaded_other_parties", "loaded_other_seed"):
        assert out[key] == 0, key
    assert out["loaded_truncated"] == 14
    assert out["replay"]["True"] == {"same": True, "grew": 0}
    assert out["replay"]["False"]["same"] is True
    assert out["discard"] == {"first": 15, "again": 0, "grew": 0, "same": True}
    assert out["epoch_calls_identical"] is True
    assert out["epoch_calls_memory"][1] == out["epoch_calls_memory"][2]
    ph = out["phases"]
    assert ph["rounds"] == 0 and ph["apply"] > 0       # single rank: no exchange rounds; iteration 35 is a backward one


@pytest.fixture(scope="module", autouse=True)
def _build_cpu_engine():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def test_host_logic_cpu_backend(tmp_path):
    check(run_worker("cpu", tmp_path))
