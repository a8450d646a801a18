"""The one-GPU bench line end to end at the `small` workload (bench.py's config5 at 1/64 scale): the contract keys, the per-kernel
roofline rows, the three dealer forms with identical digests, the steady-state offline phase, the switches in effect."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload", ["small", "cora-2p"])
def test_bench_line_keys(workload):
    env = dict(os.environ, COGNN_GATHER_GRID_CAP="1048576")          # (any COGNN_* variable set for the run must show up under `switches`)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "switches", "kernel_source_hash", "offline_ms", "offline_first_call_ms"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "u64" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["switches"] == {"COGNN_GATHER_GRID_CAP": "1048576"}
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and 0 < rf["frac"] < 1.5
    rows = rf["per_kernel"]
    assert len(rows) == 2 and {row["row_width_F"] for row in rows} == ({64, 16} if workload == "small" else {16, 7})
    for row in rows:                                          # each row stands on its own: achieved = algorithmic bytes / average launch time
        assert abs(row["achieved"] - row["algo_bytes_per_launch"] / 1e9 / (row["avg_ms"] / 1e3)) < 1e-6 * row["achieved"]
        assert abs(row["frac"] - row["achieved"] / 8000.0) < 1e-12 and row["launches"] > 0
    assert rf["kernel"] in {row["kernel"] for row in rows}
    assert d["check"]["cross_path_identical"] is True and 0 < d["ms_per_step_without_kernel_timers"] < 1.5 * d["ms_per_step"]
    assert d["offline_ms"] > 0 and d["offline_first_call_ms"] > 0
    if workload == "small":                                   # inference pass: the three dealer forms side by side, same shares
        for leg in ("dealer_streams", "dealer_minimal"):
            assert d[leg]["ms_per_step"] > 0 and d[leg]["shares_identical_to_in_register_form"] is True, d[leg]
    else:                                                     # training epoch: the offline phase is in the number too
        # (an epoch of this size is a chain of launch latencies: the dealer launches fill its gaps, and the timed region carries the
        # per-kernel HIP events while this loop does not - so the sum need not exceed the instrumented online epoch by the dealer phase)
        assert d["epoch_time_incl_offline_s"] > 0.8 * d["epoch_time_s"] > 0 and d["epoch_time_incl_offline_s"] > d["offline_ms"] / 1e3
