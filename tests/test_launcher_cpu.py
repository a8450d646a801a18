"""tools/run_cluster.py + cognn_amd/worker.py on CPU: 2 ranks over gloo with the reference CPU backend, reference-format
files written by the launcher's --synthetic mode; per-party logs must carry the reference's log lines and the oracle's metrics."""
import os
import re
import subprocess
import sys

import numpy as np

import cognn_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_launcher_logs_match_oracle(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    data, logs = tmp_path / "data", tmp_path / "log"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_cluster.py"), "--dataset", "cora_small", "--parties", "2", "--gpus", "2",
           "--iterations", "6", "--data-dir", str(data), "--log-dir", str(logs), "--synthetic", "--backend", "gloo",
           "--lib", os.path.join(ROOT, "oracle", "libcognn_engine_cpu.so")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr + res.stdout
    from cognn_amd import worker
    src, dst = worker.read_edge_list(str(data / "cora_small.edge.preprocessed"))
    part = worker.read_partition(str(data / "cora_small.part.preprocessed"))
    p = co.GnnParam.read_config(str(data / "cora_small_config.txt"))
    rows = worker.read_vertex_rows(str(data / "cora_small.vertex.preprocessed"), set(range(len(part))), p.input_dim)
    feats = np.stack([rows[v][0] for v in range(len(part))]); labels = [rows[v][1] for v in range(len(part))]
    o = co.OracleEngine(2, src, dst, part, feats, labels, p, seed=worker.fnv1a("gcn-optimize/cora_small/2s"))
    o.run(6)
    for party in range(2):
        text = (logs / ("gcn_test_%d.log" % party)).read_text()
        assert len(re.findall(r"::iteration took [0-9.]+ seconds", text)) == 6
        want = [m for m in o.metrics if m["party"] == party][0]
        assert abs(float(re.findall(r"cross-entropy-loss = ([0-9.]+)", text)[0]) - want["loss"]) < 1e-6
        assert abs(float(re.findall(r"full set accuracy = ([0-9.]+)", text)[0]) - want["full"]) < 1e-6
