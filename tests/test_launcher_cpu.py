"""tools/run_cluster.py on CPU: ranks are tests/cpu_worker.py (cognn_amd/worker.py on the reference CPU backend) over gloo,
reference-format files written by the launcher's synthetic mode; per-party logs must carry the reference's log lines -
parsed here with the expressions of the reference's plot scripts - and the oracle's metrics."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import cognn_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCH = [sys.executable, os.path.join(ROOT, "tools", "run_cluster.py")]
CPU_WORKER = os.path.join(ROOT, "tests", "cpu_worker.py")
# tools/plot/plot_duration_breakdown_and_comm.py:99 (+ the two tags its list leaves out)
TAGS = ["preprocess", "PreScatterComp Client", "Scatter_preparation", "Scatter_computation", "premerging", "premerged_extraction",
        "Gather_computation", "Apply_computation", "PreScatterComp Server", "Gather_preparation", "iteration", "preprocess_OM"]


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def extract_cognn_durations(text, tag):
    """The parser of plot_duration_breakdown_and_comm.py:23-46, on a string."""
    out = []
    for line in text.splitlines():
        if "::" + tag + " took" in line:
            out.append(float(line.split(" took ")[1].split(" ")[0].strip()))
    return out


def extract_cognn_accuracies(text):
    """The parser of tools/plot/plot_accuracy.py:8-27, on a string."""
    test_acc, border_acc = [], []
    for line in text.splitlines():
        if "border test set accuracy" in line:
            border_acc.append(float(line.split("=")[1].strip()))
        elif "test set accuracy" in line:
            test_acc.append(float(line.split("=")[1].strip()))
    return test_acc, border_acc


def _oracle_for(data, dataset, k, setting, iters):
    from cognn_amd import worker
    src, dst = worker.read_edge_list(str(data / (dataset + ".edge.preprocessed")))
    part = worker.read_partition(str(data / (dataset + ".part.preprocessed")))
    p = co.GnnParam.read_config(str(data / (dataset + "_config.txt")))
    rows = worker.read_vertex_rows(str(data / (dataset + ".vertex.preprocessed")), set(range(len(part))), p.input_dim)
    feats = np.stack([rows[v][0] for v in range(len(part))]); labels = [rows[v][1] for v in range(len(part))]
    o = co.OracleEngine(k, src, dst, part, feats, labels, p, seed=worker.fnv1a(setting))
    o.run(iters)
    return o


@pytest.mark.parametrize("placement", ["party", "vertex-set"])
def test_two_rank_launcher_logs_match_oracle(tmp_path, placement):
    """(COGNN_PLACEMENT=vertex-set: every rank holds both shares of its party's vertex set - same logs, same cache files)"""
    data, logs = tmp_path / "data", tmp_path / "log"
    cmd = LAUNCH + ["--dataset", "cora_small", "--parties", "2", "--gpus", "2", "--iterations", "6", "--data-dir", str(data),
                    "--log-dir", str(logs), "--synthetic", "--backend", "gloo", "--worker", CPU_WORKER]
    env = dict(os.environ, OMP_NUM_THREADS="1", COGNN_PLACEMENT=placement)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=tmp_path)
    assert res.returncode == 0, res.stderr + res.stdout
    o = _oracle_for(data, "cora_small", 2, "gcn-optimize/cora_small/2s", 6)
    for party in range(2):
        text = (logs / ("gcn_test_cora_small_%d.log" % party)).read_text()       # tmp_run_cluster.py:146
        assert len(extract_cognn_durations(text, "iteration")) == 6
        # full iterations 0,1,3,5 print every tag once; the apply-only ones (2,4) none (ss_...h:709-732 has no print_duration)
        for tag in TAGS[1:10]:
            assert len(extract_cognn_durations(text, tag)) == 4, tag
        assert len(extract_cognn_durations(text, "preprocess")) == 1 and len(extract_cognn_durations(text, "preprocess_OM")) == 1
        want = [m for m in o.metrics if m["party"] == party][0]
        assert abs(float(re.findall(r"cross-entropy-loss = ([0-9.]+)", text)[0]) - want["loss"]) < 1e-6
        assert abs(float(re.findall(r"full set accuracy = ([0-9.]+)", text)[0]) - 100.0 * want["full"]) < 1e-5   # per cent
        test_acc, border_acc = extract_cognn_accuracies(text)                    # plot_accuracy.py:17-24
        assert len(test_acc) == 1 and len(border_acc) == 1
        assert abs(test_acc[0] - 100.0 * want["test"]) < 1e-5 and abs(border_acc[0] - 100.0 * want["border_test"]) < 1e-5
    # the offline cache of that run serves a second one started with -n 1 (same results)
    res2 = subprocess.run(cmd + ["--no-preprocess"], capture_output=True, text=True, timeout=300, env=env, cwd=tmp_path)
    assert res2.returncode == 0, res2.stderr + res2.stdout
    text2 = (logs / "gcn_test_cora_small_0.log").read_text()
    assert re.search(r"Reused [1-9][0-9]* offline products", text2)
    want0 = [m for m in o.metrics if m["party"] == 0][0]
    assert abs(float(re.findall(r"cross-entropy-loss = ([0-9.]+)", text2)[0]) - want0["loss"]) < 1e-6


def test_reference_experiment_option_builds_the_reference_layout(tmp_path):
    """--smallest-cognn-efficiency (tmp_run_cluster.py:438-448): gcn-optimize, Cora 2s, 2 parties, 12 iterations, with preprocessing."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    res = subprocess.run(LAUNCH + ["--smallest-cognn-efficiency", "--gpus", "2", "--backend", "gloo", "--worker", CPU_WORKER,
                                   "--root", str(tmp_path)], capture_output=True, text=True, timeout=600, env=env, cwd=tmp_path)
    assert res.returncode == 0, res.stderr + res.stdout
    logd = tmp_path / "cognn-smallest" / "log" / "gcn-optimize" / "cora" / "2s"
    datad = tmp_path / "cognn-smallest" / "data" / "Cora" / "transformed" / "2s"
    cfg = co.GnnParam.read_config(str(datad / "cora_config.txt"))
    assert cfg.input_dim == 1433 and cfg.num_samples == 2 * 542                  # two fifths of Cora's 2708 vertices
    for party in range(2):
        text = (logd / ("gcn_test_cora_%d.log" % party)).read_text()
        assert len(extract_cognn_durations(text, "iteration")) == 12
        acc = [float(x) for x in re.findall(r"full set accuracy = ([0-9.]+)", text)]
        assert len(acc) == 2                                                     # two epochs, like README.md:225-235
    assert (tmp_path / "preprocess" / "gcn-optimize" / "cora" / "2s").is_dir()    # the offline cache keyed by the setting


def test_unsupported_experiments_are_refused(tmp_path):
    res = subprocess.run(LAUNCH + ["--graphsc-efficiency"], capture_output=True, text=True, timeout=60)
    assert res.returncode == 2 and "GraphSC" in res.stderr


@pytest.mark.parametrize("gpus", [1, 2])
def test_gcn_original_through_the_launcher(tmp_path, gpus):
    """The unoptimised kernel (bin/gcn-original in the reference's scripts, --cognn-unopt-*): one process hosts both parties, or one
    rank each (the reference's deployment); two epochs of 4 GAS iterations; the per-party logs carry one accuracy block per epoch
    and the oracle's numbers."""
    import original_gcn
    from cognn_amd import worker
    data, logs = tmp_path / "data", tmp_path / "log"
    cmd = LAUNCH + ["--executable", "gcn-original", "--dataset", "cora_small", "--parties", "2", "--gpus", str(gpus), "--iterations", "8",
                    "--data-dir", str(data), "--log-dir", str(logs), "--synthetic", "--backend", "gloo", "--worker", CPU_WORKER]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    src, dst = worker.read_edge_list(str(data / "cora_small.edge.preprocessed"))
    part = worker.read_partition(str(data / "cora_small.part.preprocessed"))
    p = co.GnnParam.read_config(str(data / "cora_small_config.txt"))
    rows = worker.read_vertex_rows(str(data / "cora_small.vertex.preprocessed"), set(range(len(part))), p.input_dim)
    feats = np.stack([rows[v][0] for v in range(len(part))]); labels = [rows[v][1] for v in range(len(part))]
    o = original_gcn.OriginalOracleEngine(2, src, dst, part, feats, labels, p, seed=worker.fnv1a("gcn-original/cora_small/2s"))
    o.run(8)
    for party in range(2):
        text = (logs / ("gcn_test_cora_small_%d.log" % party)).read_text()
        assert len(extract_cognn_durations(text, "iteration")) == 8
        loss = [float(x) for x in re.findall(r"cross-entropy-loss = ([0-9.]+)", text)]
        want = [m["loss"] for m in o.metrics if m["party"] == party]
        assert len(loss) == 2 and len(want) == 2 and np.allclose(loss, want, atol=1e-5), (loss, want)


def test_rank_killed_by_signal_fails_the_launch(tmp_path):
    """A rank that dies by a signal (GPU abort, segfault) must fail the run and take its peers down instead of leaving them
    blocked in recv (the launcher used to report max(0, -11) == 0)."""
    bad = tmp_path / "dying_worker.py"
    bad.write_text("import os, signal, sys, time\n"
                   "if os.environ['RANK'] == '1':\n"
                   "    os.kill(os.getpid(), signal.SIGSEGV)\n"
                   "time.sleep(600)\n")
    data, logs = tmp_path / "data", tmp_path / "log"
    cmd = LAUNCH + ["--dataset", "cora_small", "--parties", "2", "--gpus", "2", "--iterations", "2", "--data-dir", str(data),
                    "--log-dir", str(logs), "--synthetic", "--backend", "gloo", "--worker", str(bad)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert res.returncode == 128 + 11, (res.returncode, res.stderr)
    # and a hung run is cut by --timeout
    hang = tmp_path / "hanging_worker.py"
    hang.write_text("import time\ntime.sleep(600)\n")
    res = subprocess.run(cmd[:-1] + [str(hang), "--timeout", "2"], capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert res.returncode == 124


def test_sigterm_to_the_launcher_ends_its_ranks(tmp_path):
    """The outer driver times the launcher out with SIGTERM: no rank may outlive it (a rank blocked in ncclRecv would hold its
    GPU until its own timeout)."""
    import signal
    import time
    hang = tmp_path / "hanging_worker.py"
    hang.write_text("import os, sys, time\n"
                    "open(os.path.join(%r, 'pid_' + os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                    "time.sleep(600)\n" % str(tmp_path))
    data, logs = tmp_path / "data", tmp_path / "log"
    cmd = LAUNCH + ["--dataset", "cora_small", "--parties", "2", "--gpus", "2", "--iterations", "2", "--data-dir", str(data),
                    "--log-dir", str(logs), "--synthetic", "--backend", "gloo", "--worker", str(hang)]
    launcher = subprocess.Popen(cmd, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    deadline = time.monotonic() + 60
    while time.monotonic() < deadline and not all((tmp_path / ("pid_%d" % r)).exists() and (tmp_path / ("pid_%d" % r)).read_text() for r in range(2)):
        time.sleep(0.1)
    pids = [int((tmp_path / ("pid_%d" % r)).read_text()) for r in range(2)]
    launcher.send_signal(signal.SIGTERM)
    assert launcher.wait(timeout=30) == 143

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        try:                                                 # (a zombie still answers signal 0)
            return open("/proc/%d/stat" % pid).read().split(")")[-1].split()[0] != "Z"
        except OSError:
            return False
    t_end = time.monotonic() + 10
    while time.monotonic() < t_end and any(alive(p) for p in pids):
        time.sleep(0.1)
    assert not any(alive(p) for p in pids), "ranks survived the launcher"


@pytest.mark.gpu
def test_launcher_runs_the_cpp_binary_on_the_gpu(tmp_path):
    """--gpus 1 without --worker: bin/gcn-optimize (no Python in the run) once per party log, all parties co-located on the GPU."""
    data, logs = tmp_path / "data", tmp_path / "log"
    cmd = LAUNCH + ["--dataset", "cora_small", "--parties", "2", "--iterations", "6", "--data-dir", str(data), "--log-dir", str(logs), "--synthetic"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert res.returncode == 0, res.stderr + res.stdout
    o = _oracle_for(data, "cora_small", 2, "gcn-optimize/cora_small/2s", 6)
    for party in range(2):
        text = (logs / ("gcn_test_cora_small_%d.log" % party)).read_text()
        assert len(extract_cognn_durations(text, "iteration")) == 6 and len(extract_cognn_durations(text, "premerging")) == 4
        want = [m for m in o.metrics if m["party"] == party][0]
        assert abs(float(re.findall(r"cross-entropy-loss = ([0-9.]+)", text)[0]) - want["loss"]) < 1e-6
    # the reference's cluster switch with a one-rank world: the RCCL bootstrap path of bin/gcn-optimize (-c 1) on one GPU
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    files = [str(data / ("cora_small" + e)) for e in (".edge.preprocessed", ".vertex.preprocessed", ".part.preprocessed")]
    r = subprocess.run([os.path.join(ROOT, "bin", "gcn-optimize"), "-t", "2", "-g", "2", "-i", "1", "-m", "2", "-s", "c1", "-r", "1", "-c", "1", "-n", "1"] + files +
                       [str(tmp_path / "out"), str(data / "cora_small_config.txt")], capture_output=True, text=True, timeout=120, cwd=tmp_path, env=env)
    assert r.returncode == 0, r.stderr
    assert "full set accuracy" in r.stdout


@pytest.mark.gpu
def test_launcher_runs_gcn_original_on_the_gpu(tmp_path):
    """bin/gcn-original (the reference's --cognn-unopt-* executable): the engine's original-gcn variant, 4 GAS iterations per epoch,
    both parties' logs with the oracle's loss after each of two epochs (one process hosts both parties here; one rank per party:
    tests/test_multirank_gpu.py::test_original_gcn_across_ranks_hip)."""
    import original_gcn
    from cognn_amd import worker
    data, logs = tmp_path / "data", tmp_path / "log"
    cmd = LAUNCH + ["--executable", "gcn-original", "--dataset", "cora_small", "--parties", "2", "--iterations", "8", "--data-dir", str(data),
                    "--log-dir", str(logs), "--synthetic"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert res.returncode == 0, res.stderr + res.stdout
    src, dst = worker.read_edge_list(str(data / "cora_small.edge.preprocessed"))
    part = worker.read_partition(str(data / "cora_small.part.preprocessed"))
    p = co.GnnParam.read_config(str(data / "cora_small_config.txt"))
    rows = worker.read_vertex_rows(str(data / "cora_small.vertex.preprocessed"), set(range(len(part))), p.input_dim)
    feats = np.stack([rows[v][0] for v in range(len(part))]); labels = [rows[v][1] for v in range(len(part))]
    o = original_gcn.OriginalOracleEngine(2, src, dst, part, feats, labels, p, seed=worker.fnv1a("gcn-original/cora_small/2s"))
    o.run(8)
    for party in range(2):
        text = (logs / ("gcn_test_cora_small_%d.log" % party)).read_text()
        assert len(extract_cognn_durations(text, "iteration")) == 8 and len(extract_cognn_durations(text, "premerging")) == 6
        loss = [float(x) for x in re.findall(r"cross-entropy-loss = ([0-9.]+)", text)]
        want = [m["loss"] for m in o.metrics if m["party"] == party]
        assert len(loss) == 2 and np.allclose(loss, want, atol=1e-5), (loss, want)
