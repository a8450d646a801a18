"""The command-line entry point with the reference's file formats (edge list / vertex list / partition / config) and
log lines, exercised on CPU through oracle/gcn-optimize-cpuref (same harness_main.cpp, graph.cpp and engine.cpp as
bin/gcn-optimize, linked against the reference backend).  Metrics are compared with the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

import cognn_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "gcn-optimize-cpuref")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def _write_inputs(d, k, V, Eu, in_dim, lab, hid):
    src, dst = co.synth_graph(V, Eu, 9)
    feats, labels = co.synth_features(V, in_dim, lab, 10, density=0.3)
    with open(d / "edges.txt", "w") as f:                      # format: graph_io_util.h:121-147 ('#' comments, blank lines)
        f.write("# src dst\n\n")
        for s, t in zip(src, dst):
            f.write("%d %d\n" % (s, t))
    with open(d / "part.txt", "w") as f:                       # graph_io_util.h:67-73
        for v in range(V):
            f.write("%d %d\n" % (v, v % k))
    with open(d / "vertices.txt", "w") as f:                   # tools/data_transform.py:55-56: '%d %f ... %d'
        for v in range(V):
            f.write("%d %s %d\n" % (v, " ".join("%f" % x for x in feats[v]), labels[v]))
    with open(d / "config.txt", "w") as f:                     # task.h:126-160
        f.write("num_layers : 2\nnum_labels : %d\ninput_dim : %d\nhidden_dim : %d\nnum_samples : %d\nnum_edges : %d\n"
                "learning_rate : 0.5\ntrain_ratio : 0.2\nval_ratio : 0.2\ntest_ratio : 0.6" % (lab, in_dim, hid, V, len(src)))
    return src, dst, feats, labels


def _fnv1a(s):
    h = 0xcbf29ce484222325
    for c in s.encode():
        h = ((h ^ c) * 0x100000001b3) & ((1 << 64) - 1)
    return h


def test_cli_runs_reference_formats_and_matches_oracle(tmp_path):
    k, V, Eu, in_dim, lab, hid = 2, 40, 90, 6, 3, 4
    src, dst, feats, labels = _write_inputs(tmp_path, k, V, Eu, in_dim, lab, hid)
    setting = "cora-2s-test"
    cmd = [BIN, "-t", str(k), "-g", str(k), "-i", "1", "-m", "12", "-p", "1", "-s", setting, "-r", "1",
           str(tmp_path / "edges.txt"), str(tmp_path / "vertices.txt"), str(tmp_path / "part.txt"), str(tmp_path / "out.txt"),
           str(tmp_path / "config.txt")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=tmp_path)     # preprocess/<setting>/ lands in cwd
    assert res.returncode == 0, res.stderr
    out = res.stdout
    assert len(re.findall(r"::iteration took [0-9.]+ seconds", out)) == 12
    assert "::preprocess took" in out and "::preprocess_OM took" in out
    p = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    o = co.OracleEngine(k, src, dst, [v % k for v in range(V)], feats, labels, p, seed=_fnv1a(setting))
    o.run(12)
    want = [m for m in o.metrics if m["party"] == 1]
    for key, pat in (("loss", r"cross-entropy-loss = ([0-9.]+)"), ("full", r"full set accuracy = ([0-9.]+)"),
                     ("train", r"training set accuracy = ([0-9.]+)\nborder"), ("test", r"\ntest set accuracy = ([0-9.]+)"),
                     ("border_test", r"border test set accuracy = ([0-9.]+)")):
        got = [float(x) for x in re.findall(pat, out)]
        assert len(got) == 2
        for g, w in zip(got, want):                      # accuracies are logged in per cent (README.md:226-236)
            assert abs(g - w[key] * (1.0 if key == "loss" else 100.0)) < 1e-5, (key, got, [x[key] for x in want])
    nv = re.findall(r"the number of vertices is (\d+), the number of border vertices is (\d+)", out)
    assert [int(x) for x in nv[0]] == [want[0]["n"], want[0]["n_border"]]


def test_one_log_per_hosted_party(tmp_path):
    """COGNN_LOG_PREFIX (set by tools/run_cluster.py for C++ ranks): a process that hosts several parties writes the log of every
    hosted party - the one named by -i to stdout, the others to <prefix><party>.log - each with its own party's metrics, so the
    reference's one-log-per-party layout (tmp_run_cluster.py:146) is complete when parties > ranks."""
    k, V, Eu, in_dim, lab, hid = 3, 45, 100, 6, 3, 4
    src, dst, feats, labels = _write_inputs(tmp_path, k, V, Eu, in_dim, lab, hid)
    setting = "three-on-one"
    cmd = [BIN, "-t", str(k), "-g", str(k), "-i", "1", "-m", "2", "-p", "1", "-s", setting, "-r", "1",
           str(tmp_path / "edges.txt"), str(tmp_path / "vertices.txt"), str(tmp_path / "part.txt"), str(tmp_path / "out.txt"),
           str(tmp_path / "config.txt")]
    env = dict(os.environ, COGNN_LOG_PREFIX=str(tmp_path / "gcn_test_x_"))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=tmp_path, env=env)
    assert res.returncode == 0, res.stderr
    p = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    o = co.OracleEngine(k, src, dst, [v % k for v in range(V)], feats, labels, p, seed=_fnv1a(setting))
    o.run(2)
    texts = {1: res.stdout, 0: (tmp_path / "gcn_test_x_0.log").read_text(), 2: (tmp_path / "gcn_test_x_2.log").read_text()}
    assert not (tmp_path / "gcn_test_x_1.log").exists()
    for party, text in texts.items():
        want = [m for m in o.metrics if m["party"] == party][0]
        assert abs(float(re.findall(r"cross-entropy-loss = ([0-9.]+)", text)[0]) - want["loss"]) < 1e-5
        assert len(re.findall(r"::iteration took [0-9.]+ seconds", text)) == 2 and "::premerging took" in text
        assert "tid-> %d, iteration-> 1" % party in text and "%d Finish algo kernel" % party in text


def test_cli_argument_errors(tmp_path):
    r = subprocess.run([BIN, "-h"], capture_output=True, text=True)
    assert r.returncode != 0 and "Usage" in r.stderr
    r = subprocess.run([BIN, "-t", "2", "-g", "3", "a", "b", "c", "d", "e"], capture_output=True, text=True)
    assert r.returncode != 0 and "divisor" in r.stderr
    r = subprocess.run([BIN, "a"], capture_output=True, text=True)
    assert r.returncode != 0 and "Must specify number of threads" in r.stderr
    _write_inputs(tmp_path, 2, 20, 30, 4, 3, 3)
    base = [BIN, "-t", "2", "-g", "2", "-i", "0", "-m", "2", "-s", "x"]
    files = [str(tmp_path / n) for n in ("edges.txt", "vertices.txt", "part.txt", "out.txt", "config.txt")]
    r = subprocess.run(base + files, capture_output=True, text=True)            # -r 1 missing
    assert r.returncode != 0 and "no-dummy-edge" in r.stderr
    r = subprocess.run(base + ["-r", "1"] + [str(tmp_path / "missing.txt")] + files[1:], capture_output=True, text=True)
    assert r.returncode != 0 and "cannot open edge list file" in r.stderr
    (tmp_path / "bad.txt").write_text("0 x\n")
    r = subprocess.run(base + ["-r", "1", str(tmp_path / "bad.txt")] + files[1:], capture_output=True, text=True)
    assert r.returncode != 0 and "Invalid format in graph topology input files." in r.stderr
    # a misspelt placement is an error, not a silent fall-back to the party placement (ranks that disagree would hang in the transport)
    r = subprocess.run(base + ["-r", "1"] + files, capture_output=True, text=True, env=dict(os.environ, COGNN_PLACEMENT="vertexset"), cwd=tmp_path)
    assert r.returncode != 0 and "COGNN_PLACEMENT must be" in r.stderr


@pytest.mark.gpu
def test_gpu_binary_matches_oracle(tmp_path):
    """bin/gcn-optimize (HIP engine, no torch in the process) on the same files."""
    exe = os.path.join(ROOT, "bin", "gcn-optimize")
    assert os.path.exists(exe), "bin/gcn-optimize missing: run make"
    k, V, Eu, in_dim, lab, hid = 3, 60, 140, 8, 4, 6
    src, dst, feats, labels = _write_inputs(tmp_path, k, V, Eu, in_dim, lab, hid)
    cmd = [exe, "-t", str(k), "-g", str(k), "-i", "0", "-m", "6", "-s", "gpu-harness", "-r", "1", "-n", "1",
           str(tmp_path / "edges.txt"), str(tmp_path / "vertices.txt"), str(tmp_path / "part.txt"), str(tmp_path / "out.txt"),
           str(tmp_path / "config.txt")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert res.returncode == 0, res.stderr
    p = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, learning_rate=0.5)
    o = co.OracleEngine(k, src, dst, [v % k for v in range(V)], feats, labels, p, seed=_fnv1a("gpu-harness"))
    o.run(6)
    want = [m for m in o.metrics if m["party"] == 0][0]
    assert abs(float(re.findall(r"cross-entropy-loss = ([0-9.]+)", res.stdout)[0]) - want["loss"]) < 1e-6
    assert abs(float(re.findall(r"full set accuracy = ([0-9.]+)", res.stdout)[0]) - 100.0 * want["full"]) < 1e-5


def _offline_cache_roundtrip(exe, tmp_path):
    """`-n 1` reuses the preprocess/<setting>/ products written by an earlier run; the results do not change."""
    k, V, Eu, in_dim, lab, hid = 2, 40, 90, 8, 3, 4
    _write_inputs(tmp_path, k, V, Eu, in_dim, lab, hid)
    files = [str(tmp_path / n) for n in ("edges.txt", "vertices.txt", "part.txt", "out.txt", "config.txt")]
    base = [exe, "-t", "2", "-g", "2", "-i", "0", "-m", "12", "-s", "cache/test-1", "-r", "1"]
    r1 = subprocess.run(base + files, capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r1.returncode == 0, r1.stderr
    cached = os.listdir(tmp_path / "preprocess" / "cache" / "test-1")
    assert len(cached) == 2 * 2 * 5        # 2 epochs x 2 co-party sides x (2 forward + 3 backward) Beaver products per epoch
    r2 = subprocess.run(base + ["-n", "1"] + files, capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r2.returncode == 0, r2.stderr
    assert "Reused 10 offline products" in r2.stdout           # the first epoch's; the second epoch's are loaded when it starts
    pick = lambda out: re.findall(r"(cross-entropy-loss|accuracy) = ([0-9.]+)", out)
    assert pick(r1.stdout) == pick(r2.stdout) and len(pick(r1.stdout)) == 12
    # a cache written for another shape is ignored, not trusted (stale preprocess/<setting>/ of another dataset)
    _write_inputs(tmp_path, k, V + 10, Eu, in_dim, lab, hid)
    r3 = subprocess.run(base + ["-n", "1"] + files, capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r3.returncode == 0, r3.stderr
    assert "Reused 0 offline products" in r3.stdout
    # the setting string never reaches a shell
    evil = "x'; touch INJECTED; echo '"
    r4 = subprocess.run(base[:9] + ["-s", evil, "-r", "1"] + files, capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r4.returncode == 0 and not (tmp_path / "INJECTED").exists()


def _binary_container(exe, tmp_path):
    k, V, Eu, in_dim, lab, hid = 2, 40, 90, 8, 3, 4
    _write_inputs(tmp_path, k, V, Eu, in_dim, lab, hid)
    subprocess.check_call([os.sys.executable, os.path.join(ROOT, "tools", "convert_graph.py"), str(tmp_path / "edges.txt"),
                           str(tmp_path / "part.txt"), str(tmp_path / "graph.cgb")])
    tail = [str(tmp_path / n) for n in ("vertices.txt", "part.txt", "out.txt", "config.txt")]
    base = [exe, "-t", "2", "-g", "2", "-i", "1", "-m", "2", "-s", "bin", "-r", "1", "-n", "1"]
    a = subprocess.run(base + [str(tmp_path / "edges.txt")] + tail, capture_output=True, text=True, cwd=tmp_path, timeout=120)
    b = subprocess.run(base + [str(tmp_path / "graph.cgb")] + tail[:1] + ["ignored"] + tail[2:], capture_output=True, text=True,
                       cwd=tmp_path, timeout=120)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    pick = lambda out: re.findall(r"(cross-entropy-loss|accuracy) = ([0-9.]+)", out)
    assert pick(a.stdout) == pick(b.stdout) and len(pick(a.stdout)) == 6


def _log_contract(exe, tmp_path):
    """The tags of the reference's print_duration sites, parsed like tools/plot/plot_duration_breakdown_and_comm.py:23-46 does;
    the `iteration` line covers the device work of its iteration (the harness synchronises before it prints)."""
    k, V, Eu, in_dim, lab, hid = 2, 60, 150, 8, 3, 4
    _write_inputs(tmp_path, k, V, Eu, in_dim, lab, hid)
    files = [str(tmp_path / n) for n in ("edges.txt", "vertices.txt", "part.txt", "out.txt", "config.txt")]
    r = subprocess.run([exe, "-t", "2", "-g", "2", "-i", "0", "-m", "6", "-s", "tags", "-r", "1"] + files, capture_output=True, text=True,
                       cwd=tmp_path, timeout=120)
    assert r.returncode == 0, r.stderr
    blocks = r.stdout.split("tid-> 0, iteration-> ")[1:]
    assert len(blocks) == 6
    tags = ["PreScatterComp Client", "PreScatterComp Server", "Scatter_preparation", "Scatter_computation", "premerging", "premerged_extraction",
            "Gather_preparation", "Gather_computation", "Apply_computation"]
    for it, blk in enumerate(blocks):
        dur = {t: [float(l.split(" took ")[1].split(" ")[0]) for l in blk.splitlines() if "::" + t + " took" in l] for t in tags + ["iteration"]}
        assert len(dur["iteration"]) == 1
        apply_only = it in (2, 4)                                  # ss_...h:709: no print_duration on that path
        for t in tags:
            assert len(dur[t]) == (0 if apply_only else 1), (it, t)
        if not apply_only:
            phases = dur["PreScatterComp Client"][0] + dur["premerging"][0] + dur["Gather_computation"][0] + dur["Apply_computation"][0]
            # (iteration 1: the prediction layer rides in the label-wide Gather's launch - cognn_gather_pair::softmax -, so ApplyComp's own
            # phase is empty there and its time is part of "premerging")
            assert dur["premerging"][0] > 0 and (dur["Apply_computation"][0] > 0 or it == 1)
            assert dur["iteration"][0] >= 0.9 * phases, (it, dur)   # the wall time of an iteration covers its device phases
    assert "::preprocess took" in r.stdout and "::preprocess_OM took" in r.stdout


def test_offline_cache_roundtrip(tmp_path):
    _offline_cache_roundtrip(BIN, tmp_path)


def test_binary_graph_container_gives_identical_run(tmp_path):
    _binary_container(BIN, tmp_path)


def test_log_contract_tags(tmp_path):
    _log_contract(BIN, tmp_path)


HIP_BIN = os.path.join(ROOT, "bin", "gcn-optimize")


@pytest.mark.gpu
def test_gpu_offline_cache_roundtrip(tmp_path):
    _offline_cache_roundtrip(HIP_BIN, tmp_path)


@pytest.mark.gpu
def test_gpu_binary_graph_container(tmp_path):
    _binary_container(HIP_BIN, tmp_path)


@pytest.mark.gpu
def test_gpu_log_contract_tags(tmp_path):
    _log_contract(HIP_BIN, tmp_path)
