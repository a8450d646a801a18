#!/usr/bin/env python3
"""Launcher in the role of the reference's tools/tmp_run_cluster.py: starts the k parties of one run.  Where the reference
starts one process per party inside a network namespace (tmp_run_cluster.py:224-241), this one starts one process per GPU
(`--gpus N`, parties in contiguous blocks) and lets them exchange shares over RCCL; per-party logs keep the reference's
naming (`<log dir>/gcn_test_<party>.log`) and line formats so tools/plot/*.py keep working.

    python tools/run_cluster.py --executable gcn-optimize --parties 4 --gpus 4 --iterations 12 --setting gcn-optimize/pubmed/4s \
        --data-dir ./data/Pubmed/transformed/4s --dataset pubmed --log-dir ./log/gcn-optimize/pubmed/4s

With --gpus 1 (default) the C++ binary bin/gcn-optimize is used directly for party 0's log (all parties co-located).
Datasets are not shipped (the reference fetches Planetoid from the internet); --synthetic writes shape-matched files in the
reference's formats into --data-dir first (sizes from build_from_source/config/*.txt).
"""
import argparse
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = {  # dataset: (vertices, directed edges, input_dim, hidden, labels, lr, train, val, test)
    "cora": (2708, 10556, 1433, 16, 7, 0.5, 0.2, 0.2, 0.6),
    "citeseer": (3312, 10016, 3703, 16, 6, 0.8, 0.2, 0.2, 0.6),
    "pubmed": (19717, 128146, 500, 16, 3, 8.0, 0.05, 0.15, 0.8),
    "cora_small": (4, 8, 2, 3, 3, 0.5, 0.2, 0.2, 0.6),
}


def write_synthetic(data_dir, dataset, parties, seed=1):
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    V, E, in_dim, hid, lab, lr, tr, va, te = SHAPES[dataset]
    os.makedirs(data_dir, exist_ok=True)
    src, dst = bench.synth_graph(V, E // 2, seed)
    rng = np.random.default_rng(seed + 1)
    feats = (rng.random((V, in_dim)) < 0.01).astype(int)
    labels = rng.integers(0, lab, size=V)
    with open(os.path.join(data_dir, dataset + ".edge.preprocessed"), "w") as f:
        for s, d in zip(src, dst):
            f.write("%d %d\n" % (s, d))
    with open(os.path.join(data_dir, dataset + ".part.preprocessed"), "w") as f:
        for v in range(V):
            f.write("%d %d\n" % (v, v % parties))               # tools/data_transform.py:19-27
    with open(os.path.join(data_dir, dataset + ".vertex.preprocessed"), "w") as f:
        for v in range(V):
            f.write("%d %s %d\n" % (v, " ".join("%f" % x for x in feats[v]), labels[v]))
    with open(os.path.join(data_dir, dataset + "_config.txt"), "w") as f:
        f.write("num_layers : 2\nnum_labels : %d\ninput_dim : %d\nhidden_dim : %d\nnum_samples : %d\nnum_edges : %d\n"
                "learning_rate : %s\ntrain_ratio : %s\nval_ratio : %s\ntest_ratio : %s" % (lab, in_dim, hid, V, len(src), lr, tr, va, te))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--executable", default="gcn-optimize", choices=["gcn-optimize", "gcn-inference-optimize"])
    ap.add_argument("--dataset", default="cora")
    ap.add_argument("--parties", type=int, default=2)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--iterations", type=int, default=12)
    ap.add_argument("--setting", default=None)
    ap.add_argument("--data-dir", required=True)
    ap.add_argument("--log-dir", required=True)
    ap.add_argument("--no-preprocess", action="store_true", help="-n 1")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    if a.synthetic:
        write_synthetic(a.data_dir, a.dataset, a.parties)
    setting = a.setting or "%s/%s/%ds" % (a.executable, a.dataset, a.parties)
    files = [os.path.join(a.data_dir, a.dataset + ext) for ext in (".edge.preprocessed", ".vertex.preprocessed", ".part.preprocessed")]
    files += [os.path.join(a.log_dir, "gcn_test.result." + a.dataset), os.path.join(a.data_dir, a.dataset + "_config.txt")]
    os.makedirs(a.log_dir, exist_ok=True)
    common = ["-t", str(a.parties), "-g", str(a.parties), "-m", str(a.iterations), "-p", "1", "-s", setting, "-r", "1"]
    if a.no_preprocess:
        common += ["-n", "1"]
    if a.gpus == 1 and a.backend == "nccl" and a.lib is None:
        exe = os.path.join(ROOT, "bin", a.executable)
        procs = []
        for i in range(a.parties):                              # one log per party, like the reference; same co-located run
            cmd = [exe] + common + ["-i", str(i)] + files
            print(" ".join(cmd))
            with open(os.path.join(a.log_dir, "gcn_test_%d.log" % i), "w") as lf:
                procs.append(subprocess.Popen(cmd, stdout=lf))
                procs[-1].wait()                                # sequential: they share the one GPU
        return max(p.returncode for p in procs)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    variant = "optimize-gcn-inference" if "inference" in a.executable else "optimize-gcn"
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        cmd = [sys.executable, "-m", "cognn_amd.worker"] + common + ["--variant", variant, "--backend", a.backend, "--log-dir", a.log_dir]
        if a.lib:
            cmd += ["--lib", a.lib]
        cmd += files
        print("RANK=%d " % r + " ".join(cmd))
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    return rc


if __name__ == "__main__":
    sys.exit(main())
