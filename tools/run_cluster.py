#!/usr/bin/env python3
"""Launcher in the role of the reference's tools/tmp_run_cluster.py: starts the parties of one run and keeps its option
names, settings, directory layout and log naming, so tools/plot/*.py keep working:

    python tools/run_cluster.py --smallest-cognn-efficiency            # gcn-optimize, Cora "2s", 2 parties, 12 iterations
    python tools/run_cluster.py --cognn-opt-efficiency [--gpus N]      # gcn-optimize, {cora,citeseer,pubmed} x {2,3,4,5}s, 6 iterations
    python tools/run_cluster.py --cognn-opt-inference                  # gcn-inference-optimize, 2p, 2 iterations
    python tools/run_cluster.py --cognn-opt-accuracy[-no-preprocess]   # gcn-optimize, {2,3,4,5}p, 540 iterations (90 epochs)

(tmp_run_cluster.py:453-465; iterations and settings from :256-310, :396-448.)  Logs go to
<root>/log/<executable>/<dataset>/<N>{s,p}/[noPreprocess/]gcn_test_<dataset>_<party>.log like the reference's
(tmp_run_cluster.py:116-118, :146).  Where the reference starts one process per party inside a network namespace
(:224-241), this launcher either hosts all parties on one GPU (`--gpus 1`, default: bin/<executable> once per party log)
or starts one bin/<executable> -c 1 process per GPU (`--gpus N`, parties in contiguous blocks, shares over RCCL).
The --cognn-unopt-* experiments run bin/gcn-original (the unoptimised kernel) the same way; the FL / plaintext / GraphSC baselines are other programs of the reference and are refused with a message.
A single custom run keeps the explicit form:

    python tools/run_cluster.py --executable gcn-optimize --dataset pubmed --parties 4 --gpus 4 --iterations 12 \\
        --data-dir ./data/Pubmed/transformed/4s --log-dir ./log/gcn-optimize/pubmed/4s

Datasets are not shipped (the reference fetches Planetoid from the internet, tools/data_transform.py:31,69); missing files
are written as shape-matched synthetic stand-ins in the reference's formats (sizes from build_from_source/config/*.txt;
"<N>s" = the first N fifths of the vertices, one contiguous fifth per party, tools/data_transform.py:66-118; "<N>p" =
all vertices, vid % N, :19-27).
"""
import argparse
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = {  # dataset: (vertices, directed edges, input_dim, hidden, labels, lr, train, val, test)
    "cora": (2708, 10556, 1433, 16, 7, 0.5, 0.2, 0.2, 0.6),
    "citeseer": (3312, 10016, 3703, 16, 6, 0.8, 0.2, 0.2, 0.6),
    "pubmed": (19717, 128146, 500, 16, 3, 8.0, 0.05, 0.15, 0.8),
    "cora_small": (4, 8, 2, 3, 3, 0.5, 0.2, 0.2, 0.6),
}


def write_synthetic(data_dir, dataset, parties, kind="p", seed=1):
    """Shape-matched stand-in for <dataset> in the reference's file formats.  kind "p": all vertices, vid % parties;
    kind "s": `parties` fifths of the vertices, contiguous blocks (one fifth per party)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    V, E, in_dim, hid, lab, lr, tr, va, te = SHAPES[dataset]
    if kind == "s":
        fifth = -(-V // 5)
        bounds = [min(V, i * fifth) for i in range(parties + 1)]
        E = max(2, int(E * (bounds[-1] / V) ** 2) // 2 * 2)
        V = bounds[-1]
        owner = np.searchsorted(np.array(bounds[1:]), np.arange(V), side="right")
    else:
        owner = np.arange(V) % parties
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
            f.write("%d %d\n" % (v, owner[v]))
    with open(os.path.join(data_dir, dataset + ".vertex.preprocessed"), "w") as f:
        for v in range(V):
            f.write("%d %s %d\n" % (v, " ".join("%f" % x for x in feats[v]), labels[v]))
    with open(os.path.join(data_dir, dataset + "_config.txt"), "w") as f:
        f.write("num_layers : 2\nnum_labels : %d\ninput_dim : %d\nhidden_dim : %d\nnum_samples : %d\nnum_edges : %d\n"
                "learning_rate : %s\ntrain_ratio : %s\nval_ratio : %s\ntest_ratio : %s" % (lab, in_dim, hid, V, len(src), lr, tr, va, te))


LIVE = []                                                     # every child that may still run (the SIGTERM handler ends them)


def spawn(cmd, **kw):
    p = subprocess.Popen(cmd, **kw)
    LIVE.append(p)
    return p


def end_children(procs):
    """terminate(), then kill() after 5 s: a rank blocked in ncclRecv never exits by itself."""
    procs = [p for p in procs if p.poll() is None]
    for p in procs:
        p.terminate()
    t_end = time.monotonic() + 5
    for p in procs:
        try:
            p.wait(max(0.0, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


def wait_all(procs, timeout):
    """Waits for every child; on the first failure (non-zero exit or death by signal) or on timeout the others are terminated,
    then killed - and so they are when the launcher itself is interrupted.  Returns 0 only if all exited with 0; a signal -N
    maps to 128 + N."""
    deadline = time.monotonic() + timeout
    rc = 0
    live = list(procs)
    try:
        while live and rc == 0:
            for p in list(live):
                r = p.poll()
                if r is None:
                    continue
                live.remove(p)
                if r != 0:
                    rc = 128 - r if r < 0 else r
            if rc == 0 and live:
                if time.monotonic() > deadline:
                    rc = 124
                    break
                time.sleep(0.05)
    finally:                                                  # a rank died, the run timed out or the launcher is going down: the
        end_children(live)                                    # peers would block in recv (and hold their GPUs) forever
        for p in procs:
            if p in LIVE:
                LIVE.remove(p)
    return rc


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_one(executable, dataset, parties, iterations, setting, data_dir, log_dir, gpus=1, no_preprocess=False, worker=None,
            backend="nccl", timeout=3600.0, part_suffix=""):
    files = [os.path.join(data_dir, dataset + ".edge.preprocessed"), os.path.join(data_dir, dataset + ".vertex.preprocessed"),
             os.path.join(data_dir, dataset + ".part.preprocessed" + part_suffix),
             os.path.join(log_dir, "gcn_test.result." + dataset), os.path.join(data_dir, dataset + "_config.txt")]
    os.makedirs(log_dir, exist_ok=True)
    common = ["-t", str(parties), "-g", str(parties), "-m", str(iterations), "-p", "1", "-s", setting, "-r", "1"]
    if no_preprocess:
        common += ["-n", "1"]
    log_name = lambda i: os.path.join(log_dir, "gcn_test_%s_%d.log" % (dataset, i))   # tmp_run_cluster.py:146
    exe = os.path.join(ROOT, "bin", executable)
    if worker is None and gpus == 1:
        rc = 0
        for i in range(parties):                              # one log per party, like the reference; same co-located run each time
            cmd = [exe] + common + ["-i", str(i)] + files
            print(" ".join(cmd), flush=True)
            with open(log_name(i), "w") as lf:
                rc = rc or wait_all([spawn(cmd, stdout=lf)], timeout)   # sequential: they share the one GPU
        return rc
    if parties % gpus:
        raise SystemExit("the %d parties must divide evenly over %d ranks" % (parties, gpus))
    port = free_port()
    per = parties // gpus
    procs, logs = [], []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        if worker is None:                                    # C++ rank: its first hosted party's log on stdout, the others' through COGNN_LOG_PREFIX
            cmd = [exe] + common + ["-c", "1", "-i", str(r * per)] + files
            lf = open(log_name(r * per), "w")
            logs.append(lf)
            env["COGNN_LOG_PREFIX"] = os.path.join(log_dir, "gcn_test_%s_" % dataset)
            procs.append(spawn(cmd, env=env, stdout=lf))
        else:                                                 # Python rank (cognn_amd.worker or a test wrapper of it): one log per hosted party
            variant = "optimize-gcn-inference" if "inference" in executable else "original-gcn" if "original" in executable else "optimize-gcn"
            head = [sys.executable] + (["-m", worker] if not worker.endswith(".py") else [worker])
            cmd = head + common + ["--variant", variant, "--backend", backend, "--log-dir", log_dir, "--log-prefix", "gcn_test_%s_" % dataset] + files
            procs.append(spawn(cmd, env=env))
        print("RANK=%d " % r + " ".join(cmd), flush=True)
    rc = wait_all(procs, timeout)
    for lf in logs:
        lf.close()
    return rc


# the reference's experiments (tmp_run_cluster.py): name -> (executable, datasets, party counts, "s" | "p", iterations, preprocess passes)
EXPERIMENTS = {
    "smallest_cognn_efficiency": ("gcn-optimize", ["cora"], [2], "s", 12, [True], "cognn-smallest"),                     # :438-448
    "cognn_opt_efficiency": ("gcn-optimize", ["cora", "citeseer", "pubmed"], [2, 3, 4, 5], "s", 6, [True, False], "cognn-scale"),   # :256-276
    "cognn_opt_inference": ("gcn-inference-optimize", ["cora", "citeseer", "pubmed"], [2], "p", 2, [True, False], "inference"),     # :396-415
    "cognn_opt_accuracy": ("gcn-optimize", ["cora", "citeseer", "pubmed"], [2, 3, 4, 5], "p", 540, [True], "mp-accuracy"),          # :159-168
    "cognn_opt_accuracy_no_preprocess": ("gcn-optimize", ["cora", "citeseer", "pubmed"], [2], "p", 540, [False], "mp-accuracy"),    # :170-180
    # the unoptimised kernel: 4 GAS iterations = one epoch, 2 = one inference
    "cognn_unopt_accuracy": ("gcn-original", ["cora", "citeseer", "pubmed"], [2], "p", 4, [True], "mp-accuracy"),                   # :287-296
    "cognn_unopt_accuracy_no_preprocess": ("gcn-original", ["cora", "citeseer", "pubmed"], [2], "p", 4, [False], "mp-accuracy"),    # :298-307
    "cognn_unopt_efficiency": ("gcn-original", ["cora", "citeseer", "pubmed"], [2, 3, 4, 5], "s", 4, [True, False], "cognn-scale"),  # :380-395
    "cognn_unopt_inference": ("gcn-original", ["cora", "citeseer", "pubmed"], [2], "p", 2, [True, False], "inference"),              # :419-435
}
UNSUPPORTED = ["fedgnn_accuracy", "plaintextgnn_accuracy", "graphsc_efficiency"]


def run_experiment(name, a):
    executable, datasets, counts, kind, iterations, passes, app = EXPERIMENTS[name]
    root = os.path.join(a.root, app)
    rc = 0
    for pre in passes:
        for dataset in datasets:
            for n in counts:
                sub = "%d%s" % (n, kind)
                data_dir = os.path.join(root, "data", dataset[:1].upper() + dataset[1:], "transformed", sub)
                if not os.path.exists(os.path.join(data_dir, dataset + "_config.txt")):
                    write_synthetic(data_dir, dataset, n, kind)
                log_dir = os.path.join(root, "log", executable, dataset, sub) + ("" if pre else "/noPreprocess")
                setting = "%s/%s/%s" % (executable, dataset, sub)                        # tmp_run_cluster.py:124, :220
                gpus = max(d for d in range(1, a.gpus + 1) if n % d == 0)      # ranks must host equally many parties
                rc = rc or run_one(executable, dataset, n, iterations, setting, data_dir, log_dir, gpus=gpus, no_preprocess=not pre,
                                   worker=a.worker, backend=a.backend, timeout=a.timeout)
    return rc


def main():
    ap = argparse.ArgumentParser(description="Evaluate CoGNN on MI355X (options of the reference's tools/tmp_run_cluster.py).")
    for name in list(EXPERIMENTS) + UNSUPPORTED:
        ap.add_argument("--" + name.replace("_", "-"), action="store_true")
    ap.add_argument("--all", action="store_true", help="every supported experiment")
    ap.add_argument("--root", default=".", help="where <application>/{data,log} are created (the reference uses the cwd)")
    ap.add_argument("--executable", default="gcn-optimize", choices=["gcn-optimize", "gcn-inference-optimize", "gcn-original"])
    ap.add_argument("--dataset", default="cora")
    ap.add_argument("--parties", type=int, default=2)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--iterations", type=int, default=12)
    ap.add_argument("--setting", default=None)
    ap.add_argument("--data-dir", default=None)
    ap.add_argument("--log-dir", default=None)
    ap.add_argument("--no-preprocess", action="store_true", help="-n 1")
    ap.add_argument("--synthetic", action="store_true", help="write shape-matched synthetic files into --data-dir first")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="transport of Python ranks (--worker)")
    ap.add_argument("--worker", default=None, help="start Python ranks (module or script, e.g. cognn_amd.worker) instead of bin/<executable> -c 1")
    ap.add_argument("--timeout", type=float, default=3600.0, help="seconds per run before the ranks are killed")
    ap.add_argument("--placement", default=None, choices=["party", "vertex-set"],
                    help="--gpus N > 1: which rank holds which share (COGNN_PLACEMENT of every rank; default: party, the reference's deployment)")
    a = ap.parse_args()
    if a.placement:
        os.environ["COGNN_PLACEMENT"] = a.placement           # inherited by every rank this launcher starts
    for name in UNSUPPORTED:
        if getattr(a, name):
            print("--%s: the FL, plaintext and GraphSC baselines are other programs of the reference, not part of this engine"
                  % name.replace("_", "-"), file=sys.stderr)
            return 2
    chosen = [n for n in EXPERIMENTS if getattr(a, n) or a.all]
    if chosen:
        rc = 0
        for n in chosen:
            rc = rc or run_experiment(n, a)
        return rc
    if not a.data_dir or not a.log_dir:
        ap.error("a custom run needs --data-dir and --log-dir (or pick one of the experiment options)")
    if a.synthetic:
        write_synthetic(a.data_dir, a.dataset, a.parties)
    setting = a.setting or "%s/%s/%ds" % (a.executable, a.dataset, a.parties)
    return run_one(a.executable, a.dataset, a.parties, a.iterations, setting, a.data_dir, a.log_dir, gpus=a.gpus,
                   no_preprocess=a.no_preprocess, worker=a.worker, backend=a.backend, timeout=a.timeout)


def on_sigterm(*_):
    end_children(list(LIVE))                                  # no rank outlives the launcher
    sys.exit(143)


if __name__ == "__main__":
    signal.signal(signal.SIGTERM, on_sigterm)
    sys.exit(main())
