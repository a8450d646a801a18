#!/usr/bin/env python3
"""bench.py — secret-shared GCN hot path on MI355X.

A "step" is one pass of the hot path over one synthetic batch: by default the 8-party
optimize-gcn-inference pass (GAS iterations 0-1, `-m 2` in the reference, tools/tmp_run_cluster.py:397-415)
on the synthetic 2^20-vertex / 2^24-edge global graph of BASELINE.json configs[4], partition vid % 8.
The k parties are mapped onto the N GPUs in contiguous blocks (N=1: all co-located, in-device exchange;
N=8: one party per GPU, RCCL p2p), so the total work is fixed as N grows ("strong" scaling).

`value` = whole-job edge-features aggregated per second: (directed edges) x (sum of the message widths of
the step's message-passing rounds) / step time, inputs already resident in HBM.  The dealer ("offline")
phase — Beaver product shares of the GEMMs — runs before the timed region, like the reference's
preprocess phase (README.md:236-237), and is reported separately as offline_ms.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def synth_graph(num_vertices, num_undirected, seed):
    """Distinct undirected pairs without self loops, both directions emitted."""
    rng = np.random.default_rng(seed)
    keys = np.empty(0, dtype=np.int64)
    while len(keys) < num_undirected:
        m = num_undirected - len(keys)
        a = rng.integers(0, num_vertices, size=m + m // 8 + 16, dtype=np.int64)
        b = rng.integers(0, num_vertices, size=m + m // 8 + 16, dtype=np.int64)
        ok = a != b
        cand = np.minimum(a, b)[ok] * np.int64(num_vertices) + np.maximum(a, b)[ok]
        fresh = np.setdiff1d(cand, keys)
        if len(fresh) > m:
            fresh = rng.permutation(fresh)[:m]
        keys = np.union1d(keys, fresh)
    lo = keys // num_vertices; hi = keys % num_vertices
    return np.concatenate([lo, hi]), np.concatenate([hi, lo])


WORKLOADS = {
    # name: (parties, log2 V, log2 directed E, in, hid, labels, variant, iterations per step)
    "config5": (8, 20, 24, 128, 64, 16, "optimize-gcn-inference", 2),
    "config5-h16": (8, 20, 24, 128, 16, 16, "optimize-gcn-inference", 2),
    "config5-train": (8, 20, 24, 128, 64, 16, "optimize-gcn", 6),
    "small": (8, 14, 18, 128, 64, 16, "optimize-gcn-inference", 2),
    "config5-x4": (8, 22, 26, 128, 64, 16, "optimize-gcn-inference", 2),     # 4x config5: size-scaling check (2^22 vertices / 2^26 edges)
    "config5-x8": (8, 23, 27, 128, 64, 16, "optimize-gcn-inference", 2),     # 8x (2^23 / 2^27)
    # BASELINE.json configs[1..3]: one training epoch on dataset-shaped synthetic graphs (exact V / E below)
    "cora-2p": (2, None, None, 1433, 16, 7, "optimize-gcn", 6),
    "citeseer-2p": (2, None, None, 3703, 16, 6, "optimize-gcn", 6),
    "pubmed-4p": (4, None, None, 500, 16, 3, "optimize-gcn", 6),
    # BASELINE.json configs[0]: the unoptimised kernel (original-gcn, 4 GAS iterations per epoch) on the Cora shape, 2 parties
    "cora-2p-original": (2, None, None, 1433, 16, 7, "original-gcn", 4),
}
DATASET_VE = {"cora-2p": (2708, 10556), "citeseer-2p": (3312, 10016), "pubmed-4p": (19717, 128146), "cora-2p-original": (2708, 10556)}


def kernel_source_hash():
    """sha256 over the kernel and engine sources the numbers depend on (cognn_amd/csrc, cognn_amd/host, include): stored with every
    PMC summary (tools/summarize_profiles.py) and every bench line, so a `traffic` figure taken from an older build is flagged."""
    import hashlib
    h = hashlib.sha256()
    for d in ("cognn_amd/csrc", "cognn_amd/host", "include"):
        for name in sorted(os.listdir(os.path.join(ROOT, d))):
            if name.endswith((".hip", ".h", ".hpp", ".cpp")):
                h.update(name.encode())
                h.update(open(os.path.join(ROOT, d, name), "rb").read())
    return h.hexdigest()[:16]


def message_widths(variant, iters, hid, lab, in_dim=0):
    if variant == "original-gcn":           # original-gcn/gcn.h:807-830: {in, hid, (lab: apply-only), hid}
        w = [in_dim, hid, 0, hid]
        return sum(w[i % 4] for i in range(iters))
    w = [hid, lab, 0, lab, 0, hid]          # getPlainNumPerOperand per GAS iteration (gcn.h:898-927); 0 = apply-only
    return sum(w[i % 6] for i in range(iters))


def _cpu_engine_run(wl, shift, threads, steps, exact=None):
    """One child process of oracle/cpu_engine_bench.py (plain-C++ reference backend under the same engine host code).
    exact = (V, directed E): that size instead of the workload's power-of-two size scaled down by 2^shift."""
    import subprocess
    k, lv, le, in_dim, hid, lab, variant, iters = wl
    if exact:
        lv2, le2 = "=%d" % exact[0], "=%d" % exact[1]
    else:
        lv2, le2 = max(lv - shift, 8), max(le - shift, 10)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_engine_bench.py"), str(k), str(lv2), str(le2), str(in_dim),
                          str(hid), str(lab), variant, str(iters), str(steps)], env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-2000:])
    r = json.loads(out.stdout.strip().splitlines()[-1])
    return r["seconds_per_pass"], float(r["edges"]), lv2, le2


def cpu_baseline(args, wl):
    """CPU restatement of the same pass, timed on this box's host cores on a bounded sample (rank 0, N=1 only).  Preferred:
    the engine's host code on the plain-C++ reference backend (oracle/libcognn_engine_cpu.so, OpenMP), all cores on a
    1/4-scale graph plus one single-thread pass on a 1/16-scale graph; fallback (library not built): the numpy oracle."""
    k, lv, le, in_dim, hid, lab, variant, iters = wl
    widths = message_widths(variant, iters, hid, lab, in_dim)
    try:
        cores = len(os.sched_getaffinity(0))
        try:                                                # cgroup v2 CPU quota, when one is set
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                cores = max(1, min(cores, int(int(q) / int(per))))
        except (OSError, ValueError):
            pass
        cores = min(cores, 16)                              # a one-GPU box's CPU share is 16 cores; more threads only oversubscribe
        if args.workload in DATASET_VE:                     # dataset-shaped epochs are small: the CPU runs the whole workload
            ve = DATASET_VE[args.workload]
            dt_all, edges_all, _, _ = _cpu_engine_run(wl, 0, cores, 5, exact=ve)
            dt_1, edges_1, _, _ = _cpu_engine_run(wl, 0, 1, 3, exact=ve)
            return {"value": edges_all * widths / dt_all, "unit": "edges*feat/s", "cores": cores, "kind": "port",
                    "value_1core": edges_1 * widths / dt_1, "seconds_per_epoch": dt_all, "seconds_per_epoch_1core": dt_1,
                    "sample": "engine host code on the plain-C++ reference backend (oracle/cpu_backend.cpp, OpenMP, per-side loops): the whole "
                              "workload (%d-party %s epoch, %d vertices / %d directed edges, in=%d hid=%d labels=%d): %.3f s per epoch on %d "
                              "threads, %.3f s on 1 (medians of 5 and 3 timed epochs after a warm-up epoch; dealer phase outside the timed region)"
                              % (k, variant, ve[0], ve[1], in_dim, hid, lab, dt_all, cores, dt_1)}
        # all-cores leg at the bench's own size when the host has the memory for it (the plain-C++ backend keeps every "device"
        # buffer in host memory: about 1.4 x the GPU engine's allocation + the graph), on a 1/4-scale graph otherwise
        shift, why = 2, None
        need_gb = 1.4 * (getattr(args, "engine_GB", 0.0) or 32.0) + 8.0
        try:
            avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1e6
        except (OSError, IndexError, ValueError):
            avail_gb = 0.0
        if os.environ.get("COGNN_BENCH_CPU_SCALED"):
            why = "COGNN_BENCH_CPU_SCALED is set"
        elif avail_gb < need_gb:
            why = "host has %.0f GB available, the full-size CPU run needs about %.0f GB" % (avail_gb, need_gb)
        else:
            shift = 0
        try:
            dt_all, edges_all, lva, lea = _cpu_engine_run(wl, shift, cores, 3 if shift == 0 else 5)      # median of the timed passes
        except Exception as ex:  # noqa: BLE001 - e.g. the full-size run was killed: fall back to the scaled sample
            if shift == 0:
                shift, why = 2, "the full-size CPU run failed (%s)" % (str(ex)[-120:],)
                dt_all, edges_all, lva, lea = _cpu_engine_run(wl, shift, cores, 5)
            else:
                raise
        dt_1, edges_1, lv1, le1 = _cpu_engine_run(wl, 4, 1, 3)              # median of 3 passes
        return {"value": edges_all * widths / dt_all, "unit": "edges*feat/s", "cores": cores, "kind": "port",
                "scale": 1.0 / (1 << shift), "scale_reason": why, "seconds_per_pass": dt_all,
                "value_1core": edges_1 * widths / dt_1, "scale_1core": 1.0 / 16,
                "sample": "engine host code on the plain-C++ reference backend (oracle/cpu_backend.cpp, OpenMP, per-side loops): %d-party %s pass, "
                          "in=%d hid=%d labels=%d; %d threads on a 2^%d-vertex/2^%d-edge graph: %.2f s per pass; 1 thread on "
                          "2^%d/2^%d: %.2f s per pass (medians of the timed passes after a warm-up pass; dealer phase outside the timed region)"
                          % (k, variant, in_dim, hid, lab, cores, lva, lea, dt_all, lv1, le1, dt_1)}
    except Exception as ex:  # noqa: BLE001 - the baseline must not take the bench down
        sys.stderr.write("cpu_baseline: C++ reference backend unavailable (%r), using the numpy oracle\n" % (ex,))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cognn_oracle as co
    if args.workload in DATASET_VE:
        return None                                         # (the numpy oracle at dataset size takes minutes per epoch)
    lv2, le2 = max(lv - 4, 8), max(le - 4, 10)
    V, Eu = 1 << lv2, 1 << (le2 - 1)
    src, dst = co.synth_graph(V, Eu, 0xC06A11)
    part = np.arange(V) % k
    feats, labels = co.synth_features(V, in_dim, lab, 0xC06A12)
    p = co.GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V)
    o = co.OracleEngine(k, src, dst, part, feats, labels, p, seed=0xC06A11, variant=variant)
    t0 = time.perf_counter()
    for it in range(iters):
        o.iteration(it)
    dt = time.perf_counter() - t0
    ef = float(len(src)) * widths
    return {"value": ef / dt, "unit": "edges*feat/s", "cores": 1, "kind": "port",
            "sample": "numpy oracle (oracle/cognn_oracle.py), %d-party %s pass on a 2^%d-vertex/2^%d-edge graph, "
                      "in=%d hid=%d labels=%d, %.2f s" % (k, variant, lv2, le2, in_dim, hid, lab, dt)}


def attach_transport(eng, torch, dist, cdist, backend, local_rank, per_round):
    """The exchange of a multi-rank run: the native RCCL transport (ncclSend / ncclRecv groups issued from C++), or - when ANY rank could
    not create it, agreed through the process group so that no rank is left behind - the torch.distributed callback transport on the same
    devices (batch_isend_irecv on the engine's buffers: what round 2 used); with the gloo rehearsal backend the host-staged callbacks.
    Returns (RcclExchange or None, description, error or None)."""
    device = torch.device("cuda", local_rank)
    if backend != "nccl":
        eng.set_exchange(cdist.make_exchange_async(device, host_staged=True, per_round=per_round))
        return None, "torch.distributed %s, host-staged (rehearsal transport)" % backend, None
    xch, err = None, None
    try:
        xch = cdist.attach_rccl(eng, local_rank)
    except Exception as ex:  # noqa: BLE001 - the fallback below still measures
        err = str(ex)[-300:]
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        return xch, "native RCCL p2p groups on a communication stream (csrc/exchange_rccl.hip)", None
    eng.set_exchange(cdist.make_exchange_async(device, host_staged=False, per_round=per_round))
    return None, "torch.distributed nccl p2p (FALLBACK: the native RCCL transport could not be created on some rank)", err or "failed on another rank"


def cross_path_check(eng, Engine, k, src, dst, part, param, variant, iters, in_dim, lab, passes, recorded=False):
    """After the timed region (N = 1): the shares the bench's own sequence left behind - forward-only stores, retained offline
    products, the pass replayed warmup + steps times - must equal, bit for bit and for every party, those of a fresh engine
    that runs the pass once the plain way; for an inference pass the revealed rows must also be probability vectors.  (Parity
    with the oracle at sizes the oracle can run - the same workload at 1/64 scale, every row - is tests/test_small_workload_gpu.py.)"""
    import hashlib

    def digest(e):
        h = hashlib.sha256()
        for P in range(k):
            for sd in (0, 1):
                h.update(np.ascontiguousarray(e.shares(P, sd)).tobytes())
                for layer in (0, 1):                        # (a training epoch ends with an empty vertex tensor: the weights carry the result)
                    h.update(np.ascontiguousarray(e.weight(P, sd, layer)).tobytes())
        return h.hexdigest()
    t0 = time.perf_counter()
    d_bench = digest(eng)
    ref = Engine(k, src, dst, part, param, seed=0xC06A11, variant=variant)
    for P in ref.hosted:
        vids = ref.party_vids(P)
        rng = np.random.default_rng(0xC06A12 + P)
        ref.set_party_data(P, (rng.random((len(vids), in_dim)) < 0.01).astype(np.float64), rng.integers(0, lab, size=len(vids)))
    ref.start()
    if recorded:                                            # (recorded epochs deal the feature operand's mask anew every epoch: the plain run does too;
        ref.graph_epochs(True)                              #  one GAS iteration per call never records anything)
    for _ in range(passes if "inference" not in variant else 1):   # a training pass updates the weights: as many passes as the bench ran
        for it in range(iters):                             # one call per GAS iteration: none of the paths that span iterations of a call
            ref.run(it, it + 1)
    d_plain = digest(ref)
    res = {"what": "sha256 over every party's two vertex-tensor shares and weight shares: the bench sequence (forward-only / retained products / "
                   "replays, whole passes per call) vs the plain sequence on a fresh engine, one call per GAS iteration",
           "cross_path_identical": d_bench == d_plain, "digest": d_bench}
    if "inference" in variant:
        a = ref.shares(0, 0); b = ref.shares(0, 1)
        with np.errstate(over="ignore"):
            rec = (a + b).astype(np.int64).astype(np.float64) / 65536.0      # p - y of party 0, Q16
        train = int(rec.shape[0] * param.train_ratio)
        rows = rec[: min(train, 4096)]
        # p = (p - y) + onehot(y): every row sums to 1 and lies in [0, 1] whatever the label
        s1 = rows.sum(axis=1) + 1.0
        res["party0_rows_checked"] = int(rows.shape[0])
        res["prob_rows_sum_to_one"] = bool(np.allclose(s1, 1.0, atol=2e-3))
        res["rows_past_train_set_zero"] = bool(np.all(rec[train:] == 0))
    ref.close()
    res["seconds"] = time.perf_counter() - t0
    return res


def multi_rank_check(eng, Engine, dist, rank, world, placement, k, src, dst, part, param, variant, iters, in_dim, lab, passes, device):
    """After the timed region (N > 1): every rank hashes the shares and weight shares it holds; rank 0 runs the same job as ONE process
    on its GPU (all parties, one call per GAS iteration, as many passes as the bench ran for a training workload) and compares share by
    share: the N-rank run (placement, transport, overlap) must end in exactly the single-process bits."""
    import hashlib
    m = k // world

    def holder(o, sd):
        return (o if (placement == "vertex-set" or sd == 0) else (o + 1) % k) // m

    def digest(e, o, sd):
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(e.shares(o, sd)).tobytes())
        for layer in (0, 1):
            h.update(np.ascontiguousarray(e.weight(o, sd, layer)).tobytes())
        return h.hexdigest()
    t0 = time.perf_counter()
    local = {"%d/%d" % (o, sd): digest(eng, o, sd) for o in range(k) for sd in (0, 1) if holder(o, sd) == rank}
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    res = None
    try:
        if rank == 0:
            res = _single_process_reference(Engine, gathered, digest, k, src, dst, part, param, variant, iters, in_dim, lab, passes, device, world, t0)
    except Exception as ex:  # noqa: BLE001 - rank 0 must still meet the others at the barrier
        res = {"skipped": "the single-process reference could not run: %s" % (str(ex)[-200:],)}
    finally:
        dist.barrier()
    return res


def _single_process_reference(Engine, gathered, digest, k, src, dst, part, param, variant, iters, in_dim, lab, passes, device, world, t0):
    got = {}
    for g in gathered:
        got.update(g)
    ref = Engine(k, src, dst, part, param, seed=0xC06A11, variant=variant, device=device)
    for P in ref.hosted:
        vids = ref.party_vids(P)
        rng = np.random.default_rng(0xC06A12 + P)
        ref.set_party_data(P, (rng.random((len(vids), in_dim)) < 0.01).astype(np.float64), rng.integers(0, lab, size=len(vids)))
    ref.start()
    for _ in range(passes if "inference" not in variant else 1):
        for it in range(iters):
            ref.run(it, it + 1)
    bad = [key for key in sorted(got) if got[key] != digest(ref, int(key.split("/")[0]), int(key.split("/")[1]))]
    ref.close()
    res = {"what": "sha256 of every (party, share) tensor and its weight shares as the %d ranks hold them after the bench's sequence vs a "
                   "single-process engine on rank 0's GPU running the plain sequence, one call per GAS iteration" % world,
           "cross_path_identical": len(got) == 2 * k and not bad, "shares_compared": len(got), "mismatches": bad[:8],
           "seconds": time.perf_counter() - t0}
    return res


def placement_leg(torch, dist, Engine, args, placement, backend, rank, world, local_rank, k, src, dst, part, param, variant, iters, in_dim, lab, n_warm, value_of):
    """N > 1: the same job on a second engine under the other placement (same transport, same timed loop: barrier + synchronize on both
    sides, max over ranks), with its own N-rank check.  Extra keys only."""
    from cognn_amd import dist as cdist
    eng = Engine(k, src, dst, part, param, seed=0xC06A11, variant=variant, rank=rank, world=world, device=local_rank, placement=placement)
    xch, _, _ = attach_transport(eng, torch, dist, cdist, backend, local_rank, False)
    for P in eng.hosted:
        vids = eng.party_vids(P)
        rng = np.random.default_rng(0xC06A12 + P)
        eng.set_party_data(P, (rng.random((len(vids), in_dim)) < 0.01).astype(np.float64), rng.integers(0, lab, size=len(vids)))
    eng.start()
    eng.retain_offline(True)
    if "inference" in variant:
        eng.forward_only(True)
    eng.offline(0, iters)

    def barrier():
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    for _ in range(n_warm):
        eng.run(0, iters)
    x0 = xch.stats() if xch else None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run(0, iters)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    x1 = xch.stats() if xch else None
    out = {"placement": placement, "ms_per_step": dt / args.steps * 1e3, "value": value_of(dt / args.steps), "unit": "edges*feat/s"}
    if xch:
        out["exchange"] = {"rounds_per_step": (x1["rounds"] - x0["rounds"]) / args.steps, "MB_sent_per_step": (x1["bytes_sent"] - x0["bytes_sent"]) / args.steps / 1e6,
                           "comm_stream_ms_per_step": (x1["comm_ms"] - x0["comm_ms"]) / args.steps}
    if not args.no_check:
        try:
            out["check"] = multi_rank_check(eng, Engine, dist, rank, world, placement, k, src, dst, part, param, variant, iters, in_dim, lab,
                                            n_warm + 2 * args.steps, local_rank)
        except Exception as ex:  # noqa: BLE001
            out["check"] = {"skipped": "the verification could not run: %s" % (str(ex)[-200:],)}
    eng.close()
    if xch:
        xch.close()
    return out


def dealer_streams_leg(eng, torch, args, iters, k, digest_before, minimal=False):
    """The same pass with the dealer values of the co-located pairs' chains and the A masks of the grouped products READ from
    HBM (COGNN_OPT_DEALER_STREAMS) instead of regenerated in registers: what the online phase costs when the offline phase hands
    each party its correlations in memory, as the reference's does (README.md:215-216).  Extra keys only: the headline stays the
    in-register form.  The first passes deal (materialise) the values; the timed ones read them."""
    import hashlib
    mem0 = eng.memory()[1]
    if minimal:
        eng.dealer_minimal(True)                  # (reads the slabs the streamed leg materialised, when that leg ran first)
    else:
        eng.dealer_streams(True)
    for _ in range(2):
        eng.run(0, iters)
    eng.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run(0, iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_agg, ms_agg, bytes_agg = eng.timing(0)
    n_gemm, ms_gemm, ops_gemm = eng.timing(2)
    n_gepi, ms_gepi, ops_gepi = eng.timing(3)
    eng.enable_timing(False)
    res = {"what": ("the dealer-minimal form: every party regenerates what it derives from its own seed (a_p, b_p, c_0, C_0, r_0, r'_0, opening and product "
                    "masks) and READS only what a PRG-compressed dealer must send - party 1's corrections c_1, r_1, r'_1 and the ReLU's published g "
                    "(7 of the 22 per-element slots of the pair chains; the products' C_1 as in every form); same shares") if minimal else
                   ("dealer values of the pair chains (truncations, row scales, ReLUs, openings: what each party receives) and the A masks of the "
                    "grouped products read from HBM instead of regenerated from the counter PRNG; same shares"),
           "ms_per_step": dt / args.steps * 1e3, "dealt_GB_resident": (eng.memory()[1] - mem0) / 1e9,
           "gather_avg_ms": ms_agg / max(n_agg, 1), "gather_bytes_per_launch_incl_dealt": bytes_agg / max(n_agg, 1),
           "gather_GBps_incl_dealt": (bytes_agg / 1e9) / (ms_agg / 1e3) if ms_agg > 0 else None,
           "beaver_gemm_avg_ms_per_phase": ms_gemm / max(n_gemm, 1)}
    if digest_before is not None:
        h = hashlib.sha256()
        for P in range(k):
            for sd in (0, 1):
                h.update(np.ascontiguousarray(eng.shares(P, sd)).tobytes())
                for layer in (0, 1):
                    h.update(np.ascontiguousarray(eng.weight(P, sd, layer)).tobytes())
        res["shares_identical_to_in_register_form"] = (h.hexdigest() == digest_before) if "inference" in WORKLOADS[args.workload][6] else None
    eng.dealer_streams(False)
    return res


def self_launch(args):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE in the environment): this process becomes the launcher - it has not
    imported torch or touched a GPU - and starts the N ranks as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), relays
    rank 0's output, and exits non-zero if any rank does.  No exec, no interpreter state shared with the ranks."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), COGNN_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else 2))        # rank 0 prints the JSON line; whatever the others print goes to stderr
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (procs.index(p), code))
                    for q in alive:                          # (exactly the processes started above)
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config5", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the cross-path verification after the timed region")
    ap.add_argument("--no-dealer-streams", action="store_true", help="skip the extra timing of the dealt (memory-streamed dealer) form")
    ap.add_argument("--chunks", type=int, default=1, help="N > 1: row chunks of the cross-rank open -> exchange -> close steps (COGNN_OPT_EXCHANGE_CHUNKS)")
    ap.add_argument("--placement", default="party", choices=["party", "vertex-set"],
                    help="N > 1, the placement `value` is measured on: party = a GPU is a set of parties as BASELINE.json's north_star has it (one party per "
                         "GPU at N = 8; every Beaver / truncation opening of an owner / co-party pair crosses xGMI over RCCL p2p); vertex-set = every GPU "
                         "holds BOTH shares of its parties' vertex sets (the co-located mode of N = 1 extended: two-party steps stay in registers, only the "
                         "Gather's share-table replicas and the weight average cross xGMI).  With the default the other one is measured too and reported "
                         "under `vertex_set_placement`")
    ap.add_argument("--packed", action="store_true", help="N > 1: opened truncation / ReLU-product shares cross ranks as 6 bytes per element (COGNN_OPT_PACKED_OPENINGS)")
    ap.add_argument("--no-placement-leg", action="store_true", help="N > 1: skip the extra measurement of the vertex-set placement")
    ap.add_argument("--graph", action="store_true", help="training workloads: replay the recorded epoch (hipGraph, COGNN_OPT_GRAPH_EPOCHS) instead of launching every kernel")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:                        # started bare: be the launcher (before torch is imported or a GPU is touched)
        self_launch(args)
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    if world > 1:
        # a rank that waits for a peer forever (a transport that never completes) would leave no trace: after this many seconds every
        # rank dumps its Python stacks to stderr and exits non-zero (COGNN_BENCH_WATCHDOG_S; the N = 8 run takes about two minutes)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get("COGNN_BENCH_WATCHDOG_S", "1500")), exit=True)
    # COGNN_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share the devices round
    # robin, messages are staged through host memory); the real multi-GPU run uses RCCL ("nccl")
    backend = os.environ.get("COGNN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from cognn_amd.engine import Engine, GnnParam
    wl = WORKLOADS[args.workload]
    k, lv, le, in_dim, hid, lab, variant, iters = wl
    if k % world != 0:
        raise SystemExit("the %d parties must divide evenly over %d GPUs" % (k, world))
    if args.workload in DATASET_VE:
        V, Eu = DATASET_VE[args.workload][0], DATASET_VE[args.workload][1] // 2
        lv, le = int(np.log2(V)), int(np.log2(2 * Eu))
    else:
        V, Eu = 1 << lv, 1 << (le - 1)
    t_setup = time.perf_counter()
    src, dst = synth_graph(V, Eu, 0xC06A11)
    part = (np.arange(V) % k).astype(np.int32)
    param = GnnParam(num_labels=lab, input_dim=in_dim, hidden_dim=hid, num_samples=V, num_edges=len(src))
    eng = Engine(k, src, dst, part, param, seed=0xC06A11, variant=variant, rank=rank, world=world, device=local_rank, placement=args.placement)
    xch = None
    if world > 1:
        from cognn_amd import dist as cdist
        xch, transport_desc, transport_err = attach_transport(eng, torch, dist, cdist, backend, local_rank, args.chunks > 1)
        if args.chunks > 1:
            eng.exchange_chunks(args.chunks)
        if args.packed:
            eng.packed_openings(True)
    for P in eng.hosted:                      # Bernoulli(0.01) bag-of-words features, uniform labels
        vids = eng.party_vids(P)
        rng = np.random.default_rng(0xC06A12 + P)
        eng.set_party_data(P, (rng.random((len(vids), in_dim)) < 0.01).astype(np.float64), rng.integers(0, lab, size=len(vids)))
    eng.start()
    t_off = time.perf_counter()
    eng.retain_offline(True)                  # every step replays the same iterations: keep their dealt product shares
    if "inference" in variant:
        eng.forward_only(True)                # -m 2: the hidden activation and the ReLU sign mask have no reader
    # measured (MI355X, ROCm 7.2): replaying a recorded epoch is NOT faster than launching its kernels - Cora 0.43 vs 0.45 ms, CiteSeer 0.64 vs
    # 0.54, PubMed 0.80 vs 0.69, config5-train 15.1 vs 13.0 - a serial chain of dependent dispatches costs the same either way; opt-in only
    recorded = world == 1 and variant == "optimize-gcn" and iters % 6 == 0 and args.graph
    if recorded:
        eng.graph_epochs(True)                # a training epoch per step: recorded once (hipGraph), replayed (COGNN_OPT_GRAPH_EPOCHS)
    eng.offline(0, iters)
    torch.cuda.synchronize()
    offline_ms = (time.perf_counter() - t_off) * 1e3
    setup_s = time.perf_counter() - t_setup

    def barrier():
        # drain this rank's own work first: the engine's p2p groups run on their own communicator, and a torch collective
        # enqueued while they are still in flight would put two communicators' kernels on the device in rank-dependent order
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    n_warm = max(args.warmup, 2) if recorded else args.warmup   # (recorded epochs: one eager, one while recording)
    if not recorded:
        eng.enable_timing(True)               # (the warm-up creates the timing events the timed region reuses: creating one costs as much as a small kernel)
    for _ in range(n_warm):
        eng.run(0, iters)
    if not recorded:
        eng.enable_timing(True)               # restart the timers (per-kernel HIP-event timers would break up a recorded epoch)
    x0 = xch.stats() if xch else None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run(0, iters)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    x1 = xch.stats() if xch else None
    n_agg, ms_agg, bytes_agg = eng.timing(0)              # aggregate launches of the hidden-wide rounds (the dominant kernel)
    n_lab, ms_lab, bytes_lab = eng.timing(4)              # ... of the label-wide rounds (a different kernel in the fused form)
    n_part, ms_part, bytes_part = eng.timing(1)
    n_gemm, ms_gemm, ops_gemm = eng.timing(2)
    n_gepi, ms_gepi, ops_gepi = eng.timing(3)
    eng.enable_timing(False)
    # the same K steps once more without the per-kernel HIP events the roofline needs (reported beside the headline, never as `value`):
    # an epoch of a dataset-sized graph is a chain of ~20 launches of 4-25 us, and two event records per timed kernel are a visible
    # share of it; at the benchmark size they are not
    barrier()
    t0u = time.perf_counter()
    for _ in range(args.steps):
        eng.run(0, iters)
    barrier()
    dtu = time.perf_counter() - t0u
    if world > 1:
        t = torch.tensor([dtu], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtu = float(t.item())
    ranks_seen = None
    if xch:                                   # what the communicator says: ncclCommCount, this rank, an all-reduce of ones over it
        cc, cr, ones = xch.ranks()
        ranks_seen = {"ranks": cc, "comm_rank": cr, "allreduce_of_ones": ones}
    elif world > 1:
        t1 = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t1)
        ranks_seen = {"ranks": dist.get_world_size(), "comm_rank": dist.get_rank(), "allreduce_of_ones": int(t1.item())}

    # HBM traffic per kernel: NOT measured by this run - read from the newest committed PMC summary (profiles/rNN_pmc.json: the
    # builder's separate rocprofv3 --pmc passes of this same command, tools/collect_profiles.sh, reduced by
    # tools/summarize_profiles.py) and labelled so; `traffic_stale` says whether the kernel sources have changed since.
    all_pairs_local = world == 1 or args.placement == "vertex-set"
    fused = all_pairs_local and k // world <= 8            # the Gather with the pair chain as its epilogue (engine.cpp can_fuse_gather_chain)
    pmc, pmc_file = {}, None
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json") and f[:1] == "r" and f[1:3].isdigit()) \
        if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    if world == 1 and cands:
        pmc_file = cands[-1]
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
    pmc_build = pmc.get("build", {})
    src_hash = kernel_source_hash()

    def lanes(F):                                          # lanes per row of the gather kernels (kernels_gather.hip pick_lpr; 16-byte lanes for even F)
        n, l = (F // 2 if F % 2 == 0 else F), 1
        while l < n and l < 64:
            l <<= 1
        return l

    def agg_entry(n, ms, nbytes, F, label_round):
        if n <= 0 or ms <= 0:
            return None
        if variant == "original-gcn":
            kname = "scatter_gather_original_kernel"
        elif not fused:
            kname = "gather_csr_kernel<%d," % lanes(F)
        elif label_round and "inference" in variant and not os.environ.get("COGNN_NO_SOFTMAX_FUSION"):
            kname = "gather_pair_softmax_kernel<%d," % lanes(F)
        else:                                              # (a training epoch's label-wide rounds: one of each kernel, timed together)
            kname = "gather_pair_chain_kernel<%d," % lanes(F)
        e = {"kernel": kname.rstrip(","), "row_width_F": F, "launches": n, "avg_ms": ms / n, "algo_bytes_per_launch": nbytes / n,
             "achieved": (nbytes / 1e9) / (ms / 1e3), "peak": 8000.0, "unit": "GB/s", "frac": (nbytes / 1e9) / (ms / 1e3) / 8000.0,
             "traffic": None}
        hits = [v for name, v in pmc.get(args.workload, {}).items() if isinstance(v, dict) and name.startswith(kname) and "hbm_bytes_per_launch" in v]
        if len(hits) == 1:
            e["traffic"] = hits[0]["hbm_bytes_per_launch"]
            e["traffic_source"] = ("profiles/%s - the builder's PMC pass on build %s (git %s), not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                   "in separate passes, FETCH x2 gfx950 correction" % (pmc_file, pmc_build.get("tag", "?"), pmc_build.get("git_head", "?")))
            e["traffic_stale"] = pmc_build.get("kernel_source_hash") != src_hash     # kernel sources changed since the PMC pass (or the pass predates the hash)
        return e
    per_kernel = [e for e in (agg_entry(n_agg, ms_agg, bytes_agg, in_dim if variant == "original-gcn" else hid, False),
                              agg_entry(n_lab, ms_lab, bytes_lab, lab, True)) if e]
    dominant = max(per_kernel, key=lambda e: e["avg_ms"] * e["launches"]) if per_kernel else None
    ms_per_step = dt / args.steps * 1e3
    ef_per_step = float(len(src)) * message_widths(variant, iters, hid, lab, in_dim)
    value = ef_per_step / (dt / args.steps)
    wlinfo = eng.workload()
    out = {
        "metric": "secret-shared GCN epoch time (s) + edges*feat/s per party",
        "value": value, "unit": "edges*feat/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%d-party %s pass (GAS iterations 0-%d) on a synthetic %d-vertex/%d-edge global graph, "
                               "partition vid %% %d, input_dim=%d hidden_dim=%d num_labels=%d, %d part%s per GPU"
                               % (k, variant, iters - 1, V, 2 * Eu, k, in_dim, hid, lab, k // world, "y" if k // world == 1 else "ies"),
                   "name": args.workload, "parties": k, "exchange": "in-device" if world == 1 else ("rccl-p2p" if xch else "torch-nccl-p2p (fallback)" if backend == "nccl" else backend + "-host-staged"),
                   "exchange_chunks": args.chunks if world > 1 else None, "packed_openings": bool(args.packed) if world > 1 else None,
                   "placement": (args.placement if world > 1 else "all parties and both share-holders of every vertex set on the one GPU")},
        "epoch_time_s": dt / args.steps,
        "ms_per_step_without_kernel_timers": dtu / args.steps * 1e3,
        "edges_feat_per_s_per_party": value / k,
        # offline_ms: the dealer (offline) phase of one step's Beaver products in steady state (measured after the timed region on steps
        # nobody has dealt for yet); offline_first_call_ms: the first such call of the process (one-time costs included: first launches
        # of the dealer kernels, launch lanes, buffer pools)
        "offline_ms": None, "offline_first_call_ms": offline_ms, "setup_s": setup_s,
        "device_GB_allocated_by_the_engine": eng.memory()[1] / 1e9,
        # the dominant kernel (largest share of the step) with its own launches / average duration / algorithmic bytes, and every
        # aggregate kernel of the step under per_kernel: each row can be recomputed from one line of profiles/*_kernel_stats.csv
        "roofline": dict({"bound": "hbm"}, **(dominant or {"kernel": None, "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None}),
                         traffic_unit="bytes per launch", per_kernel=per_kernel,
                         note=("scatter_gather_original_kernel is bound by its per-edge dealer arithmetic, not by HBM" if variant == "original-gcn" else
                               "algorithmic bytes = SURVEY.md 8(d): 8F(E + R) read + 4E + 4(R + 1) index bytes + what the launch writes; no cache credit")),
        "kernels": {"gather_partials": {"launches": n_part, "avg_ms": ms_part / max(n_part, 1),
                                        "GBps": (bytes_part / 1e9) / (ms_part / 1e3) if ms_part > 0 else None},
                    # one timed phase = the products of all hosted sides in one GAS iteration (they overlap on two launch lanes)
                    "beaver_gemm_close": {"phases": n_gemm, "avg_ms_per_phase": ms_gemm / max(n_gemm, 1),
                                          "i8_TOPs": (ops_gemm / 1e12) / (ms_gemm / 1e3) if ms_gemm > 0 else None,
                                          "frac_of_5000_TOPs": ((ops_gemm / 1e12) / (ms_gemm / 1e3) / 5000.0) if ms_gemm > 0 else None,
                                          "what": "the launches that are Beaver products and nothing else"},
                    # products of the co-located pairs' p = 1 sides with the pair's truncation chain as their epilogue (cognn_gemm_job::epilogue,
                    # N > 16): the launch also does the chain's work (reads the p = 0 product and C_1, writes both outputs) - its operation
                    # count is the product's alone, so this rate is NOT comparable with the pure products' above
                    "beaver_gemm_close_with_chain_epilogue": {"launches": n_gepi, "avg_ms": ms_gepi / max(n_gepi, 1),
                                                              "i8_TOPs": (ops_gepi / 1e12) / (ms_gepi / 1e3) if ms_gepi > 0 else None,
                                                              "frac_of_5000_TOPs": ((ops_gepi / 1e12) / (ms_gepi / 1e3) / 5000.0) if ms_gepi > 0 else None}},
        "graph": wlinfo,
        "launch": "recorded epoch replayed (hipGraph)" if recorded else "one launch per kernel",
        # every COGNN_* variable set in this process's environment (A/B switches of the engine and the kernels): {} = all defaults
        "switches": {key: val for key, val in sorted(os.environ.items()) if key.startswith("COGNN_") and key != "COGNN_BENCH_LAUNCHED"},
        "kernel_source_hash": src_hash,
    }
    if xch:                                   # rank 0's share of the exchange: rounds and bytes per step (all ranks are symmetric up to the partition)
        out["exchange"] = {"ranks": ranks_seen["ranks"], "comm_rank": ranks_seen["comm_rank"], "allreduce_of_ones": ranks_seen["allreduce_of_ones"],
                           "rounds_per_step": (x1["rounds"] - x0["rounds"]) / args.steps,
                           "MB_sent_per_step": (x1["bytes_sent"] - x0["bytes_sent"]) / args.steps / 1e6,
                           "MB_received_per_step": (x1["bytes_received"] - x0["bytes_received"]) / args.steps / 1e6,
                           "comm_stream_ms_per_step": (x1["comm_ms"] - x0["comm_ms"]) / args.steps,
                           "GBps_while_communicating": ((x1["bytes_sent"] - x0["bytes_sent"]) / 1e9) / max((x1["comm_ms"] - x0["comm_ms"]) / 1e3, 1e-12),
                           "transport": "native RCCL p2p groups on a communication stream (csrc/exchange_rccl.hip)"}
    if world > 1 and not xch:
        out["exchange"] = dict(ranks_seen, transport=transport_desc)
        if transport_err:
            out["exchange"]["native_transport_error"] = transport_err
    if not args.no_check and world == 1:
        try:
            out["check"] = cross_path_check(eng, Engine, k, src, dst, part, param, variant, iters, in_dim, lab, n_warm + 2 * args.steps, recorded=recorded)   # (warm-up, the timed steps, the same steps without kernel timers)
        except Exception as ex:  # noqa: BLE001 - e.g. a second engine of an 8x workload does not fit beside the first: the measurement stands
            out["check"] = {"skipped": "the verification engine could not run: %s" % (str(ex)[-200:],)}
    if not args.no_check and world > 1:
        try:
            chk = multi_rank_check(eng, Engine, dist, rank, world, args.placement, k, src, dst, part, param, variant, iters, in_dim, lab,
                                   n_warm + 2 * args.steps, local_rank)
        except Exception as ex:  # noqa: BLE001 - the measurement stands
            chk = {"skipped": "the verification could not run: %s" % (str(ex)[-200:],)}
        out["check"] = chk
    if world > 1 and args.placement == "party" and not args.no_placement_leg:
        eng.close()                               # (the headline engine and its communicator are done: free them before the second one starts)
        if xch:
            xch.close()
        eng, xch = None, None
        try:
            out["vertex_set_placement"] = placement_leg(torch, dist, Engine, args, "vertex-set", backend, rank, world, local_rank, k, src, dst, part, param,
                                                        variant, iters, in_dim, lab, n_warm, lambda sec: ef_per_step / sec)
        except Exception as ex:  # noqa: BLE001 - the measurement of the headline placement stands
            out["vertex_set_placement"] = {"skipped": "could not run: %s" % (str(ex)[-200:],)}
    if world == 1 and not args.no_dealer_streams and not recorded:
        # three forms of the same online phase side by side: masks regenerated in registers (the headline), every dealt value streamed
        # from HBM, and the dealer-minimal form in between (only what a PRG-compressed dealer must send is read)
        for key, minimal in (("dealer_streams", False), ("dealer_minimal", True)):
            try:
                out[key] = dealer_streams_leg(eng, torch, args, iters, k, out.get("check", {}).get("digest"), minimal=minimal)
            except Exception as ex:  # noqa: BLE001 - the dealt values of a large workload may not fit
                out[key] = {"skipped": "the dealt form could not run: %s" % (str(ex)[-200:],)}
    if not recorded and eng is not None and variant != "original-gcn":
        # What a real multi-epoch run pays per step: the dealer (offline) phase of the step's Beaver products + the online step (the
        # timed region above replays step 0 with its product shares retained).  Later steps, same barriers; training: dealt product
        # shares recycled after use.  The reference reports its preprocess phase separately too (README.md:236-237): both are kept.
        try:
            ep_len = 6                                      # GAS iterations per epoch of the optimize-gcn variants: step j deals [6 j, 6 j + iters)
            base = n_warm + args.steps + 4
            train = "inference" not in variant
            if train:
                eng.retain_offline(False)
                for e in range(2):
                    eng.offline((base + e) * ep_len, (base + e) * ep_len + iters); eng.run((base + e) * ep_len, (base + e) * ep_len + iters)
                barrier()
                t0 = time.perf_counter()
                for e in range(2, 2 + args.steps):         # (no synchronisation inside: the dealer launches queue behind the previous epoch)
                    eng.offline((base + e) * ep_len, (base + e) * ep_len + iters)
                    eng.run((base + e) * ep_len, (base + e) * ep_len + iters)
                barrier()
                dt2 = time.perf_counter() - t0
            base += 2 + args.steps
            eng.offline(base * ep_len, base * ep_len + iters)      # (warm: the dealer kernels have run once)
            eng.offline_discard(base * ep_len, base * ep_len + iters)
            barrier()
            t1 = time.perf_counter()                       # the dealer phase of further steps alone: their shares are handed back unconsumed
            for e in range(1, 1 + args.steps):                # (host bookkeeping only), so every step deals into the buffers of the one before
                eng.offline((base + e) * ep_len, (base + e) * ep_len + iters)
                eng.offline_discard((base + e) * ep_len, (base + e) * ep_len + iters)
            barrier()
            t_off_total = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([dt2 if train else 0.0, t_off_total], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt2, t_off_total = float(t[0].item()), float(t[1].item())
            out["offline_ms"] = t_off_total / args.steps * 1e3
            if train:
                out["epoch_time_incl_offline_s"] = dt2 / args.steps
                out["value_incl_offline"] = ef_per_step / (dt2 / args.steps)
        except Exception as ex:  # noqa: BLE001 - the headline stands
            out["epoch_time_incl_offline_note"] = "could not run: %s" % (str(ex)[-200:],)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        args.engine_GB = out["device_GB_allocated_by_the_engine"]
        if eng is not None:                                # the CPU leg runs in child processes: the GPU engine is done
            eng.close(); eng = None
        out["cpu_baseline"] = cpu_baseline(args, wl)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import faulthandler
        faulthandler.cancel_dump_traceback_later()
    if eng is not None:
        eng.close()
    if xch:
        xch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
