#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/<tag>/<workload>/) into the small summaries committed
under profiles/:  <prefix>_<workload>_kernel_stats.csv (per-kernel calls / average / share, names shortened),
<prefix>_<workload>_bench.json (the bench line of the same build) and one <prefix>_pmc.json with, per workload and kernel,
HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE: gfx950 reports half the bytes of 16-B/lane coalesced reads, counters are
in KB - MI355X_MICROARCH.md, HBM section) and the MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) /
(GRBM_GUI_ACTIVE / 8 XCDs).   usage: tools/summarize_profiles.py <gpurun_out/tag> <prefix> [workload ...]"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def counters(path_glob):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for path in glob.glob(path_glob):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in launches.items()}


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    workloads = sys.argv[3:] or sorted(os.listdir(src))
    import subprocess
    try:
        head = os.environ.get("COGNN_GIT_HEAD") or subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ""
    # "build": what the passes were collected on - the tag of the gpurun directory and the commit checked out when the raw
    # outputs were reduced (the builder's PMC pass, not a measurement of whoever reads the number later)
    sys.path.insert(0, ROOT)
    import bench                                            # (kernel_source_hash: the same digest every bench line carries)
    pmc = {"method": __doc__.split("usage")[0].strip().split("\n", 3)[-1].strip(),
           "build": {"tag": os.path.basename(os.path.normpath(src)), "git_head": head, "kernel_source_hash": bench.kernel_source_hash()}}
    for wl in workloads:
        d = os.path.join(src, wl)
        if not os.path.isdir(d):
            continue
        out = {}
        stats = glob.glob(os.path.join(d, "stats", "*", "*_kernel_stats.csv"))
        if stats:
            rows = list(csv.DictReader(open(stats[0])))
            total = sum(int(r["TotalDurationNs"]) for r in rows)
            with open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (prefix, wl)), "w") as f:
                f.write("kernel,calls,total_us,avg_us,percent\n")
                for r in rows:
                    f.write("%s,%s,%.1f,%.2f,%.2f\n" % (short(r["Name"]).replace(",", ";"), r["Calls"], int(r["TotalDurationNs"]) / 1e3,
                                                       float(r["AverageNs"]) / 1e3, 100.0 * int(r["TotalDurationNs"]) / total))
        bench = os.path.join(d, "bench.json")
        if os.path.exists(bench):
            line = open(bench).read().strip().splitlines()[-1]
            json.dump(json.loads(line), open(os.path.join(ROOT, "profiles", "%s_%s_bench.json" % (prefix, wl)), "w"), indent=1)
        fetch, nf = counters(os.path.join(d, "pmc_fetch", "*", "*_counter_collection.csv"))
        write, nw = counters(os.path.join(d, "pmc_write", "*", "*_counter_collection.csv"))
        mfma, nm = counters(os.path.join(d, "pmc_mfma", "*", "*_counter_collection.csv"))
        for k in sorted(set(fetch) | set(mfma)):
            if not any(t in k for t in ("gather", "gemm", "pair_chain", "softmax")):
                continue
            e = {}
            if k in fetch and k in write and nf.get(k) and nw.get(k):
                fk = fetch[k]["FETCH_SIZE"] / nf[k]; wk = write[k]["WRITE_SIZE"] / nw[k]
                e.update(launches_profiled=nf[k], FETCH_SIZE_KB_per_launch=fk, WRITE_SIZE_KB_per_launch=wk,
                         hbm_bytes_per_launch=(2.0 * fk + wk) * 1024.0)
            if k in mfma and mfma[k].get("GRBM_GUI_ACTIVE"):
                e.update(mfma_pipe_utilisation=(mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (mfma[k]["GRBM_GUI_ACTIVE"] / 8.0),
                         gui_cycles_per_launch=mfma[k]["GRBM_GUI_ACTIVE"] / 8.0 / nm[k])
            out[k] = e
        # what bench.py reports as roofline.traffic: bytes per aggregate gather launch, averaged over the launches of a step
        g = [(v["hbm_bytes_per_launch"], v["launches_profiled"]) for k, v in out.items()
             if k.startswith(("gather_csr_kernel", "gather_pair_chain_kernel", "gather_pair_softmax_kernel")) and "hbm_bytes_per_launch" in v]
        if g:
            out["aggregate_launch_avg_bytes"] = sum(b * n for b, n in g) / sum(n for _, n in g)
        pmc[wl] = out
    json.dump(pmc, open(os.path.join(ROOT, "profiles", "%s_pmc.json" % prefix), "w"), indent=1)
    print("wrote profiles/%s_*" % prefix)


if __name__ == "__main__":
    main()
