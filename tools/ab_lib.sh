#!/bin/bash
# Runs ON THE GPU BOX: A/B of two builds of the library (cognn_amd/libcognn_hip.so against cognn_amd/libcognn_hip_base.so) on bench
# workloads, alternating on the one box.   usage: tools/ab_lib.sh <tag> <workload> [...]
set -e
TAG="$1"; shift
cp cognn_amd/libcognn_hip.so /tmp/new.so; cp cognn_amd/libcognn_hip_base.so /tmp/base.so
for rep in 1 2; do
  for w in "$@"; do
    cp /tmp/base.so cognn_amd/libcognn_hip.so; echo -n "rep $rep base: "; bash tools/bench_brief.sh ${TAG}_a$rep $w
    cp /tmp/new.so cognn_amd/libcognn_hip.so; echo -n "rep $rep new:  "; bash tools/bench_brief.sh ${TAG}_b$rep $w
  done
done
python - "$TAG" "$@" <<'PY'
import json, sys
tag = sys.argv[1]
for w in sys.argv[2:]:
    for arm, name in (("a", "base"), ("b", "new")):
        v = []
        for rep in (1, 2):
            d = json.loads(open("gpurun_out/%s_%s%d/%s.json" % (tag, arm, rep, w)).read().strip().splitlines()[-1])
            v.append((round(d["kernels"]["beaver_gemm_close"]["avg_ms_per_phase"], 4), round(d["kernels"]["beaver_gemm_close"]["frac_of_5000_TOPs"], 4)))
        print(w, name, "product phases avg ms / frac of peak:", v)
PY
