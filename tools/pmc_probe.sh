#!/bin/bash
# Runs ON THE GPU BOX: PMC counters of tools/group_probe.py (one product shape) - where the waves of the grouped Beaver product spend their cycles.
#   usage: tools/pmc_probe.sh <outdir> M K N pairs
set -e
OUT="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$OUT"
OUT=$(readlink -f "$OUT")
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/$tag" -- python3 "$ROOT/tools/group_probe.py" "$@" > "$OUT/$tag.log" 2>&1 || echo "group $tag failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for path in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "beaver_gemm_group" not in k: continue
        k = k[k.find("beaver_gemm_group"):][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-28s %.4g per launch" % (c, v / max(1, len(n[(k, c)]))))
PY
