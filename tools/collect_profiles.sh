#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 summaries of bench.py for one workload into gpurun_out/<tag>/.
#   usage: tools/collect_profiles.sh <tag> <workload>
# Four separate profiler runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (kernel-trace + stats; then one --pmc pass
# per counter group with --kernel-trace only): kernel stats, FETCH_SIZE, WRITE_SIZE, MFMA busy.  tools/summarize_profiles.py
# turns the outputs into the small files committed under profiles/.
set -e
TAG="$1"; WL="${2:-config5}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/$TAG/$WL"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --workload "$WL" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --workload "$WL" --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-dealer-streams > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-dealer-streams > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-dealer-streams > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-dealer-streams > "$OUT/pmc_mfma.log" 2>&1
echo "profiles of $WL collected in $OUT"
