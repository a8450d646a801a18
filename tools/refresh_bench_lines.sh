#!/bin/bash
# Runs ON THE GPU BOX, after tools/summarize_profiles.py has written profiles/<prefix>_pmc.json from the same sources: the bench line of
# every collected workload once more, so that the committed copy carries the PMC traffic of THIS build (a bench line reads the
# committed PMC file and marks it stale when its kernel-source hash differs).   usage: tools/refresh_bench_lines.sh <tag> <workload> [...]
set -e
TAG="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for WL in "$@"; do
  OUT="$ROOT/gpurun_out/$TAG/$WL"
  mkdir -p "$OUT"
  python3 "$ROOT/bench.py" --workload "$WL" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
done
