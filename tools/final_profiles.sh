#!/bin/bash
# Runs ON THE GPU BOX: the whole profile set of a round in one go - rocprofv3 passes of every workload (tools/collect_profiles.sh),
# their reduction to profiles/<prefix>_* (tools/summarize_profiles.py), the bench lines once more against the PMC file just written
# (tools/refresh_bench_lines.sh), the reduction again - and a copy of the result under gpurun_out/<tag>/profiles/ for the way back.
#   usage: COGNN_GIT_HEAD=<short hash> tools/final_profiles.sh <tag> <prefix> [workload ...]
set -e
TAG="$1"; PREFIX="$2"; shift 2
WLS="${@:-config5 config5-h16 config5-train cora-2p citeseer-2p pubmed-4p cora-2p-original}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
for w in $WLS; do
  bash tools/collect_profiles.sh "$TAG" "$w" > "gpurun_out/collect_${TAG}_$w.log" 2>&1
  echo "collected $w"
done
python3 tools/summarize_profiles.py "gpurun_out/$TAG" "$PREFIX" $WLS
bash tools/refresh_bench_lines.sh "$TAG" $WLS
python3 tools/summarize_profiles.py "gpurun_out/$TAG" "$PREFIX" $WLS
mkdir -p "gpurun_out/$TAG/profiles"
cp profiles/${PREFIX}_* "gpurun_out/$TAG/profiles/"
echo "profile set $PREFIX written (tag $TAG)"
