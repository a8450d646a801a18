#!/bin/bash
# Runs ON THE GPU BOX: bench.py for the given workloads, one summary line each (online step, steady-state offline phase, both together).
#   usage: tools/bench_brief.sh <tag> <workload> [...]
set -e
TAG="$1"; shift
mkdir -p gpurun_out/$TAG
for w in "$@"; do
  python bench.py --workload $w --no-cpu-baseline --no-dealer-streams > gpurun_out/$TAG/$w.json 2> gpurun_out/$TAG/$w.err
  python - gpurun_out/$TAG/$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["name"], "online %.4f ms (%.4f without kernel timers)  offline %s  incl %s" % (
    d["ms_per_step"], d["ms_per_step_without_kernel_timers"], d.get("offline_ms"), d.get("epoch_time_incl_offline_s") and 1e3 * d["epoch_time_incl_offline_s"]))
PY
done
