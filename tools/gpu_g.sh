#!/bin/bash
# genome-like workloads: bench lines for the workloads given as arguments ("g3", "g3:ENV=VAL,...")
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
for spec in "$@"; do
  wl=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  tag=$(echo "$spec" | tr ':=,' '___')
  ( IFS=,; for kv in $envs; do export "$kv"; done
    timeout -k 10 400 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $O/g_bench_$tag.json 2> $O/g_bench_$tag.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/g_bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", "ms/step %.2f" % d["ms_per_step"], "verify", d.get("verify_errors"), d["config"]["construction"][:12], "groups", d["config"]["groups"], "slots", d["config"]["slot_splits"], "passes", d["config"]["merge_passes"], {k: round(v,1) for k,v in d["phases_ms"].items() if v > 0.05})
except Exception as e:
    print("$tag failed", e, open("$O/g_bench_$tag.err").read()[-1500:])
PY
  )
done
