#!/bin/bash
# quick GPU pass: a parity subset, C3 bench variants given as "TAG:ENV=VAL,..." arguments, kernel stats of the default
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "${TESTS:-c2_256 or c3_3g or random_dna or slot_splits or skewed}" > $O/b_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/b_tests.log
tail -3 $O/b_tests.log
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  ( IFS=,; for kv in $envs; do export "$kv"; done
    timeout -k 10 200 python bench.py --workload ${WL:-c3} --steps 5 --warmup 2 ${NOVERIFY:+--no-verify} --no-cpu-baseline --no-host-path > $O/b_bench_$tag.json 2> $O/b_bench_$tag.err
    python - <<PY
import json
d=json.loads(open("$O/b_bench_$tag.json").read().strip().splitlines()[-1])
print("$tag", "ms/step %.2f" % d["ms_per_step"], "verify", d.get("verify_errors"), {k: round(v,2) for k,v in d["phases_ms"].items() if v > 0.05})
PY
  )
done
if [ -n "$PROF" ]; then
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/b_prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --workload ${WL:-c3} --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $GRAFT_REPO_ROOT/$O/b_prof.log 2>&1
cd "$GRAFT_REPO_ROOT"; f=$(find $O/b_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 $f | head -12
fi
