#!/bin/bash
# kernel stats of one genome-like workload (default g3): which tile-sort variant takes the time
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
WL=${1:-g3}
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/gprof -o gp -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-verify > $GRAFT_REPO_ROOT/$O/gprof_$WL.log 2>&1
cd "$GRAFT_REPO_ROOT"; cp $O/gprof/gp_kernel_stats.csv $O/gprof_${WL}_kernel_stats.csv; rm -rf $O/gprof
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/gprof_${WL}_kernel_stats.csv")))
for r in rows[:40]:
    nm=r["Name"]
    if "at::" in nm or "elementwise" in nm: continue
    print("%-90s calls %5s total_ms %9.2f avg_us %10.1f" % (nm[:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
tail -c 600 $O/gprof_$WL.log
