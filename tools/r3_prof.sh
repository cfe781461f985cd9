#!/bin/bash
# rocprofv3 kernel stats of one bench workload: WL=g2r bash tools/r3_prof.sh  -> gpurun_out/r3_prof_<WL>_kernel_stats.csv (caps kernels only)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
WL=${WL:-g2r}
TAG=${TAG:-r3_prof_$WL}
R=$GRAFT_REPO_ROOT
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_d -o $TAG -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --prewarm-s 0 --no-cpu-baseline --no-host-path --no-verify > $R/gpurun_out/$TAG.log 2>&1
echo "prof rc=$?"
cd $R
f=$(find gpurun_out/${TAG}_d -name "*kernel_stats.csv" | head -1)
grep -E '^"Name"|caps::' "$f" > gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/${TAG}_d
python3 - "$TAG" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(f"gpurun_out/{sys.argv[1]}_kernel_stats.csv")))
for r in rows[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:10.1f}')
PY
