#!/bin/bash
# round-3 evidence, call 1: bench lines of every workload (TAG, default r03_a) -> gpurun_out/<TAG>_*_bench.json.log
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
T=${TAG:-r03_a}
timeout -k 10 500 python bench.py > $O/${T}_c3_bench.json.log 2> $O/${T}_c3_bench.err; echo "c3 rc=$?"
timeout -k 10 300 python bench.py --workload c2 --steps 10 --warmup 2 > $O/${T}_c2_bench.json.log 2> $O/${T}_c2_bench.err; echo "c2 rc=$?"
for wl in ${WLS:-g3 g3n g3r g2 g2r}; do timeout -k 10 400 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $O/${T}_${wl}_bench.json.log 2> $O/${T}_${wl}_bench.err; echo "$wl rc=$?"; done
CAPS_SA_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/${T}_c3_sharded_world1_bench.json.log 2> $O/${T}_sharded.err; echo "sharded rc=$?"
CAPS_SA_PATH=classic timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/${T}_c3_samplesort_bench.json.log 2> $O/${T}_classic.err; echo "classic rc=$?"
python - "$T" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(f"gpurun_out/{sys.argv[1]}_*_bench.json.log")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], "verify", d.get("verify_errors"), "dom", r.get("kernel"), round(r.get("frac") or 0,3), "traffic", r.get("traffic"), {k:round(v["avg_launch_ms"],2) for k,v in (r.get("kernels") or {}).items()})
    except Exception as e: print(f, "ERR", e)
PY
