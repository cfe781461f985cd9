#!/bin/bash
# bench lines of a list of workloads: WLS="g3 g3n" TAG=r3d bash tools/r3_bench.sh   (EXTRA = more bench.py flags)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
for wl in ${WLS:-g3}; do timeout -k 10 400 python bench.py --workload $wl --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-host-path $EXTRA > $O/${TAG}_${wl}.json 2> $O/${TAG}_${wl}.err; echo "$wl rc=$?"; done
python - <<PY
import json,os
for wl in "${WLS:-g3}".split():
    try:
        d=json.loads(open(f"gpurun_out/${TAG}_{wl}.json").read().strip().splitlines()[-1])
        print(wl, round(d["ms_per_step"],1), "verify", d["verify_errors"], {k:round(v,1) for k,v in d["phases_ms"].items() if v}, d["config"]["merge_passes"])
    except Exception as e: print(wl, "ERR", e, open(f"gpurun_out/${TAG}_{wl}.err").read()[-500:])
PY
