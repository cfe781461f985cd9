#!/bin/bash
# round 3, call C: letter-run buckets on the GPU: the N-block tests + bench g3n / g3 / c3 (regression check)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "genome_like_3g or n_block or n_blocks or deep_lcp or quantile_mode or periodic" > $O/r3c_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r3c_tests.log
for wl in g3n g3 c3; do timeout -k 10 400 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/r3c_${wl}.json 2> $O/r3c_${wl}.err; echo "$wl rc=$?"; done
python - <<'PY'
import json
for wl in ("g3n","g3","c3"):
    try:
        d=json.loads(open(f"gpurun_out/r3c_{wl}.json").read().strip().splitlines()[-1])
        print(wl, round(d["ms_per_step"],1), "verify", d["verify_errors"], {k:round(v,1) for k,v in d["phases_ms"].items()}, d["config"]["merge_passes"], d["config"]["max_partition"])
    except Exception as e: print(wl, "ERR", e)
PY
