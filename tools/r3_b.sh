#!/bin/bash
# round 3, call B: the new -m gpu tests (large reference-made fixtures, genome-like at 3e9, g2r) + bench of g3r / g2r
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_golden or genome_like_3g or grch38" > $O/r3b_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r3b_tests.log
for wl in g3r g2r; do timeout -k 10 400 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $O/r3b_${wl}.json 2> $O/r3b_${wl}.err; echo "$wl rc=$?"; done
python - <<'PY'
import json
for wl in ("g3r","g2r"):
    try:
        d=json.loads(open(f"gpurun_out/r3b_{wl}.json").read().strip().splitlines()[-1])
        print(wl, round(d["ms_per_step"],1), "verify", d["verify_errors"], {k:round(v,1) for k,v in d["phases_ms"].items()}, d["config"]["merge_passes"], d["config"]["max_partition"])
    except Exception as e: print(wl, "ERR", e)
PY
