#!/bin/bash
# full GPU parity run + default bench (with CPU baseline) + forced-sharded bench at world 1
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/c_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/c_tests.log
tail -4 $O/c_tests.log
( time timeout -k 10 500 python bench.py > $O/c_bench_default.json 2> $O/c_bench_default.err ) 2>&1 | grep real
CAPS_SA_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/c_bench_sharded1.json 2> $O/c_bench_sharded1.err
python - <<'PY'
import json
for f in ("c_bench_default","c_bench_sharded1"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no json:", e, open(f"gpurun_out/{f}.err").read()[-800:]); continue
    r=d.get("roofline") or {}
    print(f, "ms/step %.2f"%d["ms_per_step"], "verify", d.get("verify_errors"), "roof", r.get("kernel"), round(r.get("frac",0),3),
          {k:(round(v["avg_launch_ms"],2), round(v["frac"],3)) for k,v in (r.get("kernels") or {}).items()}, d.get("rank0_ms"), d.get("exchange"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("seconds"))
PY
