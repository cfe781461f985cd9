#!/bin/bash
# round 3, call F: host path (waves), 2-bit arena, u64 at 2.5e9, full -m gpu suite
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r3f_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/r3f_tests.log
timeout -k 10 300 python tools/u64_rate.py 2500000001 > $O/r3f_u64.json 2> $O/r3f_u64.err; echo "u64 rc=$?"; cat $O/r3f_u64.json
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/r3f_c3.json 2> $O/r3f_c3.err; echo "c3 rc=$?"
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline --steps 10 --warmup 2 > $O/r3f_c2.json 2> $O/r3f_c2.err; echo "c2 rc=$?"
python - <<'PY'
import json
for f in ("r3f_c3","r3f_c2"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],2), "verify", d["verify_errors"], "ws_gb", d["config"]["workspace_gb"], "pcie", d.get("pcie_inclusive"))
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-800:])
PY
