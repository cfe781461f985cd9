#!/bin/bash
# round 3, call E: the default bench line (C3 from the MT19937 stream), the sharded driver at world 1, PMC traffic passes of C3
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
(time timeout -k 10 600 python bench.py > $O/r3e_c3.json 2> $O/r3e_c3.err) 2> $O/r3e_c3.time; echo "c3 rc=$?"; tail -3 $O/r3e_c3.time
CAPS_SA_FORCE_SHARDED=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-host-path > $O/r3e_sharded1.json 2> $O/r3e_sharded1.err; echo "sharded rc=$?"
WL=c3 GROUPS_="fetch write" bash tools/pmc.sh > /dev/null 2>&1
python tools/pmc_traffic.py c3 $O/pmc_c3_fetch $O/pmc_c3_write > $O/r3e_traffic.log 2>&1; echo "traffic rc=$?"
cp profiles/traffic.json $O/r3e_traffic.json
python - <<'PY'
import json
for f in ("r3e_c3","r3e_sharded1"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        r=d.get("roofline") or {}
        print(f, round(d["ms_per_step"],2), "verify", d["verify_errors"], "roof", r.get("kernel"), round(r.get("frac") or 0,3), "traffic", r.get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "pcie", (d.get("pcie_inclusive") or {}).get("pinned_results",{}).get("steady_ms"), d.get("whole_build_floor"))
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-800:])
PY
