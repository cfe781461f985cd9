#!/bin/bash
# full-size genome-like builds with other seeds than the tests' (exact device verifier in every bench line)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
for seed in ${SEEDS:-101 202 303}; do for wl in ${WLS:-g3 g3n g3r}; do
  timeout -k 10 300 python bench.py --workload $wl --seed $seed --steps 1 --warmup 0 --prewarm-s 0 --no-cpu-baseline --no-host-path > $O/seed_${wl}_$seed.json 2> $O/seed_${wl}_$seed.err; echo "$wl seed $seed rc=$?"
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/seed_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["ms_per_step"],1), "verify_errors", d["verify_errors"], d["config"]["merge_passes"])
    except Exception as e: print(f, "ERR", e)
PY
