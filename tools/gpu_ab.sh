#!/bin/bash
# A/B of library variants on bench workloads: tools/gpu_ab.sh "g3 g2" "default base ..."   (variants: caps-sa_amd/variants/libcaps_sa_hip_<tag>.so)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
for wl in $1; do
  for v in $2; do
    if [ "$v" = default ]; then unset CAPS_SA_LIB; else export CAPS_SA_LIB=$GRAFT_REPO_ROOT/caps-sa_amd/variants/libcaps_sa_hip_$v.so; fi
    timeout -k 10 400 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/ab_${wl}_$v.json 2> $O/ab_${wl}_$v.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/ab_${wl}_$v.json").read().strip().splitlines()[-1])
    print("$wl $v", "ms/step %.2f" % d["ms_per_step"], "verify", d.get("verify_errors"), {k: round(v,1) for k,v in d["phases_ms"].items() if v > 0.05})
except Exception as e:
    print("$wl $v failed", e, open("$O/ab_${wl}_$v.err").read()[-1500:])
PY
  done
done
