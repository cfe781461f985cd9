#!/bin/bash
# end-of-round evidence: bench lines of every workload, kernel stats, PMC passes (profiles/r02_c_*)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 500 python bench.py > $O/r02_c_c3_bench.json.log 2> $O/r02_c_c3_bench.err; echo "c3 rc=$?"
timeout -k 10 300 python bench.py --workload c2 --steps 10 --warmup 2 > $O/r02_c_c2_bench.json.log 2> $O/r02_c_c2_bench.err; echo "c2 rc=$?"
for wl in g3 g3n g2; do timeout -k 10 400 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $O/r02_c_${wl}_bench.json.log 2> $O/r02_c_${wl}_bench.err; echo "$wl rc=$?"; done
CAPS_SA_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/r02_c_c3_sharded_world1_bench.json.log 2> $O/r02_c_sharded.err; echo "sharded rc=$?"
CAPS_SA_PATH=classic timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $O/r02_c_c3_samplesort_bench.json.log 2> $O/r02_c_classic.err; echo "classic rc=$?"
timeout -k 10 400 python tools/shard_probe.py c3 > $O/r02_c_c3_shard_probe_rank0.json.log 2> $O/r02_c_probe.err; echo "probe rc=$?"
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r02_c_prof -o r02_c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-verify > $GRAFT_REPO_ROOT/$O/r02_c_prof.log 2>&1
cd "$GRAFT_REPO_ROOT"; cp $O/r02_c_prof/r02_c_kernel_stats.csv $O/r02_c_c3_rocprofv3_kernel_stats.csv; rm -rf $O/r02_c_prof
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r02_c_gprof -o r02_c -- python3 $GRAFT_REPO_ROOT/bench.py --workload g3 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-verify > $GRAFT_REPO_ROOT/$O/r02_c_gprof.log 2>&1
cd "$GRAFT_REPO_ROOT"; cp $O/r02_c_gprof/r02_c_kernel_stats.csv $O/r02_c_g3_rocprofv3_kernel_stats.csv; rm -rf $O/r02_c_gprof
WL=c3 GROUPS_="lds wait fetch write" bash tools/pmc.sh > /dev/null 2>&1
cp $O/pmc_c3_summary.txt $O/r02_c_c3_rocprofv3_pmc_summary.txt; rm -rf $O/pmc_c3_*
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_c_*_bench.json.log")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], "verify", d.get("verify_errors"), "dom", r.get("kernel"), round(r.get("frac") or 0,3), {k:round(v["avg_launch_ms"],2) for k,v in (r.get("kernels") or {}).items()})
    except Exception as e: print(f, "ERR", e)
PY
