#!/bin/bash
# tools/evidence.sh -- every measurement of profiles/ comes from one of these sub-commands, run on the GPU box:
#   gpurun -- 'bash tools/evidence.sh <cmd> [args] && bash tools/evidence.sh <cmd> ...'
# TAG (default r04_a) prefixes the files written under gpurun_out/ (copy the ones to be judged into profiles/).
#   bench [wl ...]      bench lines (--no-cpu-baseline --no-host-path unless EXTRA overrides; STEPS, WARMUP)
#   default             `python bench.py` exactly as the driver runs it
#   cpufull             `python bench.py --cpu-full`: the oracle on the whole C3 text + entry-by-entry comparison
#   prof <wl>           rocprofv3 --kernel-trace --stats of 3 builds (caps kernels only)
#   pmc <wl> [groups]   PMC passes (tools/pmc.sh) + traffic figures into profiles/traffic.json
#   phase <wl ...>      phase clock of the tile sorts (variant `phase`)
#   gputest [-k expr]   the -m gpu suite in one process
#   variants <wl> v ... bench one workload under each tuning variant variants/libcaps_sa_hip_<v>.so
#   fault               tools/fault_probe.sh
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
T=${TAG:-r05}
cmd=$1; shift
summ() { python3 - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d.get("roofline") or {}; p=d.get("phases_ms") or {}
        print(f.split("/")[-1], "ms %.2f"%d["ms_per_step"], "verify", d.get("verify_errors"), "dom", r.get("kernel"), round(r.get("frac") or 0,3),
              {k:round(v["avg_launch_ms"],2) for k,v in (r.get("kernels") or {}).items()},
              {k:round(v,1) for k,v in p.items() if v and k in ("ms_pack","ms_select_pivots","bucket_count_ms","merge_pass_ms","tile_sort_ms","bucket_scatter_ms")},
              (d.get("config") or {}).get("merge_passes"))
    except Exception as e: print(f, "ERR", e)
PY
}
case $cmd in
  bench)
    for wl in "${@:-c3}"; do
      timeout -k 10 ${TMO:-420} python3 bench.py --workload $wl --steps ${STEPS:-3} --warmup ${WARMUP:-1} ${EXTRA---no-cpu-baseline --no-host-path} > $O/${T}_${wl}_bench.json.log 2> $O/${T}_${wl}_bench.err
      rc=$?; echo "bench $wl rc=$rc"; summ $O/${T}_${wl}_bench.json.log
      [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
    done ;;
  default)
    timeout -k 10 560 python3 bench.py > $O/${T}_c3_bench.json.log 2> $O/${T}_c3_bench.err; rc=$?; echo "default rc=$rc"; summ $O/${T}_c3_bench.json.log; exit $rc ;;
  cpufull)
    timeout -k 10 ${TMO:-1000} python3 bench.py --cpu-full --no-host-path --steps 3 > $O/${T}_c3_cpu_full_bench.json.log 2> $O/${T}_c3_cpu_full_bench.err; rc=$?
    echo "cpufull rc=$rc"; python3 -c "
import json;d=json.loads([l for l in open('$O/${T}_c3_cpu_full_bench.json.log') if l.startswith('{')][-1]);print(round(d['ms_per_step'],2),d['verify_errors'],json.dumps(d['cpu_baseline'])[:900],d.get('cpu_baseline_full_skipped'))"; exit $rc ;;
  prof)
    wl=$1; R=$PWD
    (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/${T}_${wl}_prof_d -o prof -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --prewarm-s 0 --no-cpu-baseline --no-host-path --no-verify > $R/$O/${T}_${wl}_prof.log 2>&1); rc=$?
    echo "prof $wl rc=$rc"
    f=$(find $O/${T}_${wl}_prof_d -name "*kernel_stats.csv" | head -1)
    grep -E '^"Name"|caps::' "$f" > $O/${T}_${wl}_rocprofv3_kernel_stats.csv; rm -rf $O/${T}_${wl}_prof_d
    python3 - $O/${T}_${wl}_rocprofv3_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:10.1f}')
PY
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1 ;;
  pmc)
    wl=$1; shift
    WL=$wl GROUPS_="${*:-lds wait fetch write}" bash tools/pmc.sh > $O/${T}_${wl}_pmc.log 2>&1; echo "pmc $wl rc=$?"
    cp $O/pmc_${wl}_summary.txt $O/${T}_${wl}_rocprofv3_pmc_summary.txt
    [ -d $O/pmc_${wl}_fetch ] && [ -d $O/pmc_${wl}_write ] && python3 tools/pmc_traffic.py $wl $O/pmc_${wl}_fetch $O/pmc_${wl}_write > $O/${T}_${wl}_traffic.txt 2>&1
    rm -rf $O/pmc_${wl}_lds $O/pmc_${wl}_wait $O/pmc_${wl}_fetch $O/pmc_${wl}_write $O/pmc_${wl}_tcc $O/pmc_${wl}_tcp $O/pmc_${wl}_grbm
    cp profiles/traffic.json $O/${T}_traffic.json; head -40 $O/${T}_${wl}_rocprofv3_pmc_summary.txt ;;
  phase)
    CAPS_SA_LIB=$PWD/variants/libcaps_sa_hip_${VAR:-phase}.so timeout -k 10 ${TMO:-500} python3 tools/phase_clock.py "$@" > $O/${T}_phase_clock.log 2>&1; rc=$?
    echo "phase rc=$rc"; grep '^{' $O/${T}_phase_clock.log; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1 ;;
  gputest)
    timeout -k 10 ${TMO:-1150} python3 -m pytest tests -q -m gpu -x --durations=15 "$@" > $O/${T}_gputest.log 2>&1; rc=$?
    echo "gputest rc=$rc"; tail -25 $O/${T}_gputest.log; exit $rc ;;
  variants)
    wl=$1; shift
    for v in "$@"; do
      lib=$PWD/variants/libcaps_sa_hip_$v.so; [ "$v" = base ] && lib=$PWD/caps-sa_amd/libcaps_sa_hip.so
      CAPS_SA_LIB=$lib timeout -k 10 ${TMO:-420} python3 bench.py --workload $wl --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-host-path > $O/${T}_var_${v}_${wl}.json.log 2> $O/${T}_var_${v}_${wl}.err
      rc=$?; echo "variant $v $wl rc=$rc"; summ $O/${T}_var_${v}_${wl}.json.log
      [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
    done ;;
  fault) bash tools/fault_probe.sh ;;
  *) echo "unknown sub-command $cmd"; exit 2 ;;
esac
exit 0
