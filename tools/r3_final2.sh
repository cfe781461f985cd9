#!/bin/bash
# round-3 evidence, call 2: rocprofv3 kernel stats (c3, g3, g3r) and PMC passes (c3, g3) + profiles/traffic.json figures
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
T=${TAG:-r03_a}
for wl in ${PROF_WLS:-c3 g3 g3r}; do
  WL=$wl TAG=${T}_${wl}_rocprofv3 bash tools/r3_prof.sh > $O/${T}_${wl}_prof_top.txt 2>&1; echo "prof $wl rc=$?"
done
for wl in ${PMC_WLS:-c3 g3}; do
  WL=$wl GROUPS_="lds wait fetch write" bash tools/pmc.sh > /dev/null 2>&1
  cp $O/pmc_${wl}_summary.txt $O/${T}_${wl}_rocprofv3_pmc_summary.txt
  python3 tools/pmc_traffic.py $wl $O/pmc_${wl}_fetch $O/pmc_${wl}_write > $O/${T}_${wl}_traffic.txt 2>&1; echo "traffic $wl rc=$?"
  rm -rf $O/pmc_${wl}_lds $O/pmc_${wl}_wait $O/pmc_${wl}_fetch $O/pmc_${wl}_write
done
cp profiles/traffic.json $O/${T}_traffic.json
head -30 $O/${T}_c3_rocprofv3_pmc_summary.txt
