#!/bin/bash
# round 3, call A: phase clock of the tile sorts on g3 + c3, baseline bench of g3 / g3n on this round's box
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
CAPS_SA_LIB=$PWD/caps-sa_amd/variants/libcaps_sa_hip_phase.so timeout -k 10 500 python tools/phase_clock.py g3 c3 > $O/r3a_phase.log 2> $O/r3a_phase.err; echo "phase rc=$?"
for wl in g3 g3n; do timeout -k 10 400 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $O/r3a_${wl}.json 2> $O/r3a_${wl}.err; echo "$wl rc=$?"; done
cat $O/r3a_phase.log
