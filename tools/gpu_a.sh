#!/bin/bash
# first GPU pass of a change: parity tests, C3/C2 bench on both paths, kernel stats of the default path
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/a_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/a_tests.log
tail -5 $O/a_tests.log
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --verify --no-cpu-baseline --no-host-path > $O/a_bench_c3_direct.json 2> $O/a_bench_c3_direct.err && tail -c 2500 $O/a_bench_c3_direct.json
CAPS_SA_PATH=classic timeout -k 10 200 python bench.py --steps 5 --warmup 2 --verify --no-cpu-baseline --no-host-path > $O/a_bench_c3_classic.json 2> $O/a_bench_c3_classic.err && tail -c 600 $O/a_bench_c3_classic.json
timeout -k 10 200 python bench.py --workload c2 --steps 10 --warmup 2 --verify --no-cpu-baseline --no-host-path > $O/a_bench_c2_direct.json 2> $O/a_bench_c2_direct.err && tail -c 1500 $O/a_bench_c2_direct.json
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/a_prof -o a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $GRAFT_REPO_ROOT/$O/a_prof.log 2>&1
cd "$GRAFT_REPO_ROOT"; find $O/a_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-150 {} | head -12'
