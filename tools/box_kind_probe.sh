#!/bin/bash
# tools/box_kind_probe.sh -- which kind of box is this (level A runs in 11.7-12.0 ms on some boxes, 13.6-13.7 on others)?
# Prints what rocm-smi says about clocks, power cap, memory and partition modes, then level A's time in a C3 build.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
rocm-smi --showclocks --showpower --showmaxpower --showmemorypartition --showcomputepartition --showperflevel 2>&1 | grep -v "^=\|^$" | head -40
python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > gpurun_out/box_probe_c3.json.log 2>/dev/null &
BP=$!
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  sleep 3
  echo "t=$((3*i))s $(rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|Package Power" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ';' | cut -c1-230)"
  kill -0 $BP 2>/dev/null || break
done
wait $BP
python3 tools/bench_compare.py box_probe_c3 | cut -c1-200
