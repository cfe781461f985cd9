#!/bin/bash
# per-kernel times of the genome-like build, equalised and plain (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
for mode in eq plain; do
  if [ $mode = plain ]; then export CAPS_SA_NO_EQUALISE=1; else unset CAPS_SA_NO_EQUALISE; fi
  rm -rf /root/repo/gpurun_out/prof_genome_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_genome_$mode -- python3 /root/repo/tools/genome_like.py ${N:-268435456} > /root/repo/gpurun_out/prof_genome_$mode.log 2>&1 || exit 1
  f=$(ls /root/repo/gpurun_out/prof_genome_$mode/*/*kernel_stats.csv | head -1)
  echo "== $mode"
  python3 - $f <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "caps::" in r["Name"] and int(r["TotalDurationNs"])>2e5:
        print("  %-46s calls %3s  %8.2f ms" % (r["Name"].split("(")[0].replace("void caps::","")[:46], r["Calls"], int(r["TotalDurationNs"])/2e6))
PY
done
