#!/bin/bash
# per-kernel times of one genome-like build (rocprofv3 --kernel-trace --stats); output parsed on the box, trace not kept
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_genome
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_genome -- python3 /root/repo/tools/genome_like.py ${N:-268435456} > /root/repo/gpurun_out/prof_genome.log 2>&1 || exit 1
f=$(ls /tmp/prof_genome/*/*kernel_stats.csv | head -1)
python3 - $f <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "caps::" in r["Name"] and int(r["TotalDurationNs"])>4e5:
        print("  %-50s calls %3s  %8.2f ms per build" % (r["Name"].split("(")[0].replace("void caps::","")[:50], r["Calls"], int(r["TotalDurationNs"])/2e6))
PY
rm -rf /tmp/prof_genome
