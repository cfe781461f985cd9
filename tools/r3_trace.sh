#!/bin/bash
# per-launch kernel trace of ONE build (the last of the run): WL=g3 bash tools/r3_trace.sh -> gpurun_out/<TAG>_trace.txt
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
WL=${WL:-g3}
TAG=${TAG:-r3_trace_$WL}
R=$GRAFT_REPO_ROOT
TXT=""
case $WL in g*) python3 $R/tools/make_text.py $WL /tmp/trace_text_$WL.npy && TXT="--text-file /tmp/trace_text_$WL.npy" ;; esac
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/${TAG}_d -o $TAG -- python3 $R/bench.py --workload $WL $TXT --steps 1 --warmup 1 --prewarm-s 0 --no-cpu-baseline --no-host-path --no-verify $EXTRA > $R/gpurun_out/$TAG.log 2>&1
echo "trace rc=$?"
cd $R
f=$(find /tmp/${TAG}_d -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/${TAG}_trace.txt <<'PY'
import csv,sys,re
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "caps::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last build starts at the first launch of the last group of alphabet_kernel launches
cut=0
for i,r in enumerate(rows):
    if "alphabet_kernel" in r["Kernel_Name"] and (i==0 or "alphabet_kernel" not in rows[i-1]["Kernel_Name"]): cut=i
rows=rows[cut:]
t0=int(rows[0]["Start_Timestamp"])
out=[];
for r in rows:
    name=re.sub(r"^void caps::","",r["Kernel_Name"]); name=re.sub(r"\(.*","",name)
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if out and out[-1][0]==name: out[-1][1]+=1; out[-1][2]+=e-s; out[-1][4]=e
    else: out.append([name,1,e-s,s,e])
for name,c,d,s,e in out:
    print(f"{(s-t0)/1e6:9.3f} ms  +{d/1e6:8.3f} ms  x{c:<4d} {name[:120]}")
print("total span ms", (int(rows[-1]["End_Timestamp"])-t0)/1e6)
PY
rm -rf /tmp/${TAG}_d
head -120 gpurun_out/${TAG}_trace.txt
