#!/usr/bin/env python3
"""Per-kernel sums of the rocprofv3 PMC passes written by tools/pmc.sh.
usage: pmc_summary.py gpurun_out/pmc_<wl>_<group> [...]   (each a rocprofv3 -d directory)"""
import csv, glob, os, sys, collections, re


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("caps::", "")
    return re.sub(r"\(.*$", "", name)[:70]


for d in sys.argv[1:]:
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    traces = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        print("==", d, ": no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for row in csv.DictReader(open(files[-1])):
        k = short(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
    dur = collections.defaultdict(float)
    if traces:
        for row in csv.DictReader(open(traces[-1])):
            dur[short(row["Kernel_Name"])] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    print("==", d)
    for k in agg:
        if dur.get(k, 0) < 0.3 and traces: continue
        print(f"{k:70s} n={len(disp[k]):3d} ms={dur.get(k,0):8.2f} " + " ".join(f"{c}={v:.4g}" for c, v in sorted(agg[k].items())))
