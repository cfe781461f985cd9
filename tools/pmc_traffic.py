#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh (GROUPS_="fetch write"), stamped with the hash of
the kernel sources they profiled (bench.py uses the figures only while that hash matches the sources it runs on).
usage: pmc_traffic.py <workload> gpurun_out/pmc_<wl>_fetch gpurun_out/pmc_<wl>_write   (rocprofv3 -d directories)"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha256  # noqa: E402

# kernel name prefix -> family of bench.py's roofline (first match wins)
FAMILY = [("group_scatter_kernel", "level_a_scatter"), ("bucket_scatter_kernel", "level_b_scatter"), ("bucket_count_kernel", "level_b_count"),
          ("spill_gather_kernel", "level_b_scatter"), ("spill_place_kernel", "level_b_scatter"),
          ("tile_sort_eq_kernel", "tile_sort_kernel"), ("tile_sort_kernel", "tile_sort_kernel"), ("tile_sort_general_kernel", "tile_sort_kernel"),
          ("merge_pass_kernel", "merge_pass_kernel"), ("merge_partition_kernel", "merge_pass_kernel"),
          ("pack_kernel", "pack"), ("alphabet_kernel", "pack"), ("run_blocks_kernel", "pack"), ("run_chunk_heads_kernel", "pack"),
          ("run_carry_kernel", "pack"), ("run_resolve_kernel", "pack"),
          ("msd_", "deferred_ties"), ("finalize_kernel", "finish_gather_head_lcps"), ("head_lcp_kernel", "finish_gather_head_lcps"),
          ("sample_", "sample_pivots"), ("pick_pivots_kernel", "sample_pivots"), ("knot", "sample_pivots"), ("group_keys_kernel", "sample_pivots"),
          ("group_caps_kernel", "sample_pivots"), ("skew_probe_kernel", "sample_pivots"), ("make_keys_kernel", "sample_pivots"),
          ("locate_kernel", "locate_pivots"), ("collate_kernel", "collate"), ("run_", "run_buckets")]


def sums(d, counter):
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    out = collections.defaultdict(float)
    for row in csv.DictReader(open(files[-1])):
        if row["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void caps::", "", row["Kernel_Name"])
        for pat, fam in FAMILY:
            if name.startswith(pat + "<") or name.startswith(pat + "(") or (pat.endswith("_") and name.startswith(pat)) or (pat == "knot" and name.startswith(pat)):
                out[fam] += float(row["Counter_Value"]) * 1024.0          # both counters are reported in KB
                break
    return out


wl, dfetch, dwrite = sys.argv[1:4]
fetch, write = sums(dfetch, "FETCH_SIZE"), sums(dwrite, "WRITE_SIZE")
path = os.path.join(ROOT, "profiles", "traffic.json")
try:
    tj = json.load(open(path))
except OSError:
    tj = {}
sha = kernel_sources_sha256()
if tj.get("kernel_sources_sha256") != sha:                               # other sources: every workload's figures are stale
    tj = {}
tj["_comment"] = ("HBM bytes per BUILD and kernel family from rocprofv3 PMC passes (tools/pmc.sh: separate --pmc passes with --kernel-trace "
                  "only, one build per pass): FETCH_SIZE x 2 (the gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md) + "
                  "WRITE_SIZE, both reported in KB.  Written by tools/pmc_traffic.py; valid for the kernel sources of kernel_sources_sha256.")
tj["kernel_sources_sha256"] = sha
tj[wl] = {fam: {"traffic_bytes_per_launch": 2.0 * fetch.get(fam, 0.0) + write.get(fam, 0.0),
                "fetched_x2": 2.0 * fetch.get(fam, 0.0), "written": write.get(fam, 0.0)} for fam in sorted(set(fetch) | set(write))}
json.dump(tj, open(path, "w"), indent=2)
print(json.dumps(tj[wl], indent=1))
