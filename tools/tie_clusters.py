"""Clusters of suffixes with one and the same 32-base key in a bench workload (maximal runs of LCP >= 32 in the sorted order):
what share of the suffixes sits in clusters of which size -- the tile sort settles a cluster of c suffixes with c (c - 1) / 2
.. c (c - 1) comparisons through the text.  python tools/tie_clusters.py g3 [depth]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import caps_sa_amd  # noqa: E402
from bench import WORKLOADS, make_text  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "g3"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n_bases, kind, _ = WORKLOADS[wl]
L = caps_sa_amd.lib()
T = make_text(torch, n_bases, 42, torch.device("cuda", 0), kind)
n = T.numel()
SA = torch.empty(n, dtype=torch.int32, device="cuda")
LCP = torch.empty(n, dtype=torch.int32, device="cuda")
st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000)
edges = [1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256, 1024, 4096, 1 << 40]
elems = torch.zeros(len(edges), dtype=torch.int64, device="cuda")
pairs = torch.zeros(len(edges), dtype=torch.float64, device="cuda")
C = 1 << 27
for o in range(0, n, C):
    f = LCP[o:o + C] >= depth
    starts = (~f).nonzero().flatten()
    if starts.numel() == 0:
        continue
    sizes = torch.diff(starts, append=torch.tensor([f.numel()], device="cuda"))
    sq = (sizes * (sizes - 1)).double()
    for i, e in enumerate(edges):                          # cumulative (<= e); differences below
        m = sizes <= e
        elems[i] += (sizes * m).sum()
        pairs[i] += (sq * m).sum()
    print("chunk", o // C, file=sys.stderr, flush=True)
    del f, starts, sizes, sq
elems[1:] = elems[1:] - elems[:-1].clone()
pairs[1:] = pairs[1:] - pairs[:-1].clone()
out = {"workload": wl, "n": n, "depth": depth, "ms_total": st["ms_total"],
       "share_of_suffixes_by_cluster_size": {f"<={e}": round(float(elems[i]) / n, 6) for i, e in enumerate(edges)},
       "ordered_pairs_per_suffix_by_cluster_size": {f"<={e}": round(float(pairs[i]) / n, 5) for i, e in enumerate(edges)}}
print(json.dumps(out))
