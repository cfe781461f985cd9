"""Sizes of the groups of suffixes with equal 32-base keys (runs of LCP >= 32 in the suffix array) of a bench workload:
what the tile sorts' tie handling has to deal with.  python tools/tie_groups.py g3"""
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
n_bases, kind, _ = WORKLOADS[wl]
L = caps_sa_amd.lib()
T = make_text(torch, n_bases, 42, torch.device("cuda", 0), kind)
n = T.numel()
SA = torch.empty(n, dtype=torch.int32, device="cuda")
LCP = torch.empty(n, dtype=torch.int32, device="cuda")
L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000)
del T, SA
torch.cuda.empty_cache()
hist = {}
deep = {}
step = 1 << 27
for o in range(0, n, step):
    m = LCP[o:o + step] >= 32
    vals, counts = torch.unique_consecutive(m, return_counts=True)
    runs = counts[vals]                                   # lengths of the runs of ties: group size = run + 1
    for size, c in zip(*[x.tolist() for x in torch.unique(torch.clamp(runs + 1, max=33), return_counts=True)]):
        hist[size] = hist.get(size, 0) + c
tot_elems = sum(k * v for k, v in hist.items())
print(json.dumps({"workload": wl, "n": n, "groups_by_size(33=more)": {str(k): v for k, v in sorted(hist.items())},
                  "elements_in_groups": tot_elems, "share_of_n": round(tot_elems / n, 5),
                  "elements_in_pairs_share": round(2 * hist.get(2, 0) / max(tot_elems, 1), 4)}))
