"""LCP distribution of a bench workload (how many neighbours of the sorted order tie on their 32-base keys: what the
tile sort pays text reads for).  python tools/lcp_hist.py g3"""
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
st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000)
out = {"workload": wl, "n": n, "ms_total": st["ms_total"]}
for th in (12, 16, 20, 24, 28, 32, 40, 48, 64, 96, 128, 256, 1024):
    c = 0
    for o in range(0, n, 1 << 28):
        c += int((LCP[o:o + (1 << 28)] >= th).sum().item())
    out[f"lcp>={th}"] = round(c / n, 5)
print(json.dumps(out))
