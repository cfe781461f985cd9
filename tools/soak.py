"""Soak: the same build over and over, the exact device verifier after every one (a rare race would show as a rare error).
python tools/soak.py [workload] [builds]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import caps_sa_amd  # noqa: E402
from bench import WORKLOADS, make_text  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
builds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n_bases, kind, _ = WORKLOADS[wl]
L = caps_sa_amd.lib()
T = make_text(torch, n_bases, 42, torch.device("cuda", 0), kind)
n = T.numel()
SA = torch.empty(n, dtype=torch.int32, device="cuda")
LCP = torch.empty(n, dtype=torch.int32, device="cuda")
ws_bytes = L.workspace_bytes(n, 8000, 32)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
bad = 0
first = None
for i in range(builds):
    SA.fill_(-1)
    LCP.fill_(-1)
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes)
    e = L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr())
    if first is None:
        first = (SA.clone(), LCP.clone()) if n <= 300_000_000 else None
    elif first is not None and not (torch.equal(SA, first[0]) and torch.equal(LCP, first[1])):
        e += 1
    bad += 1 if e else 0
    if (i + 1) % 10 == 0:
        print(json.dumps({"workload": wl, "builds": i + 1, "builds_with_errors": bad, "ms_last": round(st["ms_total"], 2)}), flush=True)
print(json.dumps({"workload": wl, "builds": builds, "builds_with_errors": bad}))
