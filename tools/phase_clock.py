"""Phase clock of the tile sort kernels (measurement variant: make -C caps-sa_amd variant TAG=phase VARIANT_DEFS=-DCAPS_PHASE_CLOCK,
run with CAPS_SA_LIB=variants/libcaps_sa_hip_phase.so).  Prints, per workload, the share of thread-0 cycles that
each barrier-separated phase of tile_sort_kernel / tile_sort_eq_kernel took (summed over all workgroups of one build)."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import caps_sa_amd  # noqa: E402
from bench import WORKLOADS, make_text  # noqa: E402

NAMES = {0: "ts.load", 1: "ts.hist", 2: "ts.scan", 3: "ts.place", 4: "ts.rank", 5: "ts.final", 6: "ts.emit",
         8: "eq.load", 9: "eq.rounds", 10: "eq.hist", 11: "eq.scan", 12: "eq.place", 13: "eq.rank", 14: "eq.final", 15: "eq.emit", 16: "eq.rank_keys", 17: "eq.rank_ties", 18: "eq.big_ties"}

L = caps_sa_amd.lib()
raw = ctypes.CDLL(caps_sa_amd.LIB_PATH)
clk = (ctypes.c_uint64 * 32)()
for wl in sys.argv[1:] or ["c3"]:
    n_bases, kind, _ = WORKLOADS[wl]
    T = make_text(torch, n_bases, 42, torch.device("cuda", 0), kind)
    n = T.numel()
    SA = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP = torch.empty(n, dtype=torch.int32, device="cuda")
    L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000)
    torch.cuda.synchronize()
    assert raw.caps_sa_hip_phase_clock(clk) == 0
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000)
    torch.cuda.synchronize()
    assert raw.caps_sa_hip_phase_clock(clk) == 0
    v = list(clk)
    out = {"workload": wl, "ms_total": round(st["ms_total"], 1), "tile_sort_ms": round(st["tile_sort_ms"], 1)}
    for grp, lo, hi in (("ts", 0, 8), ("eq", 8, 19)):
        tot = sum(v[lo:hi])
        if tot:
            out[grp] = {NAMES[i]: round(v[i] / tot, 3) for i in range(lo, hi) if i in NAMES}
            out[grp + "_Gcycles"] = round(tot / 1e9, 2)
    print(json.dumps(out))
    del T, SA, LCP
    torch.cuda.empty_cache()
