"""BASELINE config 5 (C5): T = 'a'^n, 32-bit indices, default subproblem count, checked against
the closed form SA[i] = n-1-i, LCP[i] = i.  Also a text with long single-letter
blocks and tandem arrays planted in random DNA (stand-in for N-blocks / satellites), checked
with the device verifier when the planted stretches are short enough for it (it scans raw
bytes: O(sum of LCPs)).   usage: deep_lcp.py [n ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caps_sa_amd  # noqa: E402


def unary(L, n, p=0):
    T = torch.full((n,), ord("a"), dtype=torch.uint8, device="cuda")
    SA = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=p)
    torch.cuda.synchronize()
    dt = time.time() - t0
    ar = torch.arange(n, device="cuda", dtype=torch.int64)
    sa_bad = int(((SA.long() & 0xFFFFFFFF) != (n - 1 - ar)).sum().item())
    lcp_bad = int(((LCP.long() & 0xFFFFFFFF) != ar).sum().item())
    return {"case": "a^n", "n": n, "p_eff": st["p_eff"], "wall_ms": 1e3 * dt, "M_suffixes_per_s": n / dt / 1e6,
            "sa_errors": sa_bad, "lcp_errors": lcp_bad, "ms_sort_subarrays": st["ms_sort_subarrays"],
            "ms_locate_pivots": st["ms_locate_pivots"], "ms_merge_partitions": st["ms_merge_partitions"],
            "merge_passes": [st["merge_passes_phase1"], st["merge_passes_phase2"]]}


def planted(L, n, run_len, p=8000):
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    seq = torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.uint8)
    k = max(1, n // (50 * run_len))
    pos = torch.randint(0, n - run_len, (k,), device="cuda", generator=g).tolist()
    units = [b"G", b"A", b"AC", b"AAT", b"ACGTT", b"GGAT" * 3]
    for i, s in enumerate(pos):
        u = torch.tensor(list(units[i % len(units)]), dtype=torch.uint8, device="cuda")
        lut_inv = {65: 0, 67: 1, 71: 2, 84: 3}
        codes = torch.tensor([lut_inv[int(c)] for c in u.tolist()], dtype=torch.uint8, device="cuda")
        seq[s:s + run_len] = codes.repeat(run_len // len(codes) + 1)[:run_len]
    T = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[seq.long()]
    SA = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=p)
    torch.cuda.synchronize()
    dt = time.time() - t0
    errs = L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr()) if run_len * run_len * k < 4e12 else None
    return {"case": "planted runs", "n": n, "runs": k, "run_len": run_len, "wall_ms": 1e3 * dt,
            "M_suffixes_per_s": n / dt / 1e6, "verify_errors": errs, "max_lcp": int((LCP.long() & 0xFFFFFFFF).max().item()),
            "merge_passes": [st["merge_passes_phase1"], st["merge_passes_phase2"]]}


if __name__ == "__main__":
    L = caps_sa_amd.lib()
    sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000, 100_000_000]
    for n in sizes:
        print(json.dumps(unary(L, n)), flush=True)
    print(json.dumps(planted(L, 64_000_000, 20_000)), flush=True)
    print(json.dumps(planted(L, 64_000_000, 1_000_000)), flush=True)
