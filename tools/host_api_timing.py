"""Times the host-buffer entry point (what Suffix_Array::construct() calls): H2D + build + D2H,
into pageable and into page-locked result arrays; first call (allocates the cached device block)
and later calls (re-use it)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caps_sa_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 268_435_457
rs = np.random.RandomState(1)
T = np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=n)]
L = caps_sa_amd.lib()
for pinned in (False, True):
    L.release_cache()
    walls = []
    for it in range(4):
        t0 = time.time()
        SA, LCP, st = L.build(T, p=8000, pinned=pinned)
        walls.append(time.time() - t0)
        del SA, LCP
    t0 = time.time()
    a = L.pinned_empty(n, np.uint32) if pinned else np.empty(n, np.uint32)
    if not pinned:
        a[::1024] = 0                      # first touch of every page
    t_alloc = time.time() - t0
    del a
    wall = min(walls[1:])
    print(json.dumps({"n": n, "result_buffers": "pinned" if pinned else "pageable", "first_call_s": walls[0], "later_call_s": wall,
                      "suffixes_per_s_incl_pcie": n / wall, "ms_h2d": st["ms_h2d"], "ms_build": st["ms_total"], "ms_d2h": st["ms_d2h"],
                      "ms_other_host": 1e3 * wall - st["ms_h2d"] - st["ms_total"] - st["ms_d2h"],
                      "d2h_GBps": 2 * 4 * n / st["ms_d2h"] / 1e6, "h2d_GBps": n / st["ms_h2d"] / 1e6,
                      "one_result_array_alloc_s": t_alloc}), flush=True)
