"""Times the host-buffer entry point (what Suffix_Array::construct() calls): H2D + build + D2H."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caps_sa_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 268_435_457
rs = np.random.RandomState(1)
T = np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=n)]
L = caps_sa_amd.lib()
for it in range(3):
    t0 = time.time()
    SA, LCP, st = L.build(T, p=8000)
    wall = time.time() - t0
print(json.dumps({"n": n, "wall_s": wall, "suffixes_per_s_incl_pcie": n / wall, "ms_h2d": st["ms_h2d"], "ms_build": st["ms_total"],
                  "ms_d2h": st["ms_d2h"], "ms_alloc_free_and_host": 1e3 * wall - st["ms_h2d"] - st["ms_total"] - st["ms_d2h"], "d2h_GBps": 2 * 4 * n / st["ms_d2h"] / 1e6, "h2d_GBps": n / st["ms_h2d"] / 1e6}))
