"""First and later calls of the host-buffer entry point at C3 size, LCP as bytes on / off (CAPS_SA_HOST_NARROW_LCP), with the
allocation times of the cached blocks (CAPS_SA_DEBUG_ALLOC)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import caps_sa_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000_001
L = caps_sa_amd.lib()
T = L.pinned_empty(n, "uint8"); L.gen_rand_seq(42, n - 1, T); T[n - 1] = ord("C")
os.environ["CAPS_SA_DEBUG_ALLOC"] = "1"
SA = L.pinned_empty(n, np.uint32); LCP = L.pinned_empty(n, np.uint32)
for narrow in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("1", "0", "1")):
    os.environ["CAPS_SA_HOST_NARROW_LCP"] = narrow
    L.release_cache()
    w = []
    for it in range(3):
        t0 = time.time(); st = L.build_into(T, SA, LCP, p=8000); w.append(1e3 * (time.time() - t0))
    print(json.dumps({"narrow": narrow, "calls_ms": [round(x, 1) for x in w], "h2d": round(st["ms_h2d"], 1), "build": round(st["ms_total"], 1),
                      "d2h": round(st["ms_d2h"], 1), "lcp_bytes_on_link": st["lcp_bytes_on_link"]}), flush=True)
