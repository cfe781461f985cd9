"""What one rank of a sharded build costs on ITS GPU, for world sizes 1, 2, 4, 8 (one GPU per box: only rank 0's own work is
run, with its own report standing in for the other ranks' -- the offsets of the slice are then wrong, the times are not).
Local mode (default): scatter of the whole text keeping the owned groups + sort; exchange mode: CAPS_SA_SHARD_EXCHANGE=1.
python tools/shard_probe.py [workload]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import caps_sa_amd  # noqa: E402
from bench import WORKLOADS, make_text  # noqa: E402
from caps_sa_dist import ShardBuffers, _idx_dtype  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
# "c4": BASELINE config 4's shape (8 Gi random bases + the remapped newline, 64-bit indices), ranks of a world of 8 only (the
# buffers of a rank of fewer do not fit one GPU next to the text)
n_bases, kind, _ = WORKLOADS[wl]
L = caps_sa_amd.lib()
T = make_text(torch, n_bases, 42, torch.device("cuda", 0), kind)
n = T.numel()
bits = 32 if n <= 0xFFFFFFFF else 64
for world in ((8,) if wl == "c4" else (1, 2, 4, 8)):
    sh = L.shard(T.data_ptr(), n, 8000, bits, 0, world, 0)
    inf = sh.info()
    if inf["direct_fallback"] or inf["exchange"]:
        print(json.dumps({"world": world, "skipped": "not the local direct path", "fallback": inf["direct_fallback"]}))
        sh.close()
        continue
    B = ShardBuffers(inf, T.device, _idx_dtype(bits))
    rows = []
    for it in range(3):
        sh.scatter(B.send_k.data_ptr(), B.send_s.data_ptr(), B.report.data_ptr())
        rep = B.report.cpu().numpy().astype(np.uint64)
        code, sc, rc = sh.plan(np.stack([rep] * world))
        assert code == 0, code
        assert sh.sort_owned(B.send_k.data_ptr(), B.send_s.data_ptr(), B.SA.data_ptr(), B.LCP.data_ptr()) == 0
        i = sh.info()
        rows.append((i["ms_scatter"], i["ms_level_a"], i["ms_sort"], i["ms_level_b"], i["ms_tile_sort"]))
    i = sh.info()
    errs = L.verify_slice_device(T.data_ptr(), n, B.SA.data_ptr(), B.LCP.data_ptr(), i["recv_total"], True, idx_bits=bits)
    r = rows[-1]
    print(json.dumps({"world": world, "rank0_suffixes": i["recv_total"], "ms_scatter": round(r[0], 2), "of_it_level_a": round(r[1], 2),
                      "ms_sort": round(r[2], 2), "level_b": round(r[3], 2), "tile_sort": round(r[4], 2),
                      "ms_rank0_total": round(r[0] + r[2], 2), "slice_verify_errors": errs}))
    sh.close()
    del B
    torch.cuda.empty_cache()
