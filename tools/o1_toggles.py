import json, os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, caps_sa_amd
from conftest import large_golden
T, sa, lcp = large_golden("latin1_signed_136k")
L = caps_sa_amd.lib()
for tg in [{}, {"CAPS_SA_LEVEL_A": "tile"}, {"CAPS_SA_DIRECT_SUB": "1"}, {"CAPS_SA_DIRECT_K1": "2"}, {"CAPS_SA_DIRECT_K1": "30"}, {"CAPS_SA_DIRECT_MODE": "linear", "CAPS_SA_LEVEL_A": "tile"}]:
    for k, v in tg.items(): os.environ[k] = v
    res = []
    for rep in range(2):
        SA, LCP, st = L.build(T, p=0)
        res.append([int((SA != sa).sum()), int((LCP != lcp).sum())])
    print(json.dumps({"toggle": tg, "bad": res, "direct": st["path_direct"], "groups": st["direct_groups"], "quantile": st["direct_quantile"]}), flush=True)
    for k in tg: os.environ.pop(k, None)
