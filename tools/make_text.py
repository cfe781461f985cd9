"""Writes a bench workload's text to a .npy file:  python tools/make_text.py g3 /tmp/g3.npy
(bench.py --text-file reads it back: profiling runs that must not contain the generator's kernels)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import WORKLOADS, make_text  # noqa: E402

wl, out = sys.argv[1], sys.argv[2]
n_bases, kind, _ = WORKLOADS[wl]
T = make_text(torch, n_bases, 42, torch.device("cuda", 0), kind)
np.save(out, T.cpu().numpy())
print(wl, T.numel(), "->", out)
