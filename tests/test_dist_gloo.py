"""CPU: the multi-GPU path (caps_sa_dist.py) with world size 2 and 3 over gloo.  The
collectives and all host logic are the real ones; the per-rank kernels run through the host
emulation of the kernel sources (no GPU in this container).  On GPUs the same driver runs
with backend nccl (RCCL) and libcaps_sa_hip.so (bench.py --gpus N)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,port,a2a_limit", [(2, 29531, 0), (3, 29532, 0), (2, 29534, 4096), (4, 29535, 0)])
def test_sharded_build_matches_oracle(world, port, a2a_limit):
    """a2a_limit > 0: exchange in rounds of that many bytes per peer (the path taken on GPUs when a
    sub-subarray block exceeds RCCL's safe message size)."""
    from emul_util import emul
    emul()   # build the emulation library once, before the ranks race for it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ)
    if a2a_limit:
        env["CAPS_A2A_MAX_BYTES"] = str(a2a_limit)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(" OK") == 7, r.stdout
