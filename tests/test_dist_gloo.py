"""CPU: the multi-GPU path (caps_sa_dist.py) with world size 2 and 3 over gloo.  The
collectives and all host logic are the real ones; the per-rank kernels run through the host
emulation of the kernel sources (no GPU in this container).  On GPUs the same driver runs
with backend nccl (RCCL) and libcaps_sa_hip.so (bench.py --gpus N)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,port,a2a_limit,small,path,exchange", [
    (2, 29531, 0, 0, "auto", 0), (3, 29532, 0, 0, "auto", 0), (2, 29534, 4096, 0, "auto", 1), (4, 29535, 0, 0, "auto", 0),
    (2, 29536, 0, 1, "auto", 1), (3, 29537, 4096, 1, "auto", 0), (3, 29533, 0, 1, "auto", 1), (2, 29538, 0, 0, "classic", 0),
    (2, 29539, 0, 1, "classic", 0), (8, 29542, 0, 1, "auto", 0), (8, 29543, 0, 1, "auto", 1)])
def test_sharded_build_matches_oracle(world, port, a2a_limit, small, path, exchange):
    """a2a_limit > 0: exchange in rounds of that many bytes per peer (the path taken on GPUs when a block exceeds
    RCCL's safe message size).  small: the kernels' 256-element-tile build (more tiles, groups and streams per case:
    the direct path on every case that is not tiny).  path: CAPS_SA_PATH (classic = samplesort path on every case).
    exchange: 0 = the direct path's default, every rank scatters the whole text and keeps its own groups (no data-path
    collective); 1 = CAPS_SA_SHARD_EXCHANGE: every rank scatters its share of the tiles, one all-to-all of the streams."""
    from emul_util import emul, emul_small
    (emul_small if small else emul)()   # build the emulation library once, before the ranks race for it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ, CAPS_SA_PATH=path, CAPS_EMUL_SMALL=str(small))
    env.pop("CAPS_SA_SHARD_EXCHANGE", None)
    if exchange:
        env["CAPS_SA_SHARD_EXCHANGE"] = "1"
    if a2a_limit:
        env["CAPS_A2A_MAX_BYTES"] = str(a2a_limit)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    if world >= 8:                                     # (its own, shorter case list: dist_worker.py)
        assert r.stdout.count(" OK") == 4 and "path=direct" in r.stdout, r.stdout
        return
    assert r.stdout.count(" OK") == 9, r.stdout
    if path == "classic":
        assert "path=direct" not in r.stdout
    else:
        assert r.stdout.count("path=direct") >= (4 if small else 3), r.stdout
        if exchange:
            assert "fb=3" in r.stdout                  # the long run: CAPS_SA_FB_LONG_RUNS, agreed on by every rank
            assert "keys=4 retry=0 exch=1" in r.stdout     # 32-bit keys crossed the (gloo) wire ...
            assert "keys=8 retry=1 exch=1" in r.stdout     # ... and a slot overflow under them sent every rank round again with 64
        else:
            assert "path=direct fb=0 keys=8 retry=0 exch=0" in r.stdout and "path=direct fb=0 keys=4" not in r.stdout
            if not small:                              # (256-element tiles: the shapes of these cases do not admit quantile buckets)
                assert "path=direct fb=0 keys=8 retry=0 exch=0 quant=1" in r.stdout   # skewed base frequencies / the long run: quantile buckets
                assert "n=180000 p=0 bits=32 path=direct" in r.stdout      # the long run no longer sends the ranks to the samplesort path
            assert "path=direct" in r.stdout and "exch=1 " not in "".join(l for l in r.stdout.splitlines() if "path=direct" in l)


def test_imbalanced_ownership_fails_on_every_rank_together():
    """with p < 4 x world the midpoint ownership can give a rank more than its buffers hold; every rank must
    raise before the exchange instead of one raising and the others waiting for it."""
    from emul_util import emul
    emul()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29540", os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ, CAPS_SA_PATH="classic", CAPS_DIST_CASE="imbalance")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("refused") == 2, r.stdout


@pytest.mark.parametrize("small", [0, 1])
def test_sharded_deferral_and_its_retry_over_gloo(small):
    """Deferred ties in the sharded build (shard.h sort_owned): tandem arrays on two ranks, u32 and u64 -- re-keyed on the rank that
    owns them; and with CAPS_SA_TEST_MSD_FAIL=1 (the refinement's work memory "does not fit"): every rank runs its level A again
    and sorts with every tie compared.  Both against tests/sa_check.py."""
    for fail in (False, True):
        env = dict(os.environ, CAPS_DIST_CASE="repeats", CAPS_EMUL_SMALL=str(small), PYTHONPATH=ROOT)
        env.pop("CAPS_SA_SHARD_EXCHANGE", None)
        if fail:
            env["CAPS_SA_TEST_MSD_FAIL"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", "29547", os.path.join(ROOT, "tests", "dist_worker.py")]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert r.stdout.count(" OK") == 2, r.stdout
        first = [ln for ln in r.stdout.splitlines() if ln.startswith("case n=160000 ")]
        assert len(first) == 1 and "path=direct" in first[0] and "exch=0" in first[0], r.stdout
        assert ("ties=0 " in first[0]) == fail, first[0]
