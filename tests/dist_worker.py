"""Worker of tests/test_dist_gloo.py: runs the sharded driver on CPU tensors over gloo with
the host emulation of the kernels, gathers the slices and compares with the oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import caps_sa_dist
    caps_sa_dist.A2A_MAX_BYTES = int(os.environ.get("CAPS_A2A_MAX_BYTES", caps_sa_dist.A2A_MAX_BYTES))
    import oracle as O
    from emul_util import emul, emul_small
    E = emul_small() if os.environ.get("CAPS_EMUL_SMALL") == "1" else emul()
    if os.environ.get("CAPS_DIST_CASE") == "imbalance":
        # samplesort path, 3 partitions for 2 ranks: one rank would own 2/3 of the suffixes -- more than any shard's
        # buffers hold.  Every rank must fail (together, before the exchange), none may hang.
        T = torch.from_numpy(np.random.RandomState(5).choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=2_000_000))
        try:
            caps_sa_dist.build_sharded(E, T, 3, 32)
        except Exception as e:                                       # noqa: BLE001
            print(f"rank {rank}: refused: {e}", flush=True)
            dist.destroy_process_group()
            sys.exit(0 if "imbalanced" in str(e) else 1)
        sys.exit(1)
    cases = []
    rs = np.random.RandomState(11)
    dna = np.frombuffer(b"ACGT", dtype=np.uint8)
    cases.append((rs.choice(dna, size=150_001), 12, 32))
    cases.append((rs.choice(dna, size=40_000), 7, 32))                      # p not divisible by world
    cases.append((rs.choice(np.frombuffer(b"abcdefgh", dtype=np.uint8), size=60_000), 16, 64))
    cases.append((np.tile(rs.choice(dna, size=37), 300), 5, 32))            # deep LCPs, skewed partitions
    cases.append((rs.choice(dna, size=100), 0, 32))                          # p_eff = 6
    cases.append((rs.choice(dna, size=600_000), 8, 32))                      # big enough for the slot splits (kept)
    cases.append((rs.choice(dna, size=500_000, p=[0.6, 0.2, 0.1, 0.1]), 6, 64))   # ... and redone on skewed keys
    runs = rs.choice(dna, size=180_000)
    runs[20_000:26_000] = ord("G")                                           # an N-block: long run -> every rank falls back together
    cases.append((runs, 0, 32))
    # tandem arrays: thousands of suffixes with one key -- deferred and re-keyed on the rank that owns them (kernels.h "Deferred ties";
    # every rank sorts what it scattered itself: not in exchange mode), checked against the independent construction
    rep = rs.choice(dna, size=160_000)
    for at, unit, copies, rate in [(30_000, 23, 700, 0.003), (90_000, 57, 150, 0.02), (160_000 - 31 * 40, 31, 40, 0.0)]:
        seg = np.tile(rs.choice(dna, size=unit), copies)
        mut = rs.rand(seg.size) < rate
        seg[mut] = rs.choice(dna, size=int(mut.sum()))
        rep[at:at + seg.size] = seg
    cases.append((rep, 0, 32))
    if os.environ.get("CAPS_DIST_CASE") == "repeats":
        cases = [(rep, 0, 32), (rep[:120_001].copy(), 0, 64)]
        rep2 = cases[1][0]
    else:
        rep2 = None
    if world >= 8:          # a node's worth of ranks: every case with a few partitions per rank (fewer is refused, see "imbalance")
        cases = [(rs.choice(dna, size=150_001), 24, 32), (rs.choice(np.frombuffer(b"abcdefgh", dtype=np.uint8), size=60_000), 32, 64),
                 (rs.choice(dna, size=200_000, p=[0.6, 0.2, 0.1, 0.1]), 40, 64), (runs, 0, 32)]
    ok = True
    for T_np, p, bits in cases:
        T = torch.from_numpy(T_np.copy())
        SA, LCP, off, info = caps_sa_dist.build_sharded(E, T, p, bits)
        cnt = torch.tensor([SA.numel()], dtype=torch.int64)
        cnts = torch.empty(world, dtype=torch.int64)
        dist.all_gather_into_tensor(cnts, cnt)
        counts = [int(x) for x in cnts.tolist()]
        offs = torch.empty(world, dtype=torch.int64)
        dist.all_gather_into_tensor(offs, torch.tensor([off], dtype=torch.int64))
        assert offs.tolist() == [sum(counts[:r]) for r in range(world)], (offs.tolist(), counts)
        SA_all = caps_sa_dist._all_gather_var(SA, counts).numpy()
        LCP_all = caps_sa_dist._all_gather_var(LCP, counts).numpy()
        dt = np.uint32 if bits == 32 else np.uint64
        if T_np is rep or T_np is rep2:
            from sa_check import sa_lcp
            SAo, LCPo = sa_lcp(T_np, idx_bits=bits)
        else:
            SAo, LCPo = (O.naive_sa_lcp(T_np, idx_bits=bits) if T_np.size <= 200_000 else O.build_sa_lcp(T_np, p=p, idx_bits=bits)[:2])
        good = np.array_equal(SA_all.view(dt), SAo) and np.array_equal(LCP_all.view(dt), LCPo)
        ties = torch.tensor([int(info.get("tie_groups_deferred", 0))], dtype=torch.int64)
        dist.all_reduce(ties)
        if T_np is rep and info["path"] == "direct" and not info.get("exchange", 1):
            # (CAPS_SA_TEST_MSD_FAIL: the refinement "does not fit" -- the rank scatters again and compares every tie)
            good = good and (int(ties) == 0 if os.environ.get("CAPS_SA_TEST_MSD_FAIL") else int(ties) > 0)
        # the slice-wise verifier bench.py uses at N > 1 (nothing gathered): clean on the result, loud on a corrupted one
        good = good and caps_sa_dist.verify_sharded(E, T, SA, LCP, off, bits) == 0
        if T_np.size >= 40_000:
            bad_sa, bad_lcp = SA.clone(), LCP.clone()
            if rank == world - 1 and bad_sa.numel() > 1:
                bad_sa[0], bad_sa[1] = SA[1].clone(), SA[0].clone()           # the slice head swapped with its neighbour
            if rank == 0 and bad_lcp.numel() > 3:
                bad_lcp[3] += 1
            good = good and caps_sa_dist.verify_sharded(E, T, bad_sa, bad_lcp, off, bits) >= 2
        if rank == 0:
            print(f"case n={T_np.size} p={p} bits={bits} path={info['path']} fb={info.get('direct_fallback')} keys={info.get('key_bytes')} "
                  f"retry={info.get('key_retry')} exch={info.get('exchange')} quant={info.get('direct_quantile')} ties={int(ties)} counts={counts} "
                  f"{'OK' if good else 'MISMATCH'}", flush=True)
        ok = ok and good
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
