"""GPU (-m gpu): 64-bit indices with text positions beyond 2^32 (BASELINE config 4 is n = 8 Gi + 1 on eight GPUs; the
reference switches index width at src/main.cpp:76-87 and instantiates u64 at src/Suffix_Array.cpp:544).  One MI355X cannot
hold a whole build of that size, so:

* the kernel-level entry points (a2 lcp, a4 sort_suffixes, a7 upper_bound) run on a 4.5e9-char text over suffixes that
  straddle position 2^32 -- a long single-letter run and a long repeat planted across / beyond the boundary -- against the
  oracle's merge_sort / lcp / upper_bound;
* the sharded build of a C4-size text (n = 8 * 2^30 + 1, world 8) is run for ONE destination rank: every rank's level A
  (shard_scatter: each takes every 8th tile of the whole text, so all of them handle positions >= 2^32) one after the
  other on this GPU, their blocks for rank 7 copied as the all-to-all would, then rank 7's plan + sort; its slice of
  SA / LCP (the last 1/8 of the suffix array) is checked with the exact slice verifier;
* the samplesort path's phase 1 at that size: rank 7 sorts its 1000 subarrays of text positions >= 7/8 n; every subarray
  is checked for order on the device and three of them against oracle.merge_sort.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TWO32 = 1 << 32


@pytest.fixture(scope="module")
def L():
    import torch  # noqa: F401
    import caps_sa_amd
    lib = caps_sa_amd.lib()
    if lib.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests need a GPU (there is no CPU fallback)")
    return lib


def _random_dna(n, seed):
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    T = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 28
    for o in range(0, n, step):
        m = min(step, n - o)
        T[o:o + m] = lut[torch.randint(0, 4, (m,), device="cuda", generator=g, dtype=torch.int64)]
    return T


def test_kernel_entry_points_across_2_pow_32(L, oracle):
    import torch
    n = 4_500_000_000
    T = _random_dna(n, 64)
    T[TWO32 - 3000:TWO32 + 3000] = ord("G")                              # a run across the boundary (run-table comparators)
    T[TWO32 + 100_000:TWO32 + 105_000] = T[1000:6000].clone()            # a repeat: deep LCP between a high and a low position
    Th = T.cpu().numpy()
    del T
    torch.cuda.empty_cache()
    rs = np.random.RandomState(5)
    half = 10_000_000
    idx = np.arange(TWO32 - half, TWO32 + half, dtype=np.uint64)
    idx = np.concatenate([idx, np.arange(1000, 6000, dtype=np.uint64)])   # the repeat's source, far below 2^32
    rs.shuffle(idx)
    sa, lcp = L.sort_suffixes(Th, idx, idx_bits=64)
    so, lo = oracle.merge_sort(Th, idx, idx_bits=64)
    assert np.array_equal(sa, so), "sorted suffix list differs from the oracle"
    assert np.array_equal(lcp, lo), "LCPs differ from the oracle"
    assert int(lcp.max()) >= 4999 and (sa >= TWO32).sum() >= half

    # a7: upper bounds of pivots from both sides of the boundary in that sorted list
    piv = np.concatenate([rs.randint(TWO32 - 2 * half, TWO32, size=600), rs.randint(TWO32, TWO32 + 2 * half, size=600),
                          np.array([TWO32 - 1, TWO32, TWO32 - 3000, TWO32 + 2999, TWO32 + 100_000, 1000])]).astype(np.uint64)
    ub = L.upper_bound(Th, sa, piv, idx_bits=64)
    exp = np.array([oracle.upper_bound(Th, sa, int(q), idx_bits=64) for q in piv], dtype=np.uint64)
    assert np.array_equal(ub, exp)

    # a2: lcp of pairs straddling the boundary (random, inside the run, into the repeat)
    a = np.concatenate([rs.randint(TWO32 - 5_000_000, TWO32 + 5_000_000, size=100_000),
                        np.arange(TWO32 - 3000, TWO32 + 2990, 7), np.arange(TWO32 + 100_000, TWO32 + 104_990, 11)]).astype(np.uint64)
    b = np.concatenate([rs.randint(TWO32 - 5_000_000, TWO32 + 5_000_000, size=100_000),
                        np.arange(TWO32 - 2990, TWO32 + 3000, 7), np.arange(1000, 5990, 11)]).astype(np.uint64)
    got = L.lcp(Th, a, b, idx_bits=64)
    exp = np.array([oracle.lcp(Th, int(x), int(y)) if x != y else n - int(x) for x, y in zip(a, b)], dtype=np.uint64)
    assert np.array_equal(got, exp)
    assert int(got.max()) >= 4000


C4_N = 8 * (1 << 30) + 1


@pytest.fixture(scope="module")
def c4_text():
    """BASELINE config 4's text shape: 8 Gi random bases + the remapped newline (n = 8,589,934,593 > 2^32: u64)."""
    T = _random_dna(C4_N, 42)
    T[C4_N - 1] = ord("C")
    return T


def test_c4_local_direct_path_for_the_last_rank(L, c4_text, monkeypatch):
    """The sharded direct path as it runs by default (no exchange) at BASELINE config 4's size: rank 7 of 8 scatters the whole
    8 Gi text, keeps the last eighth of the groups and sorts them; its slice (all positions, a third of them >= 2^32) is checked
    by the exact slice verifier.  The slices' offsets need every rank's group sizes: ranks 0 .. 6 only run their level A."""
    import torch
    from caps_sa_dist import ShardBuffers
    monkeypatch.delenv("CAPS_SA_SHARD_EXCHANGE", raising=False)
    T, n, world, p = c4_text, C4_N, 8, 8000
    dev = T.device
    reports, bufs, keep = [], None, None
    for r in range(world):
        sh = L.shard(T.data_ptr(), n, p, 64, r, world, 0)
        inf = sh.info()
        assert inf["direct_fallback"] == 0 and inf["exchange"] == 0 and inf["idx_bytes"] == 8
        if bufs is None:
            bufs = ShardBuffers(inf, dev, torch.int64)
        sh.scatter(bufs.send_k.data_ptr(), bufs.send_s.data_ptr(), bufs.report.data_ptr())
        assert sh.info()["level_a_elems"] == n and sh.info()["key_bytes"] == 8
        rep = bufs.report.cpu().numpy().astype(np.uint64)
        assert rep[-2] == 0 and rep[-1] == 0, "pivot ties / overflow on random DNA"
        reports.append(rep)
        if r == world - 1:
            keep = sh
        else:
            sh.close()
    try:
        all_rep = np.stack(reports)
        assert int(all_rep[:, :-2].sum()) == n                    # every suffix kept by exactly one rank
        code, sc, rc = keep.plan(all_rep)
        assert code == 0 and int(sc.sum()) == 0 and int(rc.sum()) == 0
        assert keep.sort_owned(bufs.send_k.data_ptr(), bufs.send_s.data_ptr(), bufs.SA.data_ptr(), bufs.LCP.data_ptr()) == 0
        info = keep.info()
        cnt = info["recv_total"]
        assert info["slice_off"] + cnt == n and abs(cnt - n // world) < n // world // 20
        assert int((bufs.SA[:cnt] >= TWO32).sum().item()) > cnt // 3
        errs = L.verify_slice_device(T.data_ptr(), n, bufs.SA.data_ptr(), bufs.LCP.data_ptr(), cnt, False, idx_bits=64)
        assert errs == 0, f"{errs} violations in rank {world - 1}'s slice"
    finally:
        keep.close()
        del bufs
        torch.cuda.empty_cache()


def test_c4_sharded_direct_path_for_the_last_rank(L, c4_text, monkeypatch):
    """The same in exchange mode (CAPS_SA_SHARD_EXCHANGE=1): every rank's level A over every 8th tile, 32-bit keys, the blocks
    for rank 7 copied as the all-to-all would."""
    import torch
    from caps_sa_dist import ShardBuffers
    monkeypatch.setenv("CAPS_SA_SHARD_EXCHANGE", "1")
    T, n, world, p = c4_text, C4_N, 8, 8000
    dev = T.device
    target = world - 1
    reports, recv_k, recv_s, keep, bufs = [], None, None, None, None
    order = list(range(world - 1)) + [target]                 # the target rank last: its shard stays for plan + sort
    for r in order:
        sh = L.shard(T.data_ptr(), n, p, 64, r, world, 0)
        inf = sh.info()
        assert inf["direct_fallback"] == 0 and inf["idx_bytes"] == 8 and inf["direct_groups"] * inf["direct_sub"] == inf["n_streams"]
        if bufs is None:
            bufs = ShardBuffers(inf, dev, torch.int64)
            block = inf["stream_cap"] * inf["direct_sub"]      # elements of one group's streams
            g_lo, g_hi = target * inf["direct_groups"] // world, inf["direct_groups"]
            per_src = (g_hi - g_lo) * block
            recv_k = torch.empty(per_src * world, dtype=torch.int64, device=dev)
            recv_s = torch.empty(per_src * world, dtype=torch.int64, device=dev)
        sh.scatter(bufs.send_k.data_ptr(), bufs.send_s.data_ptr(), bufs.report.data_ptr())
        rep = bufs.report.cpu().numpy().astype(np.uint64)
        assert rep[-2] == 0 and rep[-1] == 0, "pivot ties / overflow on random DNA"
        assert int(rep[:-2].sum()) == sh.info()["level_a_elems"]
        reports.append((r, rep))
        kt = torch.int32 if sh.info()["key_bytes"] == 4 else torch.int64                        # world 8, 2-bit text: 32-bit keys
        assert kt == torch.int32
        recv_k.view(kt)[r * per_src:(r + 1) * per_src] = bufs.send_k.view(kt)[g_lo * block:g_hi * block]   # what the all-to-all delivers
        recv_s[r * per_src:(r + 1) * per_src] = bufs.send_s[g_lo * block:g_hi * block]
        if r == target:
            keep = sh
        else:
            sh.close()
    try:
        all_rep = np.stack([rep for _, rep in sorted(reports, key=lambda t: t[0])])
        assert int(all_rep[:, :-2].sum()) == n                    # level A distributed every suffix exactly once
        code, sc, rc = keep.plan(all_rep)
        assert code == 0 and int(rc[0]) == per_src and int(sc[target]) == per_src
        assert keep.sort_owned(recv_k.data_ptr(), recv_s.data_ptr(), bufs.SA.data_ptr(), bufs.LCP.data_ptr()) == 0
        info = keep.info()
        cnt = info["recv_total"]
        assert info["slice_off"] + cnt == n and abs(cnt - n // world) < n // world // 20
        assert info["slot_splits"] >= 1 and info["slot_splits_redone"] == 0
        SA = bufs.SA[:cnt]
        assert int((SA >= TWO32).sum().item()) > cnt // 3         # the slice does hold positions beyond 2^32
        errs = L.verify_slice_device(T.data_ptr(), n, bufs.SA.data_ptr(), bufs.LCP.data_ptr(), cnt, False, idx_bits=64)
        assert errs == 0, f"{errs} violations in rank {target}'s slice"
        # a corrupted slice must be caught
        SA[cnt // 2], SA[cnt // 2 + 1] = SA[cnt // 2 + 1].clone(), SA[cnt // 2].clone()
        assert L.verify_slice_device(T.data_ptr(), n, bufs.SA.data_ptr(), bufs.LCP.data_ptr(), cnt, False, idx_bits=64) > 0
    finally:
        keep.close()


def test_c4_samplesort_phase1_of_the_last_rank(L, oracle, c4_text, monkeypatch):
    """shard_phase1 alone at C4 size: rank 7 of 8 sorts subarrays 7000 .. 7999 (text positions >= 7/8 n, all beyond 2^32)."""
    import torch
    monkeypatch.setenv("CAPS_SA_PATH", "classic")
    T, n, world, p = c4_text, C4_N, 8, 8000
    sh = L.shard(T.data_ptr(), n, p, 64, world - 1, world, 0)
    try:
        inf = sh.info()
        sk = torch.empty(max(inf["m_local"], 1), dtype=torch.int64, device="cuda")
        ss = torch.empty(max(inf["m_local"], 1), dtype=torch.int64, device="cuda")
        sh.phase1(sk.data_ptr(), ss.data_ptr())
        cnt, slen = sh.phase1_arrays()
        assert cnt == inf["local_elems"] and slen == n // p
        keys = torch.empty(cnt, dtype=torch.int64, device="cuda")
        sa = torch.empty(cnt, dtype=torch.int64, device="cuda")
        sh.phase1_arrays(keys.data_ptr(), sa.data_ptr())
        lo_pos = (world - 1) * (p // world) * slen
        assert int(sa.min().item()) == lo_pos and int(sa.max().item()) == n - 1 and lo_pos > TWO32
        # every subarray: keys non-decreasing (as unsigned), positions inside the subarray's range, each exactly once
        G = inf["g1"] - inf["g0"]
        ukeys = keys ^ (-(1 << 63))                                # unsigned order through signed compares
        bad = (ukeys[1:] < ukeys[:-1])
        starts = torch.arange(1, G, device="cuda", dtype=torch.int64) * slen
        bad[starts - 1] = False                                    # subarray heads may step down
        assert int(bad.sum().item()) == 0
        sub_of = torch.clamp((sa - lo_pos) // slen, max=G - 1)
        own = torch.clamp(torch.arange(cnt, device="cuda", dtype=torch.int64) // slen, max=G - 1)
        assert bool((sub_of == own).all().item())
        assert int(sa.sum().item()) == (lo_pos + n - 1) * cnt // 2     # with the range check: a permutation of the positions
        Th = T.cpu().numpy()
        for g in (0, G // 2, G - 1):
            a = g * slen
            b = cnt if g == G - 1 else a + slen
            exp, _ = oracle.merge_sort(Th, np.arange(lo_pos + a, lo_pos + b, dtype=np.uint64), idx_bits=64)
            assert np.array_equal(sa[a:b].cpu().numpy().view(np.uint64), exp), f"subarray {inf['g0'] + g}"
    finally:
        sh.close()
