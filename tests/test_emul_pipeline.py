"""CPU: LOGIC of the HIP kernels, run through their host emulation (tests/emul), against
the oracle.  This is a development aid for the GPU-less container; the parity tests proper
are the -m gpu tests in test_gpu_parity.py, which run the real kernels."""
import numpy as np
import pytest

from conftest import LARGE_GOLDEN, large_golden, text_bytes
from emul_util import emul

DNA = np.frombuffer(b"ACGT", dtype=np.uint8)


def _same(E, oracle, T, p, bits=32):
    SA, LCP, st = E.build(T, p=p, idx_bits=bits)
    SAo, LCPo = oracle.naive_sa_lcp(T, idx_bits=bits)
    assert np.array_equal(SA, SAo), f"SA mismatch n={T.size} p={p}"
    assert np.array_equal(LCP, LCPo), f"LCP mismatch n={T.size} p={p}"
    return st


def test_golden_cases(oracle, golden_cases):
    E = emul()
    for c in golden_cases:
        T = text_bytes(c["text"])
        for p in (0, 2, 5):
            SA, LCP, _ = E.build(T, p=p)
            assert SA.tolist() == c["sa"], (c["name"], p)
            assert LCP.tolist() == c["lcp"], (c["name"], p)


@pytest.mark.parametrize("name", LARGE_GOLDEN)
def test_large_golden_cases_take_the_direct_path(name, monkeypatch):
    """Reference-made vectors long enough for the default construction: the direct path (and, separately, the samplesort
    path) of the emulated kernels against them."""
    T, sa, lcp = large_golden(name)
    for path in ("auto", "classic"):
        monkeypatch.setenv("CAPS_SA_PATH", path)
        SA, LCP, st = emul().build(T, p=0)
        assert np.array_equal(SA, sa) and np.array_equal(LCP, lcp), (name, path)
        assert st["path_direct"] == (1 if path == "auto" else 0), (name, path, st["path_fallback"])


@pytest.mark.parametrize("n,p", [(200000, 7), (200001, 0), (300007, 16), (100000, 1), (4096, 2), (4097, 2),
                                 (8191, 3), (8193, 0), (65536 + 17, 4)])
def test_random_dna(oracle, sa_path, n, p):
    rs = np.random.RandomState(n % 1000 + p)
    st = _same(emul(), oracle, rs.choice(DNA, size=n), p)
    assert st["bits_per_char"] == 2


def test_byte_alphabet_and_signed_order(oracle):
    rs = np.random.RandomState(3)
    st = _same(emul(), oracle, rs.choice(np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", dtype=np.uint8), size=150000), 5)
    assert st["bits_per_char"] == 8
    _same(emul(), oracle, rs.choice(np.array([0x41, 0x7F, 0x80, 0xFF, 0], dtype=np.uint8), size=100000), 6)
    _same(emul(), oracle, rs.randint(0, 256, size=50000).astype(np.uint8), 3)


def test_deep_lcp_inputs(oracle):
    rs = np.random.RandomState(4)
    E = emul()
    n = 20000
    SA, LCP, _ = E.build(np.full(n, ord("A"), dtype=np.uint8), p=4)
    assert np.array_equal(SA, np.arange(n - 1, -1, -1, dtype=np.uint32))       # SURVEY 0.8 closed form
    assert np.array_equal(LCP, np.arange(n, dtype=np.uint32))
    _same(E, oracle, np.tile(np.frombuffer(b"AC", dtype=np.uint8), 6000), 3)
    _same(E, oracle, np.tile(rs.choice(DNA, size=37), 500), 9)
    # 2-bit alphabet whose smallest symbol collides with the end-of-text padding code
    _same(E, oracle, rs.choice(np.frombuffer(b"AT", dtype=np.uint8), size=30000, p=[0.9, 0.1]), 4)


def test_u64_indices(oracle):
    rs = np.random.RandomState(5)
    _same(emul(), oracle, rs.choice(DNA, size=120000), 11, bits=64)


def _bounded(E, oracle, T, p, ctx, bits=32):
    SA, LCP, st = E.build(T, p=p, max_context=ctx, idx_bits=bits)
    SAo, LCPo = oracle.build_sa_lcp(T, p=p, max_context=ctx, idx_bits=bits)[:2]
    assert np.array_equal(SA, SAo), f"SA mismatch n={T.size} p={p} ctx={ctx}"
    assert np.array_equal(LCP, LCPo), f"LCP mismatch n={T.size} p={p} ctx={ctx}"
    return st


def test_bounded_context_follows_the_reference_merge_history(oracle):
    """SURVEY f4 (csrc/bounded.h): with 0 < max_context < n ties are broken by the reference's merge order (src/Suffix_Array.cpp:
    71-92), so the whole sequence -- merge_sort's halving, truncated-gap pivots, truncated upper_bound, sort_partition's tree --
    is reproduced, one thread per merge node.  PARITY UNPINNED by the reference (no vector exists): compared with the oracle's
    restatement of the same functions, on texts where ties are everywhere (tiny contexts, unary, periodic, a text repeated
    three times), with 64-bit indices and 8-bit codes."""
    import caps_sa_amd
    E = emul()
    rs = np.random.RandomState(5)
    assert _bounded(E, oracle, rs.choice(DNA, size=5000), 7, 3)["path_fallback"] == 7          # CAPS_SA_FB_BOUNDED
    _bounded(E, oracle, rs.choice(DNA, size=5000), 0, 5)
    _bounded(E, oracle, rs.choice(DNA, size=20001), 13, 8)
    _bounded(E, oracle, rs.choice(DNA, size=20001), 13, 1)
    _bounded(E, oracle, rs.choice(DNA, size=3000), 16, 2, bits=64)
    _bounded(E, oracle, np.tile(rs.choice(DNA, size=37), 200), 9, 20)
    _bounded(E, oracle, np.full(4000, ord("A"), np.uint8), 4, 10)
    _bounded(E, oracle, rs.choice(np.frombuffer(b"abcdefgh\x80\xff", dtype=np.uint8), size=8000), 5, 2)
    _bounded(E, oracle, rs.choice(DNA, size=5000), 7, 4999)
    S = rs.choice(DNA, size=3000)
    _bounded(E, oracle, np.concatenate([S, S, S]), 6, 50)
    _bounded(E, oracle, rs.choice(DNA, size=100), 0, 3)
    T = rs.choice(DNA, size=1000)
    a = E.build(T, max_context=1000)        # >= n is unbounded: the suffix array itself
    b = E.build(T)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(caps_sa_amd.CapsSaError):          # below the reference's domain (n < 32: it divides by zero)
        E.build(T[:20], max_context=3)


def test_bounded_context_passes_an_independent_check(oracle):
    """tests/sa_check.py check_bounded shares no code with csrc/bounded.h or the oracle -- permutation,
    neighbours in order on their first ctx + 1 chars (the char behind the context decides, src/Suffix_Array.cpp:76-77), LCPs between
    min(lcp, ctx) and lcp.  It accepts what the oracle and the kernels produce and catches a swapped pair and a wrong LCP; the
    order of TIES (ctx + 1 equal chars) is the merge history's, which only the oracle comparison above pins."""
    from sa_check import check_bounded
    E = emul()
    rs = np.random.RandomState(9)
    S = rs.choice(DNA, size=2000)
    for T, p, ctx in [(rs.choice(DNA, size=12000), 11, 5), (rs.choice(DNA, size=5000), 0, 1), (np.concatenate([S, S, S]), 6, 40),
                      (np.full(3000, ord("A"), np.uint8), 4, 7), (rs.choice(np.frombuffer(b"ab\x80\xff", dtype=np.uint8), size=9000), 5, 3)]:
        for SA, LCP in (oracle.build_sa_lcp(T, p=p, max_context=ctx)[:2], E.build(T, p=p, max_context=ctx)[:2]):
            r = check_bounded(T, SA, LCP, ctx)
            assert not any(r.values()), (T.size, p, ctx, r)
        j = next((j for j in range(T.size // 3, T.size - 1) if LCP[j + 1] < ctx), None)
        if j is not None:
            S2 = SA.copy()
            S2[j], S2[j + 1] = S2[j + 1], S2[j]
            assert check_bounded(T, S2, LCP, ctx)["order"] >= 1
        L2 = LCP.copy()
        k = int(np.argmax(LCP > 0))
        L2[k] -= 1
        assert check_bounded(T, SA, L2, ctx)["lcp_low"] == 1
        S3 = SA.copy()
        S3[5] = S3[6]
        assert check_bounded(T, S3, LCP, ctx)["not_a_permutation"] >= 1


def test_kernel_level_entry_points(oracle):
    E = emul()
    rs = np.random.RandomState(7)
    T = rs.choice(DNA, size=60000)
    idx = rs.permutation(60000)[:25000].astype(np.uint32)
    # a4 merge_sort
    sa, lcp = E.sort_suffixes(T, idx)
    so, lo = oracle.merge_sort(T, idx)
    assert np.array_equal(sa, so) and np.array_equal(lcp, lo)
    # a3 merge, ragged lengths
    xa, xl = oracle.merge_sort(T, idx[:9000])
    ya, yl = oracle.merge_sort(T, idx[9000:])
    Z, LZ = E.merge(T, xa, ya, xl, yl)
    Zo, LZo = oracle.merge(T, xa, ya, xl, yl)
    assert np.array_equal(Z, Zo) and np.array_equal(LZ, LZo)
    Z, LZ = E.merge(T, xa, ya[:0], xl, yl[:0])
    assert np.array_equal(Z, xa) and np.array_equal(LZ, xl)
    # a7 upper_bound (members and non-members of the list)
    piv = np.concatenate([sa[::997], rs.randint(0, 60000, size=50).astype(np.uint32)])
    ub = E.upper_bound(T, sa, piv)
    for pv, u in zip(piv.tolist(), ub.tolist()):
        assert u == oracle.upper_bound(T, sa, pv)
    # a2 LCP
    a = rs.randint(0, 60000, size=500).astype(np.uint32)
    b = rs.randint(0, 60000, size=500).astype(np.uint32)
    out = E.lcp(T, a, b)
    for x, y, l in zip(a.tolist(), b.tolist(), out.tolist()):
        assert l == oracle.lcp(T, x, y)


def _check_segments(LIB, oracle, T, idx, seg):
    sa, lcp = LIB.sort_segments(T, idx, seg)
    prev_last = None
    for g in range(len(seg) - 1):
        a, b = int(seg[g]), int(seg[g + 1])
        if a == b:
            continue
        so, lo = oracle.merge_sort(T, idx[a:b])
        assert np.array_equal(sa[a:b], so), f"segment {g}"
        exp = lo.copy()
        exp[0] = 0 if prev_last is None else oracle.lcp(T, prev_last, int(so[0]))
        assert np.array_equal(lcp[a:b], exp), f"segment {g} lcp"
        prev_last = int(so[-1])


def test_segmented_sort_mixed_lengths(oracle):
    """Segments of 0, 1, <1 tile, exactly 1/2/3/5/8 tiles (+-1): finished segments sit out later
    passes and end in different ping-pong buffers (parity of passes_for(len))."""
    rs = np.random.RandomState(8)
    T = rs.choice(DNA, size=200000)
    lens = [5000, 0, 1, 4096, 4097, 0, 8192, 8193, 12288, 3, 20480, 20481, 32768, 100, 0, 16385, 7]
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    idx = rs.permutation(200000)[:int(seg[-1])].astype(np.uint32)
    _check_segments(emul(), oracle, T, idx, seg)


def test_uniform_keys_take_the_bucket_sort_fast_path(oracle, sa_path):
    """Regression guard for the bucket maps: on random DNA every tile of the big sorts (two on the samplesort
    path, one on the direct path) must be sorted by the in-LDS bucket sort (known key range, no bin overflow),
    with no merge pass."""
    import ctypes
    E = emul()
    f = E.dll.caps_sa_emul_tile_stats
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    a = (ctypes.c_uint64 * 4)()
    f(a, 1)
    T = np.random.RandomState(31).choice(DNA, size=1_500_000)
    SA, LCP, st = E.build(T, p=20)
    f(a, 1)
    slow_unknown, fast_unknown, slow_known, fast_known = list(a)
    assert slow_known == 0 and slow_unknown == 0, list(a)
    assert st["path_direct"] == (0 if sa_path == "classic" else 1), st
    assert fast_known > (700 if sa_path == "classic" else 350)
    if sa_path != "classic":
        assert st["slot_splits"] >= 1 and st["slot_splits_redone"] == 0 and st["direct_groups"] >= 2
    assert st["merge_passes_phase1"] == 0 and st["merge_passes_phase2"] == 0
    SAo, LCPo = oracle.build_sa_lcp(T, p=20)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)


# ---- the same kernel sources with 256-element tiles: many tiles, merge passes, partitions that
# ---- span hundreds of runs, at sizes the naive oracle still checks in a blink
def _skewed(rs, n):
    """Markov-ish DNA with long poly-A stretches and a planted repeat: leaves every interpolation fast path."""
    T = rs.choice(DNA, size=n, p=[0.55, 0.15, 0.15, 0.15])
    T[n // 3:n // 3 + n // 20] = ord("A")
    rep = T[100:100 + n // 10].copy()
    T[n // 2:n // 2 + rep.size] = rep
    return T


@pytest.mark.parametrize("n,p", [(5000, 3), (40000, 300), (60000, 7), (200000, 400), (400000, 1000), (30011, 0), (257, 2),
                                 (256 * 9 + 1, 9)])
def test_small_tiles_random_dna(oracle, sa_path, n, p):
    from emul_util import emul_small
    rs = np.random.RandomState(n % 97 + p)
    T = rs.choice(DNA, size=n)
    SA, LCP, st = emul_small().build(T, p=p)
    SAo, LCPo = oracle.build_sa_lcp(T, p=p)[:2]
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)


@pytest.mark.parametrize("n,p", [(20000, 5), (50000, 64), (70000, 300), (60000, 1200)])
def test_small_tiles_skewed_and_repetitive(oracle, sa_path, n, p):
    from emul_util import emul_small
    rs = np.random.RandomState(n % 89 + p)
    E = emul_small()
    for T in (_skewed(rs, n), np.tile(rs.choice(DNA, size=61), n // 61 + 1)[:n],
              rs.choice(np.frombuffer(b"abcdefgh\x80\xff", dtype=np.uint8), size=n, p=[0.4] + [0.6 / 9] * 9)):
        SA, LCP, st = E.build(T, p=p)
        SAo, LCPo = oracle.build_sa_lcp(T, p=p)[:2]
        assert np.array_equal(SA, SAo), (n, p)
        assert np.array_equal(LCP, LCPo), (n, p)


def test_small_tiles_u64_and_segments(oracle, sa_path):
    from emul_util import emul_small
    rs = np.random.RandomState(77)
    E = emul_small()
    T = _skewed(rs, 70000)
    SA, LCP, _ = E.build(T, p=333, idx_bits=64)
    SAo, LCPo = oracle.build_sa_lcp(T, p=333, idx_bits=64)[:2]
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    lens = [300, 0, 1, 256, 257, 0, 512, 513, 768, 3, 1280, 1281, 2048, 100, 0, 1025, 7, 5000]
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    idx = rs.permutation(70000)[:int(seg[-1])].astype(np.uint32)
    _check_segments(E, oracle, T, idx, seg)


def test_host_entry_point_reuses_and_regrows_its_device_block(oracle):
    """caps_sa_*_build_* keep one grow-only device block between calls (capi_impl.h: HostPathCache)."""
    E = emul()
    rs = np.random.RandomState(12)
    for n, p, bits in [(50000, 4, 32), (3000, 2, 32), (120000, 9, 64), (70000, 0, 32)]:
        _same(E, oracle, rs.choice(DNA, size=n), p, bits=bits)
    E.release_cache()
    E.release_cache()                                 # idempotent
    _same(E, oracle, rs.choice(DNA, size=20000), 3)
    a = E.pinned_empty(1000, np.uint32)               # host_alloc / host_free round trip
    a[:] = 7
    assert int(a.sum()) == 7000


def test_equalised_split_never_needs_more_passes_than_the_plain_one(oracle, monkeypatch):
    """Redo path of the bucket split (slots overflowed): the count pass fills 16x finer buckets and
    bucket_group_kernel packs them into the bucket slots.  Same result; never a larger largest bucket
    (= never more LCP-merge passes) than the plain equal-key-range split."""
    from emul_util import emul_small
    rs = np.random.RandomState(13)
    texts = [rs.choice(DNA, size=150000, p=[0.7, 0.2, 0.08, 0.02]),
             rs.choice(np.frombuffer(b"abcdefgh", dtype=np.uint8), size=100000, p=[.5, .2, .1, .1, .05, .03, .01, .01])]
    # order-2 Markov chain with skewed transitions
    trans = rs.dirichlet([0.3] * 4, size=16)
    m = np.zeros(120000, dtype=np.int64)
    u = rs.rand(m.size)
    for i in range(2, m.size):
        m[i] = min(3, int(np.searchsorted(np.cumsum(trans[m[i - 2] * 4 + m[i - 1]]), u[i])))
    texts.append(DNA[m])
    monkeypatch.setenv("CAPS_SA_PATH", "classic")        # the pass counts below are those of the samplesort path's two sorts
    for E in (emul_small(), emul()):
        for T in texts:
            for p in (4, 37):
                monkeypatch.delenv("CAPS_SA_NO_EQUALISE", raising=False)
                SA, LCP, st = E.build(T, p=p)
                monkeypatch.setenv("CAPS_SA_NO_EQUALISE", "1")
                SA0, LCP0, st0 = E.build(T, p=p)
                monkeypatch.delenv("CAPS_SA_NO_EQUALISE", raising=False)
                assert np.array_equal(SA, SA0) and np.array_equal(LCP, LCP0)
                SAo, LCPo = oracle.build_sa_lcp(T, p=p)
                assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
                assert st["merge_passes_phase1"] <= st0["merge_passes_phase1"]
                assert st["merge_passes_phase2"] <= st0["merge_passes_phase2"]


def test_one_process_several_devices(oracle, monkeypatch):
    """caps_sa_*_build_multi_* (capi_impl.h build_multi): one Shard per listed device, blocks copied device to device,
    slices copied into the caller's arrays; here every "device" is the emulation's host memory."""
    from emul_util import emul_small
    E = emul_small()
    rs = np.random.RandomState(19)
    # groups that do not divide evenly among the ranks (3 or 7 groups on 8, 4 or 2 ranks: some ranks own none), with and
    # without the exchange -- the exchange layout has up to world * SUB more segments than K1 * SUB (tools/stress_gpu.py found
    # the tables one rank short of that)
    for exchange in ("1", None):
        if exchange:
            monkeypatch.setenv("CAPS_SA_SHARD_EXCHANGE", exchange)
        else:
            monkeypatch.delenv("CAPS_SA_SHARD_EXCHANGE", raising=False)
        for sub in ("8", None):
            if sub:
                monkeypatch.setenv("CAPS_SA_DIRECT_SUB", sub)
            else:
                monkeypatch.delenv("CAPS_SA_DIRECT_SUB", raising=False)
            for devs, n, p, bits in [([0] * 8, 60_000, 3, 32), ([0] * 4, 50_000, 7, 64), ([0] * 2, 40_000, 3, 32)]:
                T = rs.choice(DNA, size=n)
                SA, LCP, st = E.build_multi(T, devs, p=p, idx_bits=bits)
                SAo, LCPo = oracle.naive_sa_lcp(T, idx_bits=bits)
                assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (exchange, sub, devs, n, p)
    monkeypatch.delenv("CAPS_SA_DIRECT_SUB", raising=False)
    for devs, n, p, bits in [([0, 0], 60_000, 0, 32), ([0, 0, 0], 90_001, 700, 32), ([0] * 8, 150_000, 0, 64), ([0], 40_000, 0, 32),
                             ([0, 0], 3_000, 0, 32)]:
        T = rs.choice(DNA, size=n)
        SA, LCP, st = E.build_multi(T, devs, p=p, idx_bits=bits)
        SAo, LCPo = oracle.naive_sa_lcp(T, idx_bits=bits)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (devs, n)
        assert st["path_direct"] == (1 if n >= 8192 else 0)
    T = rs.choice(DNA, size=120_000)
    T[30_000:36_000] = ord("G")                           # a long run: the ranks agree to leave the direct path; devices[0] builds alone
    SA, LCP, st = E.build_multi(T, [0, 0, 0], p=0)
    SAo, LCPo = oracle.build_sa_lcp(T, p=64)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo) and st["path_fallback"] != 0
    # tandem arrays: the groups of equal keys are deferred and re-keyed on the device that owns them (shard.h sort_owned)
    from sa_check import sa_lcp
    T = rs.choice(DNA, size=160_000)
    for at, unit, copies, rate in [(30_000, 23, 700, 0.003), (90_000, 57, 150, 0.02)]:
        seg = np.tile(rs.choice(DNA, size=unit), copies)
        mut = rs.rand(seg.size) < rate
        seg[mut] = rs.choice(DNA, size=int(mut.sum()))
        T[at:at + seg.size] = seg
    SAo, LCPo = sa_lcp(T)
    SA, LCP, st = E.build_multi(T, [0, 0], p=0)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    assert st["path_direct"] == 1 and st["n_devices"] == 2 and st["tie_groups_deferred"] > 0, st["tie_groups_deferred"]


def test_cli_validates_before_it_writes(tmp_path):
    """arguments are checked and the build runs BEFORE the output file is opened; a bounded context (unsupported)
    or a malformed count leaves no empty file behind."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "caps-sa_amd"), "caps_sa"])
    exe = os.path.join(root, "caps-sa_amd", "caps_sa")
    inp = tmp_path / "in.txt"
    inp.write_bytes(b"ACGT" * 100)
    for extra in (["8", "50"], ["x8"], ["8", "-1"]):
        out = tmp_path / "out.bin"
        r = subprocess.run([exe, str(inp), str(out)] + extra, capture_output=True, text=True)
        assert r.returncode != 0 and not out.exists(), (extra, r.stderr)
    r = subprocess.run([exe, str(inp)], capture_output=True, text=True)
    assert r.returncode != 0 and "bounded-context" in r.stderr


def _markov(rs, n, order=3, skew=0.3):
    trans = rs.dirichlet([skew] * 4, size=4 ** order)
    cdf = np.cumsum(trans, 1)
    m = np.zeros(n, dtype=np.int64)
    u = rs.rand(n)
    st = 0
    for i in range(n):
        c = min(3, int(np.searchsorted(cdf[st], u[i])))
        m[i] = c
        st = (st * 4 + c) % (4 ** order)
    return DNA[m]


def test_direct_path_modes_and_fallbacks(oracle, monkeypatch):
    """The direct path (pipeline.h run_direct) in both level-B modes, chosen by the pivots and forced: linear (interpolated
    buckets in slots) and quantile (sample-quantile buckets, count + exact scatter); frequent keys (N-block stand-ins) with
    regions sized by expected bucket-worth; a text that is ONE key (samplesort path).  256-element-tile build."""
    from emul_util import emul_small
    E = emul_small()
    rs = np.random.RandomState(5)

    def run(T, mode=None, sub=None, bits=32):
        for k, v in (("CAPS_SA_DIRECT_MODE", mode), ("CAPS_SA_DIRECT_SUB", sub)):
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        SA, LCP, st = E.build(T, p=0, idx_bits=bits)
        SAo, LCPo = oracle.build_sa_lcp(T, p=64, idx_bits=bits)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
        return st

    T1 = _markov(rs, 200_000)
    st = run(T1)
    assert st["path_direct"] == 1 and st["direct_quantile"] == 1          # the pivots reveal the skew
    st_lin = run(T1, mode="linear")
    assert st_lin["direct_quantile"] == 0 and st_lin["merge_passes_phase2"] >= st["merge_passes_phase2"]
    uni = rs.choice(DNA, size=150_000)
    assert run(uni, mode="quantile")["direct_quantile"] == 1
    assert run(uni[:100_001], mode="quantile", bits=64)["direct_quantile"] == 1
    assert run(uni, mode="linear")["path_direct"] == 1
    T2 = rs.choice(DNA, size=300_000)
    T2[50_000:53_000] = ord("G")
    T2[200_000:200_700] = ord("G")
    st = run(T2, sub="1")                                                   # one stream per group: the fat group's region holds the run
    assert st["path_direct"] == 1 and st["direct_quantile"] == 1 and st["long_runs"] == 1
    run(T2)                                                                 # 8 sub-streams at this tiny size: a 3-tile run overloads 3 of them -> falls back, same result
    T3 = _markov(rs, 250_000)
    T3[100_000:104_000] = ord("G")
    T3[10_000:12_000] = T3[150_000:152_000]
    run(T3, sub="1")                                                        # skew + N-block + repeat: whichever path it takes, same result
    st = run(np.full(60_000, ord("A"), dtype=np.uint8))
    assert st["path_direct"] == 0 and st["path_fallback"] == 4             # CAPS_SA_FB_PIVOT_TIES: one key covers the text
    T5 = rs.choice(np.frombuffer(b"abcdefgh", dtype=np.uint8), size=120_000, p=[.5, .2, .1, .1, .05, .03, .01, .01])
    assert run(T5)["bits_per_char"] == 8


def test_quantile_split_without_a_count_pass(oracle, monkeypatch):
    """Level B in quantile mode (pipeline.h "speculative split by knots"): no count pass -- every bucket gets a slot, the runs that
    do not fit go to a stream (bucket_scatter_kernel SPILL) that borrows the caller's SA / LCP slice, the buckets that outgrew
    their slots are put together in the compact array (spill_gather_kernel, spill_place_kernel) and the tile sort reads a bucket
    from its slot or from there.  Skewed keys, N-block stand-ins (letter-run buckets: always from the compact array; runs of a
    tile long enough for the cooperative write), repeats with deferred ties, 64-bit indices, waves; slots made tiny (most of
    every bucket takes the stream) and a stream that runs full (the count split takes over).  256-element-tile build."""
    from emul_util import emul_small
    E = emul_small()
    rs = np.random.RandomState(5)
    keys = ("CAPS_SA_DIRECT_MODE", "CAPS_SA_DIRECT_SUB", "CAPS_SA_TEST_SPILL_SLOT", "CAPS_SA_TEST_SPILL_CAP", "CAPS_SA_NO_SPILL_SLOTS",
            "CAPS_SA_HOST_WAVES")

    def run(T, bits=32, **env):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv("CAPS_SA_" + k, v)
        SA, LCP, st = E.build(T, p=0, idx_bits=bits)
        SAo, LCPo = oracle.build_sa_lcp(T, p=64, idx_bits=bits)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), env
        return st

    skew = _markov(rs, 200_000)
    st = run(skew)
    assert st["direct_quantile"] == 1 and st["knot_slot_splits"] == 1 and st["knot_slot_splits_redone"] == 0
    assert 0 < st["spill_entries"] < skew.size // 8 and st["bucket_count_ms"] == 0.0
    base = st["spill_entries"]
    st = run(skew, NO_SPILL_SLOTS="1")
    assert st["knot_slot_splits"] == 0 and st["spill_entries"] == 0
    st = run(skew, TEST_SPILL_SLOT="160")                                   # slots below the mean bucket (192): most buckets outgrow them
    assert st["knot_slot_splits"] == 1 and st["spill_entries"] > 4 * base
    st = run(skew, TEST_SPILL_SLOT="32")                                    # ... so far below that the stream runs full
    assert st["knot_slot_splits"] == 0 and st["knot_slot_splits_redone"] == 1
    st = run(skew, TEST_SPILL_CAP="100")
    assert st["knot_slot_splits"] == 0 and st["knot_slot_splits_redone"] == 1
    st = run(skew[:120_001], bits=64)
    assert st["knot_slot_splits"] == 1 and st["spill_entries"] > 0
    st = run(skew[:120_001], bits=64, TEST_SPILL_SLOT="160")
    assert st["knot_slot_splits"] == 1
    for waves in ("2", "5"):
        st = run(skew, HOST_WAVES=waves)
        assert st["result_waves"] >= 2 and st["knot_slot_splits"] == st["result_waves"], st
        st = run(skew, HOST_WAVES=waves, TEST_SPILL_SLOT="160")
        assert st["knot_slot_splits"] == st["result_waves"]
    runs = rs.choice(DNA, size=300_000)
    runs[50_000:53_000] = ord("G")
    runs[200_000:200_700] = ord("G")
    st = run(runs, DIRECT_SUB="1")
    assert st["direct_quantile"] == 1 and st["long_runs"] == 1 and st["run_buckets"] >= 1 and st["knot_slot_splits"] == 1
    assert st["spill_entries"] > 2_500                                       # the G-block: one bucket, far beyond its slot
    st = run(runs, DIRECT_SUB="1", HOST_WAVES="3")
    assert st["knot_slot_splits"] >= 2
    rep = _repeat_rich(rs, 260_000)
    st = run(rep, DIRECT_MODE="quantile")
    assert st["knot_slot_splits"] == 1 and st["tie_groups_deferred"] > 0
    st = run(rep, DIRECT_MODE="quantile", TEST_SPILL_SLOT="160")
    assert st["knot_slot_splits"] == 1
    T8 = rs.choice(np.frombuffer(b"abcdefgh", dtype=np.uint8), size=150_000, p=[.5, .2, .1, .1, .05, .03, .01, .01])
    st = run(T8, DIRECT_MODE="quantile")
    assert st["bits_per_char"] == 8 and (st["path_direct"] == 0 or st["knot_slot_splits"] == 1)
    # uniform keys in quantile mode: the queue chain starts at tile_sort_kernel (linear bins), which sees the tiles of outgrown
    # buckets empty and passes them on like the plain equalised build behind it
    uni = rs.choice(DNA, size=150_000)
    st = run(uni, DIRECT_MODE="quantile")
    assert st["direct_quantile"] == 1 and st["knot_slot_splits"] == 1
    st = run(uni, DIRECT_MODE="quantile", TEST_SPILL_SLOT="160")
    assert st["knot_slot_splits"] == 1 and st["spill_entries"] > 10_000


def test_forced_substreams_on_a_nine_tile_text(oracle, monkeypatch):
    """Found by tools/stress_gpu.py on the GPU (and replayed here): quantile mode with 8 sub-streams forced on a text of nine
    level-A tiles -- the token room per stream exceeded the budget per stream, the regions overran the buffer, SA was not a
    permutation.  The attempt is now refused (shape) and the samplesort path builds it."""
    rs = np.random.RandomState(1)
    T = rs.choice(DNA, size=142_867)
    SAo, LCPo = oracle.build_sa_lcp(T, p=64)
    for mode in ("quantile", "linear"):
        for sub in ("8", "2", "1"):
            monkeypatch.setenv("CAPS_SA_DIRECT_MODE", mode)
            monkeypatch.setenv("CAPS_SA_DIRECT_SUB", sub)
            SA, LCP, st = emul().build(T, p=50)
            assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (mode, sub, st)


def test_32_bit_keys_opt_in(oracle, monkeypatch):
    """CAPS_SA_KEYS=32: the direct path's elements carry key32_of(key, group shift) (text.h) through level A, level B and the
    tile sort; order and LCPs stay exact (prefix-preserving keys, ties settled by the text), slot overflow falls back to 64."""
    rs = np.random.RandomState(2)
    monkeypatch.setenv("CAPS_SA_DIRECT_MODE", "linear")
    planted = rs.choice(DNA, size=500_000)
    for _ in range(40):                                    # repeats of 18-40 bases: ties at 32-bit, not at 64-bit keys
        a, b, ln = rs.randint(0, 499_000), rs.randint(0, 499_000), rs.randint(18, 40)
        planted[b:b + ln] = planted[a:a + ln]
    for T, p, bits in [(rs.choice(DNA, size=1_500_000), 20, 32), (rs.choice(DNA, size=400_000), 50, 32),
                       (rs.choice(DNA, size=600_001), 0, 64), (planted, 30, 32)]:
        SAo, LCPo = oracle.build_sa_lcp(T, p=64, idx_bits=bits)
        for keys, expect in (("32", 32), ("64", 64)):
            monkeypatch.setenv("CAPS_SA_KEYS", keys)
            SA, LCP, st = emul().build(T, p=p, idx_bits=bits)
            assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (T.size, keys)
            assert st["path_direct"] == 1 and st["direct_key_bits"] == expect, st


def test_equalised_tile_sort_settles_ties_once(oracle):
    """tile_sort_eq_kernel on skewed keys with planted repeats: pairs of equal keys (the common case: the member with the
    higher slot compares, its partner picks the outcome up), groups of three and more (every member compares with each of
    the others), ties deeper than the bounded scan (the tile goes on to the comparison sort), and the lcps that the tie
    scans hand to the emit phase.  The counters say that the equalised kernel did finish tiles with such ties."""
    import ctypes
    from emul_util import emul_small
    for E, n in ((emul(), 400_000), (emul_small(), 60_000)):
        f = E.dll.caps_sa_emul_tile_stats8
        f.argtypes = [ctypes.c_void_p, ctypes.c_int]
        a = (ctypes.c_uint64 * 8)()
        rs = np.random.RandomState(n % 97)
        # order-2 chain with skewed transitions: far from uniform inside every key range
        trans = rs.dirichlet([0.4] * 4, size=16)
        c = rs.randint(0, 4, size=2).tolist()
        u = rs.rand(n)
        cdf = np.cumsum(trans, axis=1)
        for i in range(2, n):
            c.append(int(min(3, np.searchsorted(cdf[c[-2] * 4 + c[-1]], u[i]))))
        T = DNA[np.array(c)]
        L = 90
        for _ in range(n // 2000):                           # pairs: a copy with one mutation somewhere
            s, d = rs.randint(0, n - L, size=2)
            T[d:d + L] = T[s:s + L]
            T[d + rs.randint(40, L)] = DNA[rs.randint(0, 4)]
        for _ in range(n // 8000):                           # groups of three and four equal 32-mers
            s = rs.randint(0, n - L)
            for d in rs.randint(0, n - L, size=rs.randint(2, 4)):
                T[d:d + L] = T[s:s + L]
        for _ in range(n // 20000):                          # frequent keys: 8 .. 13 copies of a 36-mer (the first window behind
            s = rs.randint(0, n - L)                         # the key tells them apart) and of a 90-mer (it does not)
            for d in rs.randint(0, n - L, size=rs.randint(8, 14)):
                T[d:d + 36] = T[s:s + 36]
            s = rs.randint(0, n - L)
            for d in rs.randint(0, n - L, size=rs.randint(8, 14)):
                T[d:d + L] = T[s:s + L]
        s, d = n // 5, n // 2                                 # one tie far deeper than TIE_WINDOWS windows
        T[d:d + 3000] = T[s:s + 3000]
        f(a, 1)
        SA, LCP, st = E.build(T, p=24)
        f(a, 1)
        SAo, LCPo = oracle.build_sa_lcp(T, p=24)
        assert np.array_equal(SA, SAo), "SA"
        assert np.array_equal(LCP, LCPo), "LCP"
        gave_up, finished = a[6], a[7]
        assert finished > 0, list(a)
        assert int((LCP >= 40).sum()) > n // 100 and int(LCP.max()) >= 2999


def test_lcp_leaves_as_bytes_through_the_host_path(oracle, monkeypatch):
    """The host-buffer entry point's LCP-as-bytes transfer (capi_impl.h HostCopySink, lcp_narrow_kernel) in the emulation: no
    value above 254, some (single-letter runs), more than the list of exceptions holds (full width again)."""
    from sa_check import sa_lcp
    rs = np.random.RandomState(5)
    E = emul()
    uni = rs.choice(DNA, size=400_000)
    runs = rs.choice(DNA, size=400_000)
    runs[100_000:103_000] = ord("G")
    runs[300_000:300_900] = ord("A")
    flood = rs.choice(DNA, size=400_000)
    flood[100_000:130_000] = ord("G")
    monkeypatch.setenv("CAPS_SA_HOST_NARROW_LCP", "1")
    for name, T, want in (("uniform", uni, 1), ("runs", runs, 1), ("flood", flood, 4)):
        SAo, LCPo = sa_lcp(T, 32)
        for waves in ("1", "3", "5"):
            monkeypatch.setenv("CAPS_SA_HOST_WAVES", waves)
            SA, LCP, st = E.build(T, p=100)
            assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (name, waves)
            assert st["lcp_bytes_on_link"] == want, (name, waves, st["lcp_bytes_on_link"])


def test_a_lost_tie_note_sends_the_tile_to_the_comparison_sort(oracle, monkeypatch):
    """a tie note that gets lost must not become a wrong LCP.  CAPS_SA_TEST_DROP_NOTE makes tile_sort_eq_kernel lose
    the notes of its tied pairs; its emit phase then meets equal keys without a note and hands the tile to the comparison sort (the
    counters: tiles there that were not before), and the arrays are still THE arrays."""
    import ctypes
    E = emul()
    f = E.dll.caps_sa_emul_tile_stats8
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    a = (ctypes.c_uint64 * 8)()
    rs = np.random.RandomState(77)
    n = 300_000
    T = rs.choice(DNA, size=n, p=[0.4, 0.1, 0.15, 0.35])
    for _ in range(n // 1500):                               # pairs of equal 32-mers: a copy with one mutation
        s, d = rs.randint(0, n - 90, size=2)
        T[d:d + 90] = T[s:s + 90]
        T[d + rs.randint(40, 90)] = DNA[rs.randint(0, 4)]
    SAo, LCPo = oracle.build_sa_lcp(T, p=24)
    seen = []
    for drop in (False, True):
        if drop:
            monkeypatch.setenv("CAPS_SA_TEST_DROP_NOTE", "1")
        f(a, 1)
        SA, LCP, st = E.build(T, p=24)
        f(a, 1)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), drop
        seen.append((int(a[4] + a[5]), int(a[7])))           # tiles of the comparison sort, tiles the equalised kernel finished
    assert seen[0][1] > 0 and seen[1][0] > seen[0][0] + 10, seen


def test_phases_do_not_depend_on_the_order_of_their_threads(oracle, monkeypatch):
    """the emulation runs the threads of a phase one after the other, so it cannot see a missing barrier -- unless
    the order changes the outcome.  The same sources compiled with the threads of every phase in DESCENDING order
    (kernel_lang.h CAPS_EMUL_REVERSE; both tile geometries) must still produce THE suffix and LCP arrays: uniform and skewed
    keys, key ties settled by the workgroup (pairs, groups, deep ones), long runs, byte alphabets, both constructions, the
    one-process multi-rank build with and without the exchange."""
    from emul_util import emul_rev
    rs = np.random.RandomState(3)

    def ties(n):
        trans = rs.dirichlet([0.4] * 4, size=16)
        cdf = np.cumsum(trans, axis=1)
        c = rs.randint(0, 4, size=2).tolist()
        u = rs.rand(n)
        for i in range(2, n):
            c.append(int(min(3, np.searchsorted(cdf[c[-2] * 4 + c[-1]], u[i]))))
        T = DNA[np.array(c)]
        for _ in range(n // 1500):
            s, d = rs.randint(0, n - 90, size=2)
            T[d:d + 90] = T[s:s + 90]
            T[d + rs.randint(40, 90)] = DNA[rs.randint(0, 4)]
        for _ in range(n // 6000):
            s = rs.randint(0, n - 90)
            for d in rs.randint(0, n - 90, size=rs.randint(2, 9)):
                T[d:d + 40] = T[s:s + 40]
        T[n // 2:n // 2 + 2500] = T[n // 5:n // 5 + 2500]
        return T

    for small in (True, False):
        E = emul_rev(small)
        n_big = 60_000 if small else 300_000
        cases = [(rs.choice(DNA, size=n_big), 0, 32), (rs.choice(DNA, size=n_big // 2 + 1), 12, 64),
                 (rs.choice(DNA, size=n_big, p=[0.6, 0.2, 0.1, 0.1]), 7, 32), (ties(n_big), 24, 32),
                 (rs.choice(np.frombuffer(b"abcdefgh\xf0", dtype=np.uint8), size=n_big // 2), 16, 32), (rs.choice(DNA, size=300), 0, 32)]
        runs = rs.choice(DNA, size=n_big // 2)
        runs[1000:7000] = ord("G")
        cases.append((runs, 0, 32))
        for T, p, bits in cases:
            SAo, LCPo = oracle.build_sa_lcp(T, p=p, idx_bits=bits)
            for path in ("auto", "classic"):
                monkeypatch.setenv("CAPS_SA_PATH", path)
                SA, LCP, st = E.build(T, p=p, idx_bits=bits)
                assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (small, T.size, p, bits, path)
        monkeypatch.setenv("CAPS_SA_PATH", "auto")
        for exchange in ("1", None):
            if exchange:
                monkeypatch.setenv("CAPS_SA_SHARD_EXCHANGE", exchange)
            else:
                monkeypatch.delenv("CAPS_SA_SHARD_EXCHANGE", raising=False)
            T = rs.choice(DNA, size=n_big)
            SAo, LCPo = oracle.build_sa_lcp(T, p=16)
            SA, LCP, st = E.build_multi(T, [0, 0, 0], p=16)
            assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (small, exchange)


def test_a_stream_that_outgrows_its_region_sends_the_build_to_the_samplesort_path(oracle, monkeypatch):
    """CAPS_SA_FB_GROUP_OVERFLOW (pipeline.h run_direct, shard.h plan()): level A writes every group into a fixed region; a stream
    that outgrows its region voids the attempt and the samplesort path builds the text.  A real text gets there through one
    key that holds several per cent of it; CAPS_SA_TEST_STREAM_CAP (regions of 80 % of the mean stream) forces it on any text.
    One device, and several ranks that must take the way out TOGETHER."""
    from emul_util import emul_small
    E = emul_small()
    rs = np.random.RandomState(23)
    T = rs.choice(DNA, size=150_000)
    SAo, LCPo = oracle.build_sa_lcp(T, p=64)
    assert E.build(T, p=0)[2]["path_direct"] == 1
    monkeypatch.setenv("CAPS_SA_TEST_STREAM_CAP", "80")
    for mode in ("linear", "quantile"):
        monkeypatch.setenv("CAPS_SA_DIRECT_MODE", mode)
        SA, LCP, st = E.build(T, p=0)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), mode
        assert st["path_direct"] == 0 and st["path_fallback"] == 5, (mode, st["path_fallback"])
    monkeypatch.delenv("CAPS_SA_DIRECT_MODE")
    for exchange in (None, "1"):
        if exchange:
            monkeypatch.setenv("CAPS_SA_SHARD_EXCHANGE", exchange)
        SA, LCP, st = E.build_multi(T, [0, 0, 0], p=0)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), exchange
        assert st["path_direct"] == 0 and st["path_fallback"] == 5, (exchange, st["path_fallback"])
    monkeypatch.delenv("CAPS_SA_SHARD_EXCHANGE")
    monkeypatch.delenv("CAPS_SA_TEST_STREAM_CAP")
    assert E.build(T, p=0)[2]["path_direct"] == 1


def test_results_leave_in_waves_on_small_texts(oracle, monkeypatch):
    """The host-buffer entry point streams finished slices of SA / LCP out while later groups are sorted (capi_impl.h
    HostCopySink, pipeline.h set_waves): per-wave segment tables, knots offset by the wave's first group, the LCP at every
    wave's first entry (wave_head_lcp_kernel), work arrays in the scratch + B.  The default only does that from 400 Mi chars on;
    CAPS_SA_HOST_WAVES forces it here: uniform keys (linear buckets), skewed keys (quantile buckets), N-block stand-ins
    (letter-run buckets inside a wave), 64-bit indices, and a group larger than the scratch (then: one wave)."""
    from emul_util import emul_small
    E = emul_small()
    rs = np.random.RandomState(31)
    uni = rs.choice(DNA, size=200_000)
    skew = _markov(rs, 180_000)
    runs = rs.choice(DNA, size=300_000)
    runs[50_000:53_000] = ord("G")
    runs[200_000:200_700] = ord("G")
    cases = [("uniform", uni, 32, None), ("skewed", skew, 32, None), ("runs", runs, 32, "1"), ("uniform64", uni[:120_001], 64, None)]
    for name, T, bits, sub in cases:
        if sub:
            monkeypatch.setenv("CAPS_SA_DIRECT_SUB", sub)
        else:
            monkeypatch.delenv("CAPS_SA_DIRECT_SUB", raising=False)
        SAo, LCPo = oracle.build_sa_lcp(T, p=64, idx_bits=bits)
        for waves in ("2", "3", "5"):
            monkeypatch.setenv("CAPS_SA_HOST_WAVES", waves)
            SA, LCP, st = E.build(T, p=0, idx_bits=bits)
            assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (name, waves)
            assert st["path_direct"] == 1 and 2 <= st["result_waves"] <= int(waves) + 1, (name, waves, st["result_waves"])
        if name == "skewed":
            assert st["direct_quantile"] == 1
        if name == "runs":
            assert st["direct_quantile"] == 1 and st["long_runs"] == 1
    # one group holds a fifth of the text (a 60,000-char run): more than the scratch of one of twelve waves -> one wave, same result
    big = rs.choice(DNA, size=300_000)
    big[100_000:160_000] = ord("G")
    monkeypatch.setenv("CAPS_SA_DIRECT_SUB", "1")
    monkeypatch.setenv("CAPS_SA_HOST_WAVES", "12")
    SA, LCP, st = E.build(big, p=0)
    SAo, LCPo = oracle.build_sa_lcp(big, p=64)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    assert st["path_direct"] == 0 or st["result_waves"] == 1, st["result_waves"]
    # (the samplesort path streams nothing: one copy after the build)
    monkeypatch.setenv("CAPS_SA_PATH", "classic")
    monkeypatch.setenv("CAPS_SA_HOST_WAVES", "4")
    SA, LCP, st = E.build(uni, p=0)
    SAo, LCPo = oracle.build_sa_lcp(uni, p=64)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo) and st["result_waves"] == 1


def _repeat_rich(rs, n, alphabet=DNA):
    """GRCh38-shaped repeat content in small (tools/genome_like.py plant_genome_repeats): a tandem array of a 57-char monomer with
    2 % divergence, a higher-order array (unit of 5 monomers 20 % apart, units 0.7 % apart), 300 copies of a 60-char family at
    10 %, one exact long duplicate, and the end of the text inside a repeat."""
    T = rs.choice(alphabet, size=n)

    def mutate(seg, rate):
        seg = seg.copy()
        m = rs.rand(seg.size) < rate
        seg[m] = rs.choice(alphabet, size=int(m.sum()))
        return seg
    mono = rs.choice(alphabet, size=57)
    a0, half = n // 10, n // 30
    T[a0:a0 + half] = mutate(np.tile(mono, half // 57 + 1)[:half], 0.02)
    unit = np.concatenate([mutate(mono, 0.2) for _ in range(5)])
    T[a0 + half:a0 + 2 * half] = mutate(np.tile(unit, half // unit.size + 1)[:half], 0.007)
    cons = rs.choice(alphabet, size=60)
    for pos in rs.randint(n // 3, n - n // 8, size=100):
        T[pos:pos + 60] = mutate(cons, 0.10)
    T[n - n // 16:n - n // 16 + 3000] = T[n // 50:n // 50 + 3000]          # an exact duplicate
    T[n - 700:] = np.tile(mono, 13)[:700]                                  # the text ends inside the tandem array's content
    return T


def test_large_groups_of_equal_keys_are_rekeyed_not_compared(monkeypatch):
    """Deferred ties (kernels.h "Deferred ties", pipeline.h msd_refine): the tiles the equalised tile sort cannot finish and the
    buckets larger than a tile order equal keys without reading the text and leave sentinel LCPs; the groups are then ordered
    by re-keying them 32 chars deeper per level, small ones by direct comparison.  Tandem arrays, a repeat family, an exact
    duplicate and a text that ends inside a repeat, 2-bit and 8-bit codes, both index widths; against the independent
    construction of tests/sa_check.py (the oracle is quadratic on such texts), and against the build with CAPS_SA_NO_DEFER."""
    from emul_util import emul_small
    from sa_check import sa_lcp
    E = emul_small()
    rs = np.random.RandomState(47)
    for n, alphabet, bits in [(150_000, DNA, 32), (90_000, DNA, 64), (120_000, np.frombuffer(b"acgtn\x80\xfe", dtype=np.uint8), 32)]:
        T = _repeat_rich(rs, n, alphabet)
        SAo, LCPo = sa_lcp(T, idx_bits=bits)
        for mode in (None, "quantile", "linear"):
            if mode:
                monkeypatch.setenv("CAPS_SA_DIRECT_MODE", mode)
            else:
                monkeypatch.delenv("CAPS_SA_DIRECT_MODE", raising=False)
            SA, LCP, st = E.build(T, p=0, idx_bits=bits)
            assert np.array_equal(SA, SAo), (n, bits, mode, st["path_direct"], st["tie_groups_deferred"])
            assert np.array_equal(LCP, LCPo), (n, bits, mode)
            if st["path_direct"]:
                assert st["tie_groups_deferred"] > 0 and st["tie_elems_deferred"] >= 2 * st["tie_groups_deferred"], (n, mode, st)
        monkeypatch.setenv("CAPS_SA_NO_DEFER", "1")
        SA, LCP, st = E.build(T, p=0, idx_bits=bits)
        monkeypatch.delenv("CAPS_SA_NO_DEFER")
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo) and st["tie_groups_deferred"] == 0
        # the groups do not fit the work memory (forced): the build is done again with every tie compared
        monkeypatch.delenv("CAPS_SA_DIRECT_MODE", raising=False)
        monkeypatch.setenv("CAPS_SA_TEST_MSD_FAIL", "1")
        for waves in (None, "3"):
            if waves:
                monkeypatch.setenv("CAPS_SA_HOST_WAVES", waves)
            SA, LCP, st = E.build(T, p=0, idx_bits=bits)
            assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo) and st["tie_groups_deferred"] == 0 and st["path_direct"] == 1
        monkeypatch.delenv("CAPS_SA_TEST_MSD_FAIL")
        # ... and results that leave in waves: every wave settles its own groups before its slice is copied out
        SA, LCP, st = E.build(T, p=0, idx_bits=bits)
        monkeypatch.delenv("CAPS_SA_HOST_WAVES")
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo) and st["tie_groups_deferred"] > 0 and st["result_waves"] >= 2


def _tandem(rs, T, at, unit_len, copies, rate, alphabet=DNA):
    unit = rs.choice(alphabet, size=unit_len)
    seg = np.tile(unit, copies)
    m = rs.rand(seg.size) < rate
    seg[m] = rs.choice(alphabet, size=int(m.sum()))
    T[at:at + seg.size] = seg


def test_groups_larger_than_a_tile_take_the_level_loop(monkeypatch):
    """A group of equal keys of at most a tile of members is finished by one workgroup in LDS (msd_quick_kernel: multikey quicksort,
    no level of msd_refine's loop); a larger one is re-keyed and sorted level by level until its parts fit.  A tandem array of a
    23-char unit with 700 copies at the 256-element tiles (groups of ~700 > 256), 6000 copies at the 4096-element tiles; texts
    that end inside the array; both index widths.  tie_levels counts the levels of the loop."""
    from emul_util import emul_small
    from sa_check import sa_lcp
    rs = np.random.RandomState(71)
    for E, n, copies, bits in [(emul_small(), 120_000, 700, 32), (emul_small(), 150_001, 900, 64), (emul(), 1_000_000, 6000, 32)]:
        T = rs.choice(DNA, size=n)
        _tandem(rs, T, n // 3, 23, copies, 0.003)
        _tandem(rs, T, n - 23 * 40, 23, 40, 0.0)                  # (another unit: the text ends inside a short exact array)
        SAo, LCPo = sa_lcp(T, idx_bits=bits)
        SA, LCP, st = E.build(T, p=0, idx_bits=bits)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (n, bits)
        assert st["path_direct"] == 1 and st["tie_groups_deferred"] > 0 and st["tie_levels"] >= 2, (n, st["tie_groups_deferred"], st["tie_levels"])
    # an EXACT array of 900 copies: every level only peels the copies that leave the array inside its 32 chars -- ~650 levels; past
    # MSD_MAX_LEVELS (pipeline.h) the groups go to the comparators after all: the build is done again without deferring
    E = emul_small()
    T = rs.choice(DNA, size=150_000)
    _tandem(rs, T, 50_000, 23, 900, 0.0)
    SAo, LCPo = sa_lcp(T)
    SA, LCP, st = E.build(T, p=0)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    assert st["path_direct"] == 1 and st["tie_groups_deferred"] == 0, (st["tie_groups_deferred"], st["tie_levels"])
    monkeypatch.setenv("CAPS_SA_TEST_MSD_MAX_LEVELS", "100000")     # ... and level by level to the end, with the cap lifted
    SA, LCP, st = E.build(T, p=0)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    assert st["tie_groups_deferred"] > 0 and st["tie_levels"] > 128, (st["tie_groups_deferred"], st["tie_levels"])


def test_deferred_groups_that_agree_for_thousands_of_chars_jump(monkeypatch):
    """Found on the genome-like text with N-block stand-ins (bench g3n: 100 single-letter runs of 50,000): the suffixes that start one
    char before such a run -- 45 of them with the same char -- share 50,001 chars; 32 chars per level were 1,500 levels (98 -> 181 ms).
    A level that settles nothing makes every open group jump to what all its members share (msd_jump_kernel; the run table makes a
    periodic stretch one step).  Here: 40 runs of 1,200 G's behind an A, 36 exact copies of a 2,500-char segment, and both at once."""
    from emul_util import emul_small
    from sa_check import sa_lcp
    E = emul_small()
    rs = np.random.RandomState(61)
    n = 400_000
    T = rs.choice(DNA, size=n)
    for k in range(40):
        a = 5_000 + k * 9_000
        T[a] = ord("A")
        T[a + 1:a + 1_201] = ord("G")
        T[a + 1_201] = DNA[k % 3 if k % 3 != 2 else 3]          # terminators A / C / T
    seg = rs.choice(DNA, size=2_500)
    for k in range(36):
        a = 9_500 + k * 9_000
        T[a:a + 2_500] = seg
    SAo, LCPo = sa_lcp(T)
    SA, LCP, st = E.build(T, p=0)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    assert st["path_direct"] == 1 and st["tie_groups_deferred"] > 0 and st["tie_levels"] <= 4, (st["tie_groups_deferred"], st["tie_levels"])


def test_a_suffix_that_ends_inside_the_key_of_a_deferred_group(oracle):
    """Found by the -m gpu suite (test_skewed_and_texty_inputs_device): in a text of 70 % A's some suffix near the end of the text
    has -- padded -- the key of a longer one; the stable merges of a sort that defers ties can leave it BEHIND the longer one, and
    the LCP the sort emitted for the suffix after the pair was capped by its length.  The LCPs at both ends of every group are
    taken from the text again once the groups are in order (msd_fix_edges_kernel)."""
    from emul_util import emul_small
    rs = np.random.RandomState(21)
    T = rs.choice(DNA, size=3_000_000, p=[0.7, 0.1, 0.1, 0.1])
    SAo, LCPo = oracle.build_sa_lcp(T, p=50)
    SA, LCP, st = emul().build(T, p=50)
    assert st["path_direct"] == 1 and st["tie_groups_deferred"] > 0
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    for seed in range(6):                                   # the same shape at the 256-element tiles: many more group edges per suffix
        rs = np.random.RandomState(100 + seed)
        T = rs.choice(DNA, size=150_000 + seed, p=[0.72, 0.1, 0.1, 0.08])
        T[-40:] = ord("A")                                  # the text ends in a run: every one of its suffixes ends inside a key
        SAo, LCPo = oracle.build_sa_lcp(T, p=20)
        SA, LCP, st = emul_small().build(T, p=20)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), seed
