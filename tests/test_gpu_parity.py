"""GPU (-m gpu): parity of the HIP path with the oracle, through the C ABI of
libcaps_sa_hip.so.  Bit-exact: SA and LCP are integer arrays.

Sizes: oracle comparisons at sizes the oracle finishes in seconds; committed golden
fixtures; and, at BASELINE.json's sizes, size-independent properties (exact verifier on
the device: permutation + adjacent-pair order + exact LCP) plus closed forms."""
import numpy as np
import pytest

from conftest import LARGE_GOLDEN, large_golden, text_bytes

pytestmark = pytest.mark.gpu
DNA = np.frombuffer(b"ACGT", dtype=np.uint8)


@pytest.fixture(scope="module")
def L():
    import torch  # noqa: F401  -- first, so that this process has ONE HIP runtime (torch's)
    import caps_sa_amd
    lib = caps_sa_amd.lib()
    if lib.device_count() < 1:
        pytest.fail("no HIP device: the -m gpu tests need a GPU (there is no CPU fallback)")
    return lib


def _same(L, oracle, T, p, bits=32, ref="build"):
    SA, LCP, st = L.build(T, p=p, idx_bits=bits)
    if ref == "build":
        SAo, LCPo = oracle.build_sa_lcp(T, p=p, idx_bits=bits)
    else:
        SAo, LCPo = oracle.naive_sa_lcp(T, idx_bits=bits)
    assert np.array_equal(SA, SAo), f"SA mismatch n={T.size} p={p}"
    assert np.array_equal(LCP, LCPo), f"LCP mismatch n={T.size} p={p}"
    return st


def test_golden_cases(L, golden_cases):
    for c in golden_cases:
        T = text_bytes(c["text"])
        for p in (0, 2, 5):
            SA, LCP, _ = L.build(T, p=p)
            assert SA.tolist() == c["sa"], (c["name"], p)
            assert LCP.tolist() == c["lcp"], (c["name"], p)


@pytest.mark.parametrize("name", LARGE_GOLDEN)
def test_large_golden_cases_drive_the_direct_path(L, name, monkeypatch):
    """Reference-made vectors (chatgpt_baseline.py, tests/golden/make_golden.py) of 136k .. 250k chars: long enough for the
    DEFAULT construction, so the direct path is compared with reference output itself, not only through the oracle; the
    latin-1 case pins the signed-char order (src/Suffix_Array.cpp:75-77) the same way."""
    T, sa, lcp = large_golden(name)
    for path in ("auto", "classic"):
        monkeypatch.setenv("CAPS_SA_PATH", path)
        for p in (0, 8000):
            SA, LCP, st = L.build(T, p=p)
            assert np.array_equal(SA, sa) and np.array_equal(LCP, lcp), (name, path, p)
            if path == "classic":
                assert st["path_direct"] == 0
            elif p == 0:
                assert st["path_direct"] == 1, (name, st["path_fallback"])


def test_reference_dump_digest_pins(L, oracle):
    """Same digests the real reference produced (tests/test_oracle_pins.py)."""
    for seed, N, dump_sha in [(1, 1000, "fb18c177a9caa2ae"), (7, 4096, "af073a56da834b57"),
                              (123, 100000, "24db5e1b31a804d4")]:
        T = oracle.remap(oracle.gen_rand_seq(seed, N))
        SA, LCP, _ = L.build(T)
        assert oracle.dump_sha256(SA, LCP)[:16] == dump_sha


@pytest.mark.parametrize("n,p", [(200000, 7), (200001, 0), (300007, 16), (100000, 1), (4096, 2), (4097, 2),
                                 (8191, 3), (8193, 0), (65536 + 17, 4), (1, 0), (5, 0), (31, 0), (32, 0), (33, 3)])
def test_random_dna_vs_naive(L, oracle, sa_path, n, p):
    rs = np.random.RandomState(n % 1000 + p)
    st = _same(L, oracle, rs.choice(DNA, size=n), p, ref="naive")
    assert st["bits_per_char"] == 2


def test_c1_like_fasta_standin_vs_oracle(L, oracle):
    """BASELINE config 0 stand-in: header line + bases wrapped at 70 columns, CLI remap, p=8000."""
    rs = np.random.RandomState(1)
    body = rs.choice(DNA, size=4_641_652)
    lines = [b">synthetic stand-in for data/ecoli.fa\n"]
    raw = body.tobytes()
    lines += [raw[i:i + 70] + b"\n" for i in range(0, len(raw), 70)]
    T = oracle.remap(np.frombuffer(b"".join(lines), dtype=np.uint8))
    st = _same(L, oracle, T, 8000)
    assert st["p_eff"] == 8000


def test_16mi_vs_oracle_and_reference_digest(L, oracle, sa_path):
    """16 Mi bases of the reference's own generator: equal to the oracle's build array by array, and to the sha256 of the
    dump file the reference produced in the survey session (a bonus pin)."""
    T = oracle.remap(oracle.gen_rand_seq(42, 16 * 1024 * 1024))
    SA, LCP, st = L.build(T, p=8000)
    SAo, LCPo = oracle.build_sa_lcp(T, p=8000)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    assert oracle.dump_sha256(SA, LCP) == "8feac4aca81da6d0457628f62282ea6225837697123508e7359b9dacf5abf64e"
    assert st["path_direct"] == (0 if sa_path == "classic" else 1)


def test_byte_alphabets_and_signed_order(L, oracle):
    rs = np.random.RandomState(3)
    st = _same(L, oracle, rs.choice(np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", dtype=np.uint8), size=1_000_000), 37)
    assert st["bits_per_char"] == 8
    _same(L, oracle, rs.choice(np.array([0x41, 0x7F, 0x80, 0xFF, 0], dtype=np.uint8), size=300000), 6)
    _same(L, oracle, rs.randint(0, 256, size=500000).astype(np.uint8), 0)


def test_deep_lcp_inputs(L, oracle):
    rs = np.random.RandomState(4)
    n = 100_000                                   # BASELINE config 5 shape (a^n); full size: test_c5_unary_text_1e8_closed_form
    SA, LCP, _ = L.build(np.full(n, ord("a"), dtype=np.uint8), p=0)
    assert np.array_equal(SA, np.arange(n - 1, -1, -1, dtype=np.uint32))       # SURVEY 0.8 closed form
    assert np.array_equal(LCP, np.arange(n, dtype=np.uint32))
    _same(L, oracle, np.tile(np.frombuffer(b"AC", dtype=np.uint8), 20000), 3, ref="naive")
    _same(L, oracle, np.tile(rs.choice(DNA, size=37), 2000), 9, ref="naive")
    _same(L, oracle, rs.choice(np.frombuffer(b"AT", dtype=np.uint8), size=300000, p=[0.9, 0.1]), 4)
    # planted long repeats inside random DNA (genome-like)
    T = rs.choice(DNA, size=2_000_000)
    T[1_000_000:1_050_000] = T[100_000:150_000]
    T[1_500_000:1_500_400] = ord("G")
    _same(L, oracle, T, 64)


def test_u64_indices(L, oracle):
    rs = np.random.RandomState(5)
    _same(L, oracle, rs.choice(DNA, size=1_200_000), 111, bits=64)


def test_bounded_context_follows_the_reference_merge_history(L, oracle):
    """SURVEY f4 (csrc/bounded.h): 0 < max_context < n reproduces the reference's merge history (ties go to run "Y",
    src/Suffix_Array.cpp:71-92) -- PARITY UNPINNED by any reference-held vector; compared with the oracle's restatement."""
    import caps_sa_amd
    rs = np.random.RandomState(5)
    S = rs.choice(DNA, size=30000)
    for T, p, ctx, bits in [(rs.choice(DNA, size=50000), 7, 3, 32), (rs.choice(DNA, size=200001), 0, 9, 32), (rs.choice(DNA, size=200001), 113, 1, 32),
                            (rs.choice(DNA, size=30000), 16, 2, 64), (np.tile(rs.choice(DNA, size=37), 2000), 9, 20, 32),
                            (np.full(40000, ord("A"), np.uint8), 4, 10, 32), (np.concatenate([S, S, S]), 6, 50, 32),
                            (rs.choice(np.frombuffer(b"abcdefgh\x80\xff", dtype=np.uint8), size=80000), 5, 2, 32),
                            (rs.choice(DNA, size=1_000_000), 8000, 12, 32)]:
        SA, LCP, st = L.build(T, p=p, max_context=ctx, idx_bits=bits)
        SAo, LCPo = oracle.build_sa_lcp(T, p=p, max_context=ctx, idx_bits=bits)[:2]
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (T.size, p, ctx, bits)
        assert st["path_fallback"] == 7                      # CAPS_SA_FB_BOUNDED
        # ... and the check that shares no code with bounded.h or the oracle (tests/sa_check.py check_bounded: permutation, neighbours
        # in order on their first ctx + 1 chars, LCPs between min(lcp, ctx) and lcp; it cannot pin the order of ties)
        from sa_check import check_bounded
        assert not any(check_bounded(T, SA, LCP, ctx).values()), (T.size, p, ctx, check_bounded(T, SA, LCP, ctx))
    T = rs.choice(DNA, size=1000)
    a, b = L.build(T, max_context=1000), L.build(T)          # >= n is unbounded
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(caps_sa_amd.CapsSaError):
        L.build(T[:20], max_context=3)


def test_kernel_level_entry_points(L, oracle):
    rs = np.random.RandomState(7)
    T = rs.choice(DNA, size=600000)
    idx = rs.permutation(600000)[:250000].astype(np.uint32)
    sa, lcp = L.sort_suffixes(T, idx)                                   # a4 merge_sort
    so, lo = oracle.merge_sort(T, idx)
    assert np.array_equal(sa, so) and np.array_equal(lcp, lo)
    xa, xl = oracle.merge_sort(T, idx[:90001])                          # a3 merge
    ya, yl = oracle.merge_sort(T, idx[90001:])
    Z, LZ = L.merge(T, xa, ya, xl, yl)
    Zo, LZo = oracle.merge(T, xa, ya, xl, yl)
    assert np.array_equal(Z, Zo) and np.array_equal(LZ, LZo)
    piv = np.concatenate([sa[::9973], rs.randint(0, 600000, size=50).astype(np.uint32)])
    ub = L.upper_bound(T, sa, piv)                                      # a7 upper_bound
    for pv, u in zip(piv.tolist(), ub.tolist()):
        assert u == oracle.upper_bound(T, sa, pv)
    a = rs.randint(0, 600000, size=5000).astype(np.uint32)              # a2 LCP
    b = rs.randint(0, 600000, size=5000).astype(np.uint32)
    out = L.lcp(T, a, b)
    for x, y, l in zip(a.tolist(), b.tolist(), out.tolist()):
        assert l == oracle.lcp(T, x, y)


def test_suffix_array_class_mirror(L, oracle, tmp_path):
    import caps_sa_amd
    T = oracle.remap(oracle.gen_rand_seq(7, 4096))
    sa = caps_sa_amd.SuffixArray(T, subproblem_count=16)
    with pytest.raises(RuntimeError):
        sa.SA()
    sa.construct()
    assert sa.n() == 4097
    path = tmp_path / "dump.bin"
    sa.dump(str(path))
    import hashlib
    assert hashlib.sha256(path.read_bytes()).hexdigest()[:16] == "af073a56da834b57"   # reference dump digest


def _device_build_and_verify(L, n, p, seed):
    """Device-resident build + exact device verifier; returns stats."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    T = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 28
    for o in range(0, n, step):
        m = min(step, n - o)
        T[o:o + m] = lut[torch.randint(0, 4, (m,), device="cuda", generator=g, dtype=torch.int64)]
    T[n - 1] = ord("C")                                   # the CLI maps the trailing newline to 'C'
    SA = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=p)
    errs = L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr())
    assert errs == 0, f"{errs} violations"
    # a corrupted result must be caught by the verifier
    SA[n // 2], SA[n // 2 + 1] = SA[n // 2 + 1].clone(), SA[n // 2].clone()
    assert L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr()) > 0
    return st


def test_c2_256mib_device_resident(L, sa_path):
    """BASELINE config 1: 256 MiB random DNA, u32, p = 8000 (both constructions)."""
    st = _device_build_and_verify(L, 268_435_457, 8000, 42)
    assert st["p_eff"] == 8000 and st["bits_per_char"] == 2
    assert st["path_direct"] == (0 if sa_path == "classic" else 1), st


def test_c3_3g_device_resident(L, sa_path):
    """BASELINE headline size: 3e9 bases (+1), u32, p = 8000.  Property check on the device (both constructions;
    the default must be the direct path with its splits kept in slots)."""
    st = _device_build_and_verify(L, 3_000_000_001, 8000, 42)
    assert st["p_eff"] == 8000
    assert st["path_direct"] == (0 if sa_path == "classic" else 1), st
    if sa_path != "classic":
        assert st["path_fallback"] == 0 and st["slot_splits_redone"] == 0 and st["direct_groups"] == 1000, st


@pytest.mark.parametrize("kind", ["genome", "genome+n", "genome+r"])
def test_c3_genome_like_3g_device_resident(L, kind):
    """BASELINE config 3 is GRCh38; its stand-ins at full size (bench.py --workload g3 / g3n / g3r: skewed order-5 Markov
    text with planted mutated repeats; + N-block stand-ins; + satellite arrays, a 1e5-copy repeat family and segmental
    duplications) take kernels the uniform 3e9 text never reaches -- quantile level B, tile_sort_eq_kernel, the run-table
    merge passes -- with counts near 2^32.  Exact device verifier; the default construction must be the direct path in
    quantile mode."""
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import make_text
    T = make_text(torch, 3_000_000_000, 42, torch.device("cuda", 0), kind)
    n = T.numel()
    SA = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP = torch.empty(n, dtype=torch.int32, device="cuda")
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000)
    errs = L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr())
    brief = {k: st[k] for k in ("path_direct", "path_fallback", "direct_quantile", "long_runs", "direct_groups", "direct_max_group", "ms_total")}
    assert errs == 0, (errs, brief)
    assert st["p_eff"] == 8000 and st["path_direct"] == 1 and st["direct_quantile"] == 1, brief
    if kind == "genome+n":
        assert st["long_runs"] == 1 and int(LCP.max().item()) >= 1_999_000, brief
    if kind == "genome+r":
        assert int(LCP.max().item()) >= 49_999, brief            # the exact 50-kb duplicate
    del T, SA, LCP
    torch.cuda.empty_cache()


def test_genome_like_256mi_with_grch38_shaped_repeats_device(L, sa_path):
    """tools/genome_like.py plant_genome_repeats at BASELINE config 1's size: a satellite array of period 171 and its
    higher-order repeat of period 2052 (thousands of suffixes share a 32-base key and differ ~200 chars on), a 300-base
    family in 6,250 copies, 100-kb duplications at 1 % and an exact 50-kb duplicate.  Both constructions, exact verifier."""
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import make_text
    T = make_text(torch, 268_435_456, 42, torch.device("cuda", 0), "genome+r")
    n = T.numel()
    SA = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP = torch.empty(n, dtype=torch.int32, device="cuda")
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000)
    assert L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr()) == 0
    assert st["path_direct"] == (0 if sa_path == "classic" else 1), st
    assert int(LCP.max().item()) >= 49_999
    if sa_path != "classic":                         # the cluster tiles' ties were deferred and re-keyed, many levels deep
        assert st["tie_groups_deferred"] > 1000, (st["tie_groups_deferred"], st["tie_levels"])


def test_sharded_driver_single_rank_rccl(L, sa_path):
    """The multi-GPU driver (caps_sa_dist.py) with the real kernels and backend nccl (RCCL) at
    world size 1 -- all a 1-GPU box allows; world sizes 2 and 3 run on CPU over gloo
    (tests/test_dist_gloo.py).  Checked with the device verifier and against the 1-GPU build."""
    import os
    import torch
    import torch.distributed as dist
    import caps_sa_dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n = 20_000_001
        g = torch.Generator(device="cuda")
        g.manual_seed(3)
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
        T = lut[torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int64)]
        SA, LCP, off, info = caps_sa_dist.build_sharded(L, T, 500, 32)
        assert off == 0 and SA.numel() == n
        assert info["path"] == ("samplesort" if sa_path == "classic" else "direct"), info
        assert L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr()) == 0
        SA1 = torch.empty(n, dtype=torch.int32, device="cuda")
        LCP1 = torch.empty(n, dtype=torch.int32, device="cuda")
        L.build_device(T.data_ptr(), n, SA1.data_ptr(), LCP1.data_ptr(), p=500)
        assert torch.equal(SA, SA1) and torch.equal(LCP, LCP1)
    finally:
        dist.destroy_process_group()


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


def test_segmented_sort_mixed_lengths(L, oracle):
    """Segments of 0, 1, <1 tile, exactly 1/2/3/5/8 tiles (+-1): finished segments sit out later
    passes and end in different ping-pong buffers (parity of passes_for(len))."""
    rs = np.random.RandomState(8)
    T = rs.choice(DNA, size=200000)
    lens = [5000, 0, 1, 4096, 4097, 0, 8192, 8193, 12288, 3, 20480, 20481, 32768, 100, 0, 16385, 7]
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    idx = rs.permutation(200000)[:int(seg[-1])].astype(np.uint32)
    _check_segments(L, oracle, T, idx, seg)


def test_cli_dump_matches_reference_digest(L, oracle, tmp_path):
    """The C++ host side (caps-sa_amd/csrc/Suffix_Array.hpp + caps_sa_cli.cpp, the mirror of the
    reference's src/main.cpp): same arguments, same remap, same dump bytes as the real reference."""
    import hashlib
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "caps-sa_amd"), "caps_sa"])
    exe = os.path.join(root, "caps-sa_amd", "caps_sa")
    raw = oracle.gen_rand_seq(123, 100000)                     # file bytes incl. the trailing newline
    inp, out = tmp_path / "in.txt", tmp_path / "out.bin"
    inp.write_bytes(raw.tobytes())
    subprocess.check_call([exe, str(inp), str(out), "64"])
    assert hashlib.sha256(out.read_bytes()).hexdigest()[:16] == "24db5e1b31a804d4"
    # --pretty-print: the text form of src/main.cpp:32-40 / chatgpt_baseline.py:40-42
    small = tmp_path / "small.txt"
    small.write_bytes(open(os.path.join(root, "tests", "golden", "simpletest2.input"), "rb").read())
    txt = tmp_path / "small.out"
    subprocess.check_call([exe, str(small), str(txt), "--pretty-print"])
    lines = txt.read_text().strip().split("\n")
    SA, LCP = oracle.build_sa_lcp(oracle.remap(small.read_bytes()))
    assert lines[0].split() == [str(x) for x in SA.tolist()]
    assert lines[1].split() == [str(x) for x in LCP.tolist()]
    # argv[4] = bounded context (src/main.cpp:57): the dump of the reference's bounded result (csrc/bounded.h; oracle restatement)
    outb = tmp_path / "bounded.bin"
    subprocess.check_call([exe, str(inp), str(outb), "64", "7"])
    SAb, LCPb = oracle.build_sa_lcp(oracle.remap(raw), p=64, max_context=7)[:2]
    assert hashlib.sha256(outb.read_bytes()).hexdigest() == oracle.dump_sha256(SAb, LCPb)


def test_u64_device_resident_50m(L):
    """64-bit index kernels at a size where every stage (bucketing, chunked scans) is exercised."""
    import torch
    n = 50_000_001
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    T = lut[torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int64)]
    SA = torch.empty(n, dtype=torch.int64, device="cuda")
    LCP = torch.empty(n, dtype=torch.int64, device="cuda")
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=1000, idx_bits=64)
    assert st["idx_bytes"] == 8
    assert L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), idx_bits=64) == 0


def test_skewed_and_texty_inputs_device(L, oracle, sa_path):
    """Keys far from uniform in their range: buckets overflow their tile and are finished by the
    LCP-merge passes (skip_finished / unify / finalize-with-records paths)."""
    rs = np.random.RandomState(21)
    T = rs.choice(DNA, size=3_000_000, p=[0.7, 0.1, 0.1, 0.1])
    st = _same(L, oracle, T, 50)
    if sa_path == "classic":
        assert st["merge_passes_phase1"] + st["merge_passes_phase2"] > 0
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz ", dtype=np.uint8)
    _same(L, oracle, rs.choice(letters, size=2_000_000, p=np.r_[np.full(26, 0.03), 0.22]), 40)


def test_genome_like_markov_with_repeats_device(L, sa_path):
    """Order-5 Markov chain with skewed transitions + planted mutated repeats (tools/genome_like.py):
    most tiles leave the bucket-sort fast path (samplesort / rank-merge levels), buckets overflow into
    LCP-merge passes.  Checked with the exact device verifier."""
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from genome_like import markov_dna
    n = 30_000_001
    T = markov_dna(n, seed=11)
    SA = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP = torch.empty(n, dtype=torch.int32, device="cuda")
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=1000)
    assert L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr()) == 0
    assert int(LCP.max().item()) > 200          # the planted repeats are there


def test_skewed_texts_with_64_bit_indices_and_8_bit_codes_device(L):
    """The builds of tile_sort_eq_kernel / tile_sort_general_kernel that take one queue entry per workgroup (skewed texts: quantile
    level B, every tile queued) exist per index width and code width: genome-like text with 64-bit indices (no slot-order
    ranking there: no room in LDS), and a skewed 20-letter text (8-bit codes) with both widths.  Exact device verifier."""
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from genome_like import markov_dna
    n = 40_000_001
    T = markov_dna(n, seed=13)
    SA = torch.empty(n, dtype=torch.int64, device="cuda")
    LCP = torch.empty(n, dtype=torch.int64, device="cuda")
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=1000, idx_bits=64)
    assert st["path_direct"] == 1 and st["direct_quantile"] == 1
    assert L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), idx_bits=64) == 0
    del SA, LCP
    # 20 letters, geometric frequencies, every 50th block of 200 chars repeats an earlier block (equal keys, deep ties)
    g = torch.Generator(device="cuda").manual_seed(5)
    w = torch.tensor([0.75 ** i for i in range(20)], device="cuda")
    n = 24_000_000
    T = (torch.multinomial(w / w.sum(), n, replacement=True, generator=g) + 65).to(torch.uint8)
    blocks = T.view(-1, 200)
    src = torch.randint(0, blocks.shape[0], (blocks.shape[0] // 50,), device="cuda", generator=g)
    blocks[torch.arange(0, (blocks.shape[0] // 50) * 50, 50, device="cuda")] = blocks[src]
    for bits, dt in ((32, torch.int32), (64, torch.int64)):
        SA = torch.empty(n, dtype=dt, device="cuda")
        LCP = torch.empty(n, dtype=dt, device="cuda")
        st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=800, idx_bits=bits)
        assert st["bits_per_char"] == 8
        assert L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), idx_bits=bits) == 0
        assert int(LCP.max().item()) >= 200
        del SA, LCP


def test_slot_splits_kept_on_uniform_keys_and_redone_on_skew(L, oracle, sa_path):
    """The bucket splits of phase 1 and phase 2 first scatter into fixed-capacity slots (no count pass).
    Uniform keys: both are kept (and C2 / C3 rely on that for their speed).  Skewed keys: a bucket
    overflows its slot, the split is redone with the count pass; both ways the result is exact."""
    st = _device_build_and_verify(L, 60_000_001, 200, 5)
    assert st["slot_splits"] == (2 if sa_path == "classic" else 1) and st["slot_splits_redone"] == 0, st
    assert st["path_direct"] == (0 if sa_path == "classic" else 1), st
    assert st["merge_passes_phase1"] == 0 and st["merge_passes_phase2"] == 0
    rs = np.random.RandomState(22)
    T = rs.choice(DNA, size=4_000_000, p=[0.6, 0.2, 0.1, 0.1])
    st = _same(L, oracle, T, 16)
    assert st["slot_splits_redone"] >= 1, st


def test_partitions_spanning_more_runs_than_a_tile_stages(L, monkeypatch):
    """p = 8000 on 40 M bases: a 4096-element tile of a 5000-element partition spans more than 4096
    of the p sorted subarrays' runs, so the phase-2 scatter resolves runs by binary search in global
    memory instead of in its LDS stage (bucket_scatter_kernel<SRC_RUNS>, !runs_staged)."""
    monkeypatch.setenv("CAPS_SA_PATH", "classic")          # the samplesort path's phase 2 is what this is about
    st = _device_build_and_verify(L, 40_000_001, 8000, 6)
    assert st["p_eff"] == 8000 and st["max_partition"] > 4096 and st["path_direct"] == 0


def test_more_subproblems_than_a_dispatch_has_threads(L, oracle, sa_path):
    """p = 72,845 (found by tools/stress_gpu.py): locate_kernel's natural grid of p^2 threads exceeds the
    2^32 work-items of one dispatch, which HIP truncates silently; the kernel loops over its blocks."""
    rs = np.random.RandomState(172)
    n = 2_913_828
    T = rs.choice(DNA, size=n, p=[0.55, 0.25, 0.15, 0.05])
    _same(L, oracle, T, n // 40, bits=64)


# ---- BASELINE config 5 and the run table (csrc/text.h) ------------------------------------------
def _check_independent(L, T, p, bits=32, long_runs=None):
    """GPU build vs tests/sa_check.py (prefix doubling: not quadratic on repeats like the oracle)."""
    from sa_check import sa_lcp
    SA, LCP, st = L.build(T, p=p, idx_bits=bits)
    SAo, LCPo = sa_lcp(T, bits)
    assert np.array_equal(SA, SAo), f"SA mismatch n={T.size} p={p}"
    assert np.array_equal(LCP, LCPo), f"LCP mismatch n={T.size} p={p}"
    if long_runs is not None:
        assert st["long_runs"] == int(long_runs)


def test_c5_unary_text_1e8_closed_form(L):
    """BASELINE config 5 at FULL size: T = 'a'^1e8, u32, default subproblem count.  The reference
    (and a plain window loop) is Theta(n^2) here; SA[i] = n-1-i, LCP[i] = i."""
    import torch
    n = 100_000_000
    T = torch.full((n,), ord("a"), dtype=torch.uint8, device="cuda")
    SA = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP = torch.empty(n, dtype=torch.int32, device="cuda")
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=0)
    assert st["p_eff"] == 8192 and st["long_runs"] == 1
    ar = torch.arange(n, device="cuda", dtype=torch.int32)
    assert bool((SA == (n - 1 - ar)).all())
    assert bool((LCP == ar).all())
    # u64 indices, smaller
    n = 5_000_000
    SA64 = torch.empty(n, dtype=torch.int64, device="cuda")
    LCP64 = torch.empty(n, dtype=torch.int64, device="cuda")
    L.build_device(T.data_ptr(), n, SA64.data_ptr(), LCP64.data_ptr(), p=100, idx_bits=64)
    ar = torch.arange(n, device="cuda", dtype=torch.int64)
    assert bool((SA64 == (n - 1 - ar)).all()) and bool((LCP64 == ar).all())


def test_periodic_stretches_vs_independent_construction(L):
    rs = np.random.RandomState(21)
    b = lambda s: np.frombuffer(s, dtype=np.uint8)                              # noqa: E731
    _check_independent(L, np.full(300_001, 0x80, np.uint8), 0, long_runs=True)   # code 0 = end-of-text padding
    _check_independent(L, np.tile(b(b"AC"), 150_000), 3, long_runs=True)
    _check_independent(L, np.tile(rs.choice(DNA, size=16), 20_000), 5, long_runs=True)
    _check_independent(L, np.tile(rs.choice(DNA, size=17), 3_000), 5, long_runs=False)
    _check_independent(L, np.tile(b(b"abcd"), 60_000), 4, long_runs=True)
    parts = [np.full(50_000, ord("G"), np.uint8), rs.choice(DNA, size=70_000), np.tile(b(b"ACG"), 30_000), rs.choice(DNA, size=100),
             np.full(90_000, ord("G"), np.uint8), rs.choice(DNA, size=3), np.tile(b(b"ACG"), 20_000), np.full(40_000, ord("A"), np.uint8)]
    _check_independent(L, np.concatenate(parts), 64, long_runs=True)
    _check_independent(L, np.concatenate(parts[::-1]), 0, bits=64, long_runs=True)
    short = [rs.choice(DNA, size=300_000), np.full(700, ord("T"), np.uint8), rs.choice(DNA, size=200_000), np.tile(b(b"GA"), 400),
             rs.choice(DNA, size=1500), np.full(900, ord("T"), np.uint8)]
    _check_independent(L, np.concatenate(short), 50, long_runs=False)
    for d in (1, 2, 3, 5, 8, 15, 16):                                              # stretch ends at every block offset
        unit = rs.choice(DNA, size=d)
        while d > 1 and len(set(unit.tolist())) == 1:
            unit = rs.choice(DNA, size=d)
        parts = []
        for off in range(0, 33, 3):
            parts += [rs.choice(DNA, size=40 + off), np.tile(unit, (1100 + 7 * off) // d + 1)[:1100 + 7 * off]]
        _check_independent(L, np.concatenate(parts), 3, long_runs=True)


def test_n_block_like_runs_in_256m_dna_device(L):
    """Random DNA with a 1e6-long single-letter block (the CLI maps N to G: src/main.cpp:61-68),
    tandem arrays and a run at the very end; exact device verifier (its own scan of the raw bytes
    is linear in the sum of LCPs: ~5e11 char steps here)."""
    import torch
    n = 268_435_457
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    T = lut[torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int64)]
    T[50_000_000:51_000_000] = ord("G")
    T[120_000_003:120_200_003] = torch.tensor(list(b"AC"), dtype=torch.uint8, device="cuda").repeat(100_000)
    T[200_000_001:200_030_001] = torch.tensor(list(b"AATCG"), dtype=torch.uint8, device="cuda").repeat(6_000)
    T[n - 100_000:] = ord("T")
    SA = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP = torch.empty(n, dtype=torch.int32, device="cuda")
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000)
    assert st["long_runs"] == 1
    assert L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr()) == 0
    assert int(LCP.max().item()) == 999_999


def test_host_entry_point_cache_and_pinned_results(L, oracle):
    """The host-buffer entry point keeps and re-grows its device block; page-locked result arrays."""
    rs = np.random.RandomState(31)
    L.release_cache()
    for n, p, bits in [(500_000, 40, 32), (3000, 2, 32), (1_200_000, 9, 64), (700_000, 0, 32)]:
        _same(L, oracle, rs.choice(DNA, size=n), p, bits=bits)
    L.release_cache()
    T = rs.choice(DNA, size=300_000)
    SA, LCP, _ = L.build(T, p=7, pinned=True)
    SAo, LCPo = oracle.build_sa_lcp(T, p=7)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)


@pytest.mark.parametrize("world,n,p,bits,exchange", [(2, 20_000_001, 8000, 32, 0), (2, 20_000_001, 8000, 32, 1), (3, 6_000_000, 500, 32, 0),
                                                     (4, 9_000_001, 0, 64, 0), (4, 9_000_001, 0, 64, 1), (8, 40_000_000, 8000, 32, 0),
                                                     (8, 40_000_000, 8000, 32, 1)])
def test_shard_kernels_at_world_sizes_above_one_loopback(L, sa_path, world, n, p, bits, exchange, monkeypatch):
    """Every rank's shard on the one GPU of the box, collectives replaced by copies
    (tests/loopback_world.py): result == the single-GPU build, bit for bit.  Both constructions; world 3 carries a long
    run, on which every rank of the direct path must fall back together.  exchange = 0: the direct path's default (every rank
    scatters the whole text and keeps its groups, nothing travels); 1: CAPS_SA_SHARD_EXCHANGE (tiles shared, streams exchanged)."""
    import torch
    from loopback_world import build_world
    if exchange and sa_path == "classic":
        pytest.skip("the samplesort path always exchanges")
    if exchange:
        monkeypatch.setenv("CAPS_SA_SHARD_EXCHANGE", "1")
    else:
        monkeypatch.delenv("CAPS_SA_SHARD_EXCHANGE", raising=False)
    g = torch.Generator(device="cuda")
    g.manual_seed(world * 1000 + 7)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    T = lut[torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int64)]
    if world == 3:
        T[1_000_000:1_200_000] = ord("G")                     # a long run: run-table comparators in the shard path
        T[3_000_000:3_050_000] = T[100_000:150_000]           # a long repeat
    dt = torch.int32 if bits == 32 else torch.int64
    SA1 = torch.empty(n, dtype=dt, device="cuda")
    LCP1 = torch.empty(n, dtype=dt, device="cuda")
    L.build_device(T.data_ptr(), n, SA1.data_ptr(), LCP1.data_ptr(), p=p, idx_bits=bits)
    assert L.verify_device(T.data_ptr(), n, SA1.data_ptr(), LCP1.data_ptr(), idx_bits=bits) == 0
    SA, LCP = build_world(L, T, p, world, bits)
    assert torch.equal(SA, SA1)
    assert torch.equal(LCP, LCP1)
    assert build_world.last_path == ("direct" if sa_path != "classic" and world != 3 else "samplesort")
    if build_world.last_path == "direct":                     # exchange mode on random DNA: 32-bit keys cross the wire, no slot overflows
        assert (build_world.last_exchange, build_world.last_key_bytes, build_world.last_key_retry) == ((1, 4, 0) if exchange else (0, 8, 0))


@pytest.mark.parametrize("world,kind", [(4, "genome"), (2, "genome+n"), (8, "genome+n"), (2, "genome+r"), (4, "genome+r")])
def test_shard_quantile_mode_on_genome_like_text_loopback(L, world, kind, monkeypatch):
    """The sharded direct path on skewed keys and N-block stand-ins (tools/genome_like.py, 96 Mi bases): every rank takes the
    quantile buckets of Builder::run_direct for the groups it owns (no exchange) -- same arrays as the single-GPU build."""
    import os
    import sys
    import torch
    from loopback_world import build_world
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import make_text
    monkeypatch.delenv("CAPS_SA_SHARD_EXCHANGE", raising=False)
    T = make_text(torch, 100_663_296, 42, torch.device("cuda", 0), kind)
    n = T.numel()
    SA1 = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP1 = torch.empty(n, dtype=torch.int32, device="cuda")
    st = L.build_device(T.data_ptr(), n, SA1.data_ptr(), LCP1.data_ptr(), p=8000)
    assert st["path_direct"] == 1 and st["direct_quantile"] == 1
    assert L.verify_device(T.data_ptr(), n, SA1.data_ptr(), LCP1.data_ptr()) == 0
    SA, LCP = build_world(L, T, 8000, world, 32)
    assert torch.equal(SA, SA1) and torch.equal(LCP, LCP1)
    assert build_world.last_path == "direct" and build_world.last_quantile == 1 and build_world.last_exchange == 0
    if kind == "genome+r":          # the satellite arrays' groups of equal keys are re-keyed on the ranks that own them, not compared
        assert build_world.last_tie_groups > 100, build_world.last_tie_groups


def test_shard_key_width_retry_on_skewed_keys_loopback(L, monkeypatch):
    """Exchange mode, skewed base frequencies: level B's slots overflow under 32-bit keys, shard_sort says CAPS_SA_FB_KEY32 on
    some rank and every rank goes round again with 64-bit keys (tests/loopback_world.py asserts the code and that it happens once)."""
    import torch
    from loopback_world import build_world
    monkeypatch.setenv("CAPS_SA_SHARD_EXCHANGE", "1")
    n, p, world = 24_000_000, 3000, 4
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    u = torch.rand(n, device="cuda", generator=g)
    T = lut[(u > 0.6).long() + (u > 0.8).long() + (u > 0.9).long()]          # base frequencies 0.6 / 0.2 / 0.1 / 0.1
    del u
    SA1 = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP1 = torch.empty(n, dtype=torch.int32, device="cuda")
    L.build_device(T.data_ptr(), n, SA1.data_ptr(), LCP1.data_ptr(), p=p)
    assert L.verify_device(T.data_ptr(), n, SA1.data_ptr(), LCP1.data_ptr()) == 0
    SA, LCP = build_world(L, T, p, world, 32)
    assert torch.equal(SA, SA1) and torch.equal(LCP, LCP1)
    if build_world.last_path == "direct":
        print("key bytes", build_world.last_key_bytes, "retries", build_world.last_key_retry)
        assert build_world.last_key_bytes == 8 and build_world.last_key_retry == 1


@pytest.mark.parametrize("devices,n,p,bits", [([0, 0], 20_000_001, 8000, 32), ([0, 0, 0], 9_000_000, 0, 64), ([0], 5_000_000, 64, 32)])
def test_one_process_several_ranks_on_this_gpu(L, oracle, devices, n, p, bits):
    """caps_sa_hip_build_multi_*: the C++ multi-device entry point under Suffix_Array(T, n, p, ctx, devices).  A test box has
    one GPU, so the device list repeats ordinal 0: every rank is a Shard of its own with its own stream and buffers, the
    blocks travel by device-to-device copies exactly as they would between two GPUs."""
    rs = np.random.RandomState(len(devices) * 7 + 1)
    T = rs.choice(DNA, size=n)
    SA, LCP, st = L.build_multi(T, devices, p=p, idx_bits=bits, pinned=True)
    SA1, LCP1, st1 = L.build(T, p=p, idx_bits=bits)
    assert np.array_equal(SA, SA1) and np.array_equal(LCP, LCP1)
    assert st["path_direct"] == 1 and st["path_fallback"] == 0
    L.release_cache()


@pytest.mark.parametrize("exchange", [0, 1])
def test_ranks_with_groups_that_do_not_divide_evenly(L, oracle, exchange, monkeypatch):
    """3 or 7 groups on 8, 4, 3, 2 ranks (some own none), forced sub-streams, u64: with the exchange a rank's level B then has
    up to world * SUB more segments than K1 * SUB -- tools/stress_gpu.py (STRESS_MULTI) found the segment tables short of that
    (HIP refused the copy; the entry points also left HIP's sticky error for the next build to trip over)."""
    if exchange:
        monkeypatch.setenv("CAPS_SA_SHARD_EXCHANGE", "1")
    else:
        monkeypatch.delenv("CAPS_SA_SHARD_EXCHANGE", raising=False)
    monkeypatch.setenv("CAPS_SA_DIRECT_SUB", "8")
    rs = np.random.RandomState(5)
    T = rs.choice(np.frombuffer(b"AT", dtype=np.uint8), size=996_385, p=[0.8, 0.2])
    SAo, LCPo = oracle.build_sa_lcp(T, p=16, idx_bits=64)
    for world in (8, 4, 3, 2):
        for p in (3, 7, 16):
            SA, LCP, st = L.build_multi(T, [0] * world, p=p, idx_bits=64)
            assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (world, p)
            assert st["path_direct"] == 1
    L.release_cache()


def test_genome_like_256mi_with_n_blocks_device(L):
    """the genome-like workload at BASELINE config 1's size (tools/genome_like.py, seeded: order-5 Markov
    chain with skewed transitions + planted mutated repeats) with N-block stand-ins on top -- the direct path in quantile
    mode (frequent keys, fat group regions, run-table comparators).  Exact device verifier."""
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import make_text
    T = make_text(torch, 268_435_456, 42, torch.device("cuda", 0), "genome+n")
    n = T.numel()
    SA = torch.empty(n, dtype=torch.int32, device="cuda")
    LCP = torch.empty(n, dtype=torch.int32, device="cuda")
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000)
    assert L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr()) == 0
    brief = {k: st[k] for k in ("path_direct", "path_fallback", "direct_quantile", "long_runs", "direct_groups", "direct_max_group")}
    assert st["path_direct"] == 1 and st["direct_quantile"] == 1 and st["long_runs"] == 1, brief
    assert int(LCP.max().item()) >= 1_999_000          # the 2e6-long single-letter block


def test_32_bit_keys_opt_in_device(L, monkeypatch):
    """CAPS_SA_KEYS=32 at BASELINE config 1's size and on 60 M bases with 64-bit indices: exact device verifier, and the same
    arrays as the default (64-bit keys) build."""
    import torch
    for n, p, bits in [(268_435_457, 8000, 32), (60_000_001, 200, 64)]:
        g = torch.Generator(device="cuda")
        g.manual_seed(n % 1000)
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
        T = lut[torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int64)]
        dt = torch.int32 if bits == 32 else torch.int64
        out = {}
        for keys in ("32", "64"):
            monkeypatch.setenv("CAPS_SA_KEYS", keys)
            SA = torch.empty(n, dtype=dt, device="cuda")
            LCP = torch.empty(n, dtype=dt, device="cuda")
            st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=p, idx_bits=bits)
            assert st["path_direct"] == 1 and st["direct_key_bits"] == int(keys), st
            assert L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), idx_bits=bits) == 0
            out[keys] = (SA, LCP)
        assert torch.equal(out["32"][0], out["64"][0]) and torch.equal(out["32"][1], out["64"][1])


def test_alphabet_guessed_from_a_sample_is_validated(L, oracle):
    """prepare_text packs with the alphabet of the text's first MiB and validates every byte while packing: a symbol that
    only shows up later (here the 5th, which also changes the code width) sends the preparation to the exact path."""
    rs = np.random.RandomState(77)
    T = rs.choice(DNA, size=20_000_000)
    T[19_000_000] = ord("N")
    st = _same(L, oracle, T, 300)
    assert st["bits_per_char"] == 8
    T = rs.choice(DNA, size=20_000_000)
    T[:3_000_000] = rs.choice(np.frombuffer(b"AC", dtype=np.uint8), size=3_000_000)       # the sample sees two of the four symbols
    st = _same(L, oracle, T, 300)
    assert st["bits_per_char"] == 2


def test_a_stream_that_outgrows_its_region_falls_back_device(L, oracle, monkeypatch):
    """CAPS_SA_FB_GROUP_OVERFLOW on the real kernels (pipeline.h run_direct: group_scatter_kernel drops what does not fit a
    region and reports the largest stream; shard.h plan(): every rank takes the samplesort sequence together).
    CAPS_SA_TEST_STREAM_CAP shrinks the regions to 80 % of the mean stream, so level A overflows on any text."""
    rs = np.random.RandomState(41)
    T = rs.choice(DNA, size=6_000_000)
    SAo, LCPo = oracle.build_sa_lcp(T, p=500)
    assert L.build(T, p=500)[2]["path_direct"] == 1
    monkeypatch.setenv("CAPS_SA_TEST_STREAM_CAP", "80")
    for mode in ("linear", "quantile"):
        monkeypatch.setenv("CAPS_SA_DIRECT_MODE", mode)
        SA, LCP, st = L.build(T, p=500)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), mode
        assert st["path_direct"] == 0 and st["path_fallback"] == 5, (mode, st["path_fallback"])
    monkeypatch.delenv("CAPS_SA_DIRECT_MODE")
    for exchange in (None, "1"):
        if exchange:
            monkeypatch.setenv("CAPS_SA_SHARD_EXCHANGE", exchange)
        SA, LCP, st = L.build_multi(T, [0, 0, 0], p=500)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), exchange
        assert st["path_direct"] == 0 and st["path_fallback"] == 5, (exchange, st["path_fallback"])


def test_results_leave_in_waves_host_path(L, oracle, monkeypatch):
    """the wave-streamed host build (capi_impl.h HostCopySink + pipeline.h set_waves) is what every host / CLI build from
    400 Mi chars on runs; CAPS_SA_HOST_WAVES forces it on texts the oracle checks in seconds.  Real streams here: the copies of
    a finished slice run on a second stream while the next wave is sorted, into page-locked and into pageable arrays.  Uniform
    keys (linear buckets), skewed keys (quantile buckets), N-block stand-ins (letter-run buckets inside a wave), 64-bit
    indices, 8-bit codes, and a group larger than the scratch of one wave (then: one wave)."""
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from genome_like import markov_dna
    rs = np.random.RandomState(43)
    uni = rs.choice(DNA, size=6_000_001)
    skew = markov_dna(5_000_000, seed=17).cpu().numpy()
    runs = rs.choice(DNA, size=8_000_000)
    runs[1_000_000:1_060_000] = ord("G")
    runs[5_000_000:5_009_000] = ord("G")
    txt = rs.choice(np.frombuffer(b"abcdefghijklmnopqrst", dtype=np.uint8), size=5_000_000, p=np.array([0.75 ** i for i in range(20)]) / sum(0.75 ** i for i in range(20)))
    cases = [("uniform", uni, 32), ("skewed", skew, 32), ("runs", runs, 32), ("uniform64", uni[:3_000_001], 64), ("text8", txt, 32)]
    L.release_cache()
    for name, T, bits in cases:
        SAo, LCPo = oracle.build_sa_lcp(T, p=1000, idx_bits=bits)
        for waves, pinned in (("2", True), ("3", False), ("5", True)):
            monkeypatch.setenv("CAPS_SA_HOST_WAVES", waves)
            SA, LCP, st = L.build(T, p=1000, idx_bits=bits, pinned=pinned)
            assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (name, waves)
            assert st["path_direct"] == 1 and 2 <= st["result_waves"] <= int(waves) + 1, (name, waves, st["result_waves"])
        if name in ("skewed", "runs"):
            assert st["direct_quantile"] == 1, name
        if name == "runs":
            assert st["long_runs"] == 1 and st["run_buckets"] >= 1
        if name == "text8":
            assert st["bits_per_char"] == 8
    big = rs.choice(DNA, size=6_000_000)
    big[2_000_000:3_200_000] = ord("G")                   # one group holds a fifth of the text: more than one of twelve waves' scratch
    monkeypatch.setenv("CAPS_SA_HOST_WAVES", "12")
    SA, LCP, st = L.build(big, p=1000)
    dT = torch.from_numpy(big).cuda()
    dSA, dLCP = torch.from_numpy(SA.astype(np.int32)).cuda(), torch.from_numpy(LCP.astype(np.int32)).cuda()
    assert L.verify_device(dT.data_ptr(), big.size, dSA.data_ptr(), dLCP.data_ptr()) == 0
    assert st["path_direct"] == 0 or st["result_waves"] == 1, st["result_waves"]
    L.release_cache()


@pytest.mark.gpu
def test_quantile_split_without_a_count_pass_device(L, oracle, monkeypatch):
    """VERDICT r4 item 1 (quantile level B): the split by knots runs without its count pass -- slots, a spill stream in the
    caller's SA / LCP slice for what outgrows them, the outgrown buckets put together in the compact array (pipeline.h
    "speculative split by knots"; kernels.h Spill).  Against the oracle: skewed keys, N-block stand-ins (letter-run buckets and
    long runs on the stream), repeats (deferred ties), 64-bit indices, waves, slots made small (most buckets outgrow them),
    a stream that runs full (the count split takes over), and the count split itself -- the same arrays every way."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from genome_like import markov_dna
    keys = ("CAPS_SA_DIRECT_MODE", "CAPS_SA_DIRECT_SUB", "CAPS_SA_TEST_SPILL_SLOT", "CAPS_SA_TEST_SPILL_CAP", "CAPS_SA_NO_SPILL_SLOTS",
            "CAPS_SA_HOST_WAVES")

    def run(T, want, bits=32, direct=True, **env):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv("CAPS_SA_" + k, v)
        SA, LCP, st = L.build(T, p=1000, idx_bits=bits)
        assert np.array_equal(SA, want[0]) and np.array_equal(LCP, want[1]), env
        assert not direct or (st["path_direct"] == 1 and st["direct_quantile"] == 1), env
        return st

    rs = np.random.RandomState(91)
    skew = markov_dna(5_000_000, seed=23).cpu().numpy()
    want = oracle.build_sa_lcp(skew, p=1000)
    st = run(skew, want)
    assert st["knot_slot_splits"] == 1 and st["knot_slot_splits_redone"] == 0 and st["bucket_count_ms"] == 0.0, st
    assert 0 < st["spill_entries"] < skew.size // 8
    base = st["spill_entries"]
    st = run(skew, want, NO_SPILL_SLOTS="1")
    assert st["knot_slot_splits"] == 0 and st["bucket_count_ms"] > 0.0
    st = run(skew, want, TEST_SPILL_SLOT="2816")                            # slots below the mean bucket (3072)
    assert st["knot_slot_splits"] == 1 and st["spill_entries"] > 4 * base
    st = run(skew, want, TEST_SPILL_SLOT="512")                             # ... and so small that the stream runs full
    assert st["knot_slot_splits"] == 0 and st["knot_slot_splits_redone"] == 1
    st = run(skew, want, TEST_SPILL_CAP="1000")
    assert st["knot_slot_splits"] == 0 and st["knot_slot_splits_redone"] == 1
    for waves in ("2", "5"):
        st = run(skew, want, HOST_WAVES=waves)
        assert st["result_waves"] >= 2 and st["knot_slot_splits"] == st["result_waves"]
        st = run(skew, want, HOST_WAVES=waves, TEST_SPILL_SLOT="2816")
        assert st["knot_slot_splits"] == st["result_waves"]
    want64 = oracle.build_sa_lcp(skew[:3_000_001], p=1000, idx_bits=64)
    st = run(skew[:3_000_001], want64, bits=64)
    assert st["knot_slot_splits"] == 1
    run(skew[:3_000_001], want64, bits=64, TEST_SPILL_SLOT="2816")
    runs = rs.choice(DNA, size=8_000_000)
    runs[1_000_000:1_060_000] = ord("G")
    runs[5_000_000:5_009_000] = ord("G")
    want = oracle.build_sa_lcp(runs, p=1000)
    run(runs, want, direct=False)               # (8 sub-streams at this size: the 60,000-char run may overload some -> the samplesort path)
    st = run(runs, want, DIRECT_SUB="1")        # one stream per group: the fat group's region holds the run
    assert st["long_runs"] == 1 and st["run_buckets"] >= 1 and st["knot_slot_splits"] == 1 and st["spill_entries"] > 50_000
    st = run(runs, want, DIRECT_SUB="1", HOST_WAVES="3")
    assert st["knot_slot_splits"] >= 2
    run(runs, want, DIRECT_SUB="1", TEST_SPILL_SLOT="2816")
    rep = markov_dna(6_000_000, seed=29).cpu().numpy()
    unit = rep[1000:1171].copy()
    for k in range(3000):                                                   # a tandem array with a mutation here and there
        rep[2_000_000 + 171 * k:2_000_000 + 171 * (k + 1)] = unit
    mut = rs.randint(2_000_000, 2_000_000 + 171 * 3000, size=4000)
    rep[mut] = rs.choice(DNA, size=mut.size)
    rep[5_000_000:5_040_000] = rep[100_000:140_000]                         # an exact 40-kb duplicate
    want = oracle.build_sa_lcp(rep, p=1000)
    st = run(rep, want)
    assert st["knot_slot_splits"] == 1
    run(rep, want, TEST_SPILL_SLOT="2816")
    run(rep, want, NO_SPILL_SLOTS="1")
    # uniform keys in quantile mode: the queue chain starts at tile_sort_kernel, which sees the tiles of outgrown buckets empty
    uni = rs.choice(DNA, size=5_000_001)
    want = oracle.build_sa_lcp(uni, p=1000)
    st = run(uni, want, DIRECT_MODE="quantile")
    assert st["knot_slot_splits"] == 1
    st = run(uni, want, DIRECT_MODE="quantile", TEST_SPILL_SLOT="2816")
    assert st["knot_slot_splits"] == 1 and st["spill_entries"] > 200_000
    L.release_cache()


@pytest.mark.gpu
def test_lcp_leaves_the_device_as_bytes_host_path(L, oracle, monkeypatch):
    """on the link the LCP array travels as bytes (capi_impl.h HostCopySink, lcp_narrow_kernel) -- values below
    255 as they are, the others as (position, value) pairs at the end -- and host threads widen them into the caller's array while
    the next slice is copied.  Forced here at sizes the oracle checks (it is the default from 400 Mi chars on): no exception at all,
    a few thousand (single-letter runs), more than the list holds (then the whole array at full width), page-locked and pageable
    result arrays, several worker counts."""
    from sa_check import sa_lcp
    rs = np.random.RandomState(91)
    uni = rs.choice(DNA, size=5_000_001)
    runs = rs.choice(DNA, size=6_000_000)
    runs[1_000_000:1_003_000] = ord("G")
    runs[4_000_000:4_000_900] = ord("A")
    flood = rs.choice(DNA, size=6_000_000)
    flood[2_000_000:2_400_000] = ord("G")                  # 400,000 values above 254: the list holds n / 32
    L.release_cache()
    monkeypatch.setenv("CAPS_SA_HOST_NARROW_LCP", "1")
    for name, T, want_bytes in (("uniform", uni, 1), ("runs", runs, 1), ("flood", flood, 4)):
        SAo, LCPo = oracle.build_sa_lcp(T, p=1000) if name == "uniform" else sa_lcp(T, 32)
        for waves, pinned, threads in (("3", True, "1"), ("4", False, "3"), ("6", True, "8")):
            monkeypatch.setenv("CAPS_SA_HOST_WAVES", waves)
            monkeypatch.setenv("CAPS_SA_HOST_THREADS", threads)
            SA, LCP, st = L.build(T, p=1000, pinned=pinned)
            assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (name, waves)
            assert st["path_direct"] == 1 and st["result_waves"] >= 2 and st["lcp_bytes_on_link"] == want_bytes, (name, st["lcp_bytes_on_link"])
    monkeypatch.setenv("CAPS_SA_HOST_NARROW_LCP", "0")
    SAo, LCPo = sa_lcp(runs, 32)
    SA, LCP, st = L.build(runs, p=1000)
    assert st["lcp_bytes_on_link"] == 4 and np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    L.release_cache()


@pytest.mark.gpu
def test_a_lost_tie_note_sends_the_tile_to_the_comparison_sort_device(L, oracle, monkeypatch):
    """tile_sort_eq_kernel's plain build reads the lcp of two neighbours with EQUAL keys from the note the tie
    phases left; if a note were ever lost it would emit the capped key length -- a wrong LCP nothing but the verifier would see.
    CAPS_SA_TEST_DROP_NOTE makes it lose the notes of all tied pairs: the emit phase must notice (equal keys, no note), queue the
    tile for the comparison sort, and the arrays must still be THE arrays."""
    rs = np.random.RandomState(77)
    n = 1_500_000
    T = rs.choice(DNA, size=n, p=[0.4, 0.1, 0.15, 0.35])
    for _ in range(n // 1500):                               # pairs of equal 32-mers: a copy with one mutation
        s, d = rs.randint(0, n - 90, size=2)
        T[d:d + 90] = T[s:s + 90]
        T[d + rs.randint(40, 90)] = DNA[rs.randint(0, 4)]
    SAo, LCPo = oracle.build_sa_lcp(T, p=200)
    for mode in ("quantile", "linear"):
        monkeypatch.setenv("CAPS_SA_DIRECT_MODE", mode)
        monkeypatch.setenv("CAPS_SA_TEST_DROP_NOTE", "1")
        SA, LCP, st = L.build(T, p=200)
        assert st["path_direct"] == 1
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), mode
        monkeypatch.delenv("CAPS_SA_TEST_DROP_NOTE")
        SA, LCP, st = L.build(T, p=200)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), mode


def _repeat_rich(rs, n, alphabet=DNA):
    """GRCh38-shaped repeat content at oracle-checkable size (tools/genome_like.py plant_genome_repeats in small): a tandem array of
    a 171-char monomer at 2 %, a higher-order array (12 monomers 20 % apart, units 0.7 % apart), 3000 copies of a 300-char family
    at 10 %, one exact 20-kb duplicate, and a text that ends inside the tandem array's content."""
    T = rs.choice(alphabet, size=n)

    def mutate(seg, rate):
        seg = seg.copy()
        m = rs.rand(seg.size) < rate
        seg[m] = rs.choice(alphabet, size=int(m.sum()))
        return seg
    mono = rs.choice(alphabet, size=171)
    a0, half = n // 10, n // 80
    T[a0:a0 + half] = mutate(np.tile(mono, half // 171 + 1)[:half], 0.02)
    unit = np.concatenate([mutate(mono, 0.2) for _ in range(12)])
    T[a0 + half:a0 + 2 * half] = mutate(np.tile(unit, half // unit.size + 1)[:half], 0.007)
    cons = rs.choice(alphabet, size=300)
    for pos in rs.randint(n // 3, n - n // 8, size=400):
        T[pos:pos + 300] = mutate(cons, 0.10)
    T[n - n // 16:n - n // 16 + 20_000] = T[n // 50:n // 50 + 20_000]
    T[n - 2000:] = np.tile(mono, 12)[:2000]
    return T


def test_large_groups_of_equal_keys_are_rekeyed_not_compared_device(L, oracle, monkeypatch):
    """Deferred ties on the real kernels (kernels.h "Deferred ties", pipeline.h msd_refine) against the oracle: tandem arrays, a
    repeat family, an exact duplicate and a text that ends inside a repeat -- thousands of suffixes share their key, the tiles that
    hold them and the buckets larger than a tile leave sentinel LCPs, msd_refine orders the groups level by level.  2-bit and
    8-bit codes, both index widths, both level-B modes; and the same arrays with CAPS_SA_NO_DEFER (every tie compared)."""
    rs = np.random.RandomState(53)
    for n, alphabet, bits in [(4_000_000, DNA, 32), (3_000_001, DNA, 64), (3_000_000, np.frombuffer(b"acgtn\x80\xfe", dtype=np.uint8), 32)]:
        T = _repeat_rich(rs, n, alphabet)
        SAo, LCPo = oracle.build_sa_lcp(T, p=800, idx_bits=bits)
        for mode in (None, "quantile", "linear"):
            if mode:
                monkeypatch.setenv("CAPS_SA_DIRECT_MODE", mode)
            else:
                monkeypatch.delenv("CAPS_SA_DIRECT_MODE", raising=False)
            SA, LCP, st = L.build(T, p=800, idx_bits=bits)
            assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (n, bits, mode)
            assert st["path_direct"] == 1
            if alphabet is DNA:     # (8-char keys of a byte alphabet tie all the time: most tiles would be deferred, so none is -- segmented_sort defer_check)
                assert st["tie_groups_deferred"] > 0, (n, mode, st["tie_groups_deferred"], st["tie_levels"])
        monkeypatch.delenv("CAPS_SA_DIRECT_MODE", raising=False)
        monkeypatch.setenv("CAPS_SA_NO_DEFER", "1")
        SA, LCP, st = L.build(T, p=800, idx_bits=bits)
        monkeypatch.delenv("CAPS_SA_NO_DEFER")
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo) and st["tie_groups_deferred"] == 0


def test_groups_larger_than_a_tile_take_the_level_loop_device(L):
    """A group of equal keys of at most a tile of members is finished by one workgroup in LDS (msd_quick_kernel); a larger one goes
    through msd_refine's level loop until its parts fit.  A tandem array of a 23-char unit with 9000 copies (groups of ~9000
    members > 4096), a text that ends inside an exact array; both index widths; against tests/sa_check.py (the oracle is
    quadratic on such texts)."""
    from sa_check import sa_lcp
    rs = np.random.RandomState(71)
    for n, copies, bits in [(1_500_000, 9000, 32), (1_000_001, 7000, 64)]:
        T = rs.choice(DNA, size=n)
        for at, k, rate in [(n // 3, copies, 0.002), (n - 23 * 300, 300, 0.0)]:
            unit = rs.choice(DNA, size=23)
            seg = np.tile(unit, k)
            m = rs.rand(seg.size) < rate
            seg[m] = rs.choice(DNA, size=int(m.sum()))
            T[at:at + seg.size] = seg
        SAo, LCPo = sa_lcp(T, idx_bits=bits)
        SA, LCP, st = L.build(T, p=0, idx_bits=bits)
        assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo), (n, bits)
        assert st["path_direct"] == 1 and st["tie_groups_deferred"] > 0 and st["tie_levels"] >= 2, (n, st["tie_groups_deferred"], st["tie_levels"])
    # an EXACT array of 9000 copies would take a level per 32 chars of the array (~6,500): past MSD_MAX_LEVELS the build is done
    # again without deferring (the comparators finish such groups)
    T = rs.choice(DNA, size=1_200_000)
    T[400_000:400_000 + 23 * 9000] = np.tile(rs.choice(DNA, size=23), 9000)
    SAo, LCPo = sa_lcp(T)
    SA, LCP, st = L.build(T, p=0)
    assert np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    assert st["path_direct"] == 1 and st["tie_groups_deferred"] == 0, (st["tie_groups_deferred"], st["tie_levels"])
