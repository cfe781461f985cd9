"""CPU: the run table (csrc/text.h) -- constant-time skipping of periodic stretches in deep
comparisons -- through the host emulation of the kernels, against an independent
prefix-doubling construction (tests/sa_check.py: the oracle follows the reference and is
quadratic on such inputs).  Covers both builds of the comparator kernels: with the table
(stretches >= 1024 chars) and plain (shorter ones)."""
import numpy as np
import pytest

from emul_util import emul, emul_small
from sa_check import sa_lcp

DNA = np.frombuffer(b"ACGT", dtype=np.uint8)


def _b(s):
    return np.frombuffer(s, dtype=np.uint8)


def _chk(E, T, p, bits=32, long_runs=None):
    SA, LCP, st = E.build(T, p=p, idx_bits=bits)
    SAo, LCPo = sa_lcp(T, bits)
    assert np.array_equal(SA, SAo), f"SA mismatch n={T.size} p={p}"
    assert np.array_equal(LCP, LCPo), f"LCP mismatch n={T.size} p={p}"
    if long_runs is not None:
        assert st["long_runs"] == int(long_runs)
    return st


def test_sa_check_agrees_with_the_oracle(oracle):
    rs = np.random.RandomState(1)
    for T in (rs.choice(DNA, size=3000), rs.randint(0, 256, size=3000).astype(np.uint8),
              np.tile(_b(b"ACCA"), 300), np.full(500, 0x80, np.uint8)):
        a, b = sa_lcp(T), oracle.naive_sa_lcp(T)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("which", ["big", "small"])
def test_unary_and_short_periods(which):
    E = emul() if which == "big" else emul_small()
    rs = np.random.RandomState(2)
    n = 60000
    SA, LCP, st = E.build(np.full(n, ord("A"), np.uint8), p=4)
    assert st["long_runs"] == 1
    assert np.array_equal(SA, np.arange(n - 1, -1, -1, dtype=np.uint32))       # SURVEY 0.8 closed form
    assert np.array_equal(LCP, np.arange(n, dtype=np.uint32))
    _chk(E, np.full(20001, 0x80, np.uint8), 0, long_runs=True)                 # code 0 = the end-of-text padding
    _chk(E, np.tile(_b(b"AC"), 30000), 3, long_runs=True)
    _chk(E, np.tile(rs.choice(DNA, size=16), 3000), 5, long_runs=True)         # the longest period the table knows
    _chk(E, np.tile(rs.choice(DNA, size=17), 600), 5, long_runs=False)         # beyond it: plain comparators
    _chk(E, np.tile(_b(b"abcd"), 8000), 4, long_runs=True)                     # 8-bit codes: periods <= 4
    _chk(E, np.tile(_b(b"abcde"), 1500), 4, long_runs=False)


@pytest.mark.parametrize("which", ["big", "small"])
def test_runs_planted_in_random_text(which):
    E = emul() if which == "big" else emul_small()
    rs = np.random.RandomState(3)
    parts = [np.full(5000, ord("G"), np.uint8), rs.choice(DNA, size=7000), np.tile(_b(b"ACG"), 3000), rs.choice(DNA, size=100),
             np.full(9000, ord("G"), np.uint8), rs.choice(DNA, size=3), np.tile(_b(b"ACG"), 2000), np.full(4000, ord("A"), np.uint8)]
    _chk(E, np.concatenate(parts), 7, long_runs=True)                          # run at the start, run at the end
    _chk(E, np.concatenate(parts[::-1]), 0, long_runs=True)
    # only short stretches (< 1024 chars): the table is built but the plain kernels run
    short = [rs.choice(DNA, size=3000), np.full(700, ord("T"), np.uint8), rs.choice(DNA, size=2000), np.tile(_b(b"GA"), 400),
             rs.choice(DNA, size=1500), np.full(900, ord("T"), np.uint8)]
    _chk(E, np.concatenate(short), 5, long_runs=False)
    L = _b(b"abcdefgh")
    bts = [np.full(3000, ord("z"), np.uint8), rs.choice(L, size=5000), np.tile(_b(b"ab"), 4000), rs.choice(L, size=50),
           np.tile(_b(b"abc"), 3000), np.full(6001, ord("z"), np.uint8), np.tile(_b(b"abcd"), 2500), np.tile(_b(b"abcde"), 2000)]
    _chk(E, np.concatenate(bts), 6, long_runs=True)
    _chk(E, np.concatenate(bts), 6, bits=64, long_runs=True)


def test_stretch_ends_at_every_offset_of_a_block():
    """Both ends of a planted stretch at every alignment relative to the 32-char blocks, and every period up to 16."""
    E = emul_small()
    rs = np.random.RandomState(4)
    for d in (1, 2, 3, 5, 8, 15, 16):
        unit = rs.choice(DNA, size=d)
        while d > 1 and len(set(unit.tolist())) == 1:
            unit = rs.choice(DNA, size=d)
        parts = []
        for off in range(0, 33, 3):
            parts += [rs.choice(DNA, size=40 + off), np.tile(unit, (1100 + 7 * off) // d + 1)[:1100 + 7 * off]]
        _chk(E, np.concatenate(parts), 3, long_runs=True)


def _n_block_text(rs, scale):
    """Random DNA with runs of every letter (A = the code of the end-of-text padding), one of them at the end of the text,
    two G runs of one length followed by the same 40 chars, and a run shorter than a tile."""
    n = 20000 * scale
    T = rs.choice(DNA, size=n)
    for at, ln, c in ((1000, 1500, "G"), (5000, 700, "G"), (9000, 900, "A"), (12000, 800, "T"), (15000, 500, "C")):
        T[at * scale:(at + ln) * scale] = ord(c)
    T[n - 600 * scale:] = ord("A")
    fl = rs.choice(DNA, size=40)
    fl[0] = ord("A")
    for a in (17000 * scale, 18000 * scale):
        T[a:a + 400 * scale] = ord("G")
        T[a + 400 * scale:a + 400 * scale + 40] = fl
    T[300:300 + 90] = ord("G")
    return T


@pytest.mark.parametrize("which", ["big", "small"])
def test_letter_run_buckets_order_n_blocks_without_comparing(which, monkeypatch):
    """text.h "letter runs": the suffixes deep inside single-letter runs (N-blocks: the CLI maps N to G, src/main.cpp:61-68)
    get a bucket of their own and are ordered by (terminator class, rest of the run, text behind it); the LCP-merge passes
    they used to need are gone.  Same arrays with the buckets switched off; 64-bit indices; 8-bit codes."""
    E = emul() if which == "big" else emul_small()
    scale = 16 if which == "big" else 1
    T = _n_block_text(np.random.RandomState(3), scale)
    for p in (0, 64):
        st = _chk(E, T, p, long_runs=True)
        assert st["path_direct"] == 1 and st["direct_quantile"] == 1 and st["run_buckets"] >= 3, st
        assert st["merge_passes_phase2"] <= 1
    st64 = _chk(E, T, 0, bits=64, long_runs=True)
    assert st64["run_buckets"] >= 3
    monkeypatch.setenv("CAPS_SA_NO_RUN_BUCKETS", "1")
    st0 = _chk(E, T, 0, long_runs=True)
    assert st0["run_buckets"] == 0 and st0["merge_passes_phase2"] >= 2
    monkeypatch.delenv("CAPS_SA_NO_RUN_BUCKETS")
    # 8-bit codes: runs of the smallest byte (0x80: the padding code), of the largest, and of a letter in between
    B = rs_bytes = np.random.RandomState(5).choice(np.array([0x80, 0x41, 0x61, 0x7F, 0xFF, 0x00], dtype=np.uint8), size=20000 * scale)
    for at, ln, c in ((2000, 1200, 0x80), (6000, 1000, 0x7F), (10000, 1100, 0x61)):
        B[at * scale:(at + ln) * scale] = c
    B[B.size - 500 * scale:] = 0x80
    st8 = _chk(E, B, 0, long_runs=True)
    assert st8["bits_per_char"] == 8
    assert st8["run_buckets"] >= 1 or st8["path_direct"] == 0, st8


@pytest.mark.parametrize("which", ["big", "small"])
def test_exact_long_duplicates_are_settled_by_the_third_tie_stage(which):
    """tile_sort_eq_kernel<VDEEP>: a pair of suffixes that agree on more than the first two tie rounds see (2,300 bases: an exact
    duplicate of several kb, a copy that runs into the end of the text, a triple) is settled by the whole workgroup, any depth,
    instead of sending its tile to the comparison sort."""
    E = emul() if which == "big" else emul_small()
    scale = 10 if which == "big" else 1
    rs = np.random.RandomState(4)
    n = 40000 * scale
    T = rs.choice(DNA, size=n)
    T[30000 * scale:35000 * scale] = T[2000 * scale:7000 * scale]                 # an exact duplicate: lcps up to 5000 * scale
    T[20000 * scale:20000 * scale + 3000] = T[10000 * scale:10000 * scale + 3000]
    T[20000 * scale + 1500] = ord("A") if T[20000 * scale + 1500] != ord("A") else ord("C")
    T[33000 * scale:33000 * scale + 2500] = T[5000:7500]                          # three copies of one piece
    T[36500 * scale:36500 * scale + 2500] = T[5000:7500]
    for TT in (T, np.concatenate([T, T[:7000]])):                                 # ... and a duplicate that ends with the text
        for p in (0, 16):
            _chk(E, TT, p)
