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
