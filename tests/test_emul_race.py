"""CPU: the barrier-race detector of the emulation (tests/emul/race_rt.h) and the poison-filled emulation.

The plain emulation runs the phases of a kernel one after the other, so a hand-off between two threads of a workgroup WITHOUT a
barrier in between cannot show in its results.  The race build instruments every LDS access (g++ -fsanitize=thread hooks, own
runtime) and reports such hand-offs; here: its self-test (it must see the seeded races and only those), and real builds of both
constructions over several text shapes, which must be exact AND race-free.  The poison build starts every LDS array and every
per-thread register as 0xA5 bytes (on the GPU they hold what the last workgroup left): results must not change."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import text_bytes
from emul_util import EMUL_DIR, ROOT

DNA = np.frombuffer(b"ACGT", dtype=np.uint8)


def _make(name):
    subprocess.check_call(["make", "-s", "-C", EMUL_DIR, name])
    return os.path.join(EMUL_DIR, name)


def _load(name):
    import caps_sa_amd
    path = _make(name)
    return caps_sa_amd.CapsLib(path, "caps_sa_emul_"), ctypes.CDLL(path)


def test_the_detector_sees_seeded_races_and_only_those():
    L = ctypes.CDLL(_make("librace_selftest.so"))
    L.caps_race_selftest.restype = ctypes.c_ulonglong
    found = [int(L.caps_race_selftest(m)) for m in range(7)]
    # 0: hand-off across a barrier; 1: the same without it; 2: write after another thread's read; 3: different values into one word;
    # 4: every writer stores the same value; 5: atomics only; 6: a plain read of a word others bump atomically
    assert found[0] == 0 and found[4] == 0 and found[5] == 0, found
    assert found[1] >= 60 and found[2] >= 60 and found[3] >= 60 and found[6] >= 60, found


def _texts():
    rs = np.random.RandomState(7)
    n = 60000
    yield "uniform", rs.choice(DNA, size=n), 20, 32
    yield "skew", rs.choice(DNA, size=n, p=rs.dirichlet([0.4] * 4)), 13, 32
    T = rs.choice(DNA, size=n)
    for _ in range(5):
        ln = int(rs.randint(100, n // 8))
        a, b = rs.randint(0, n - ln, size=2)
        T[b:b + ln] = T[a:a + ln]
    yield "planted", T, 16, 32
    T = rs.choice(DNA, size=n)
    T[1000:9000] = T[1000]
    yield "runs", T, 16, 32
    yield "bytes", rs.randint(0, 200, size=n).astype(np.uint8), 9, 32
    T = rs.choice(DNA, size=n)
    T[5000:5000 + 171 * 120] = np.tile(rs.choice(DNA, size=171), 120)
    yield "tandem", T, 16, 32
    yield "u64", rs.choice(DNA, size=50001), 12, 64
    # the text on which two builds of the product sources with -mllvm -amdgpu-spill-sgpr-to-vgpr=false went wrong
    yield "missort_skew_40240", np.load(os.path.join(ROOT, "tests", "golden", "missort_skew_40240.npy")), 16, 32


@pytest.mark.slow
def test_builds_are_exact_and_free_of_barrier_races(oracle, monkeypatch):
    E, raw = _load("libcaps_sa_emul_small_race.so")
    raw.caps_sa_emul_races_found.restype = ctypes.c_ulonglong
    for name, T, p, bits in _texts():
        sa, lcp = oracle.build_sa_lcp(T, p=p, idx_bits=bits)[:2]
        for path in ("auto", "classic"):
            monkeypatch.setenv("CAPS_SA_PATH", path)
            SA, LCP, _ = E.build(T, p=p, idx_bits=bits)
            races = int(raw.caps_sa_emul_races_found())
            raw.caps_sa_emul_races_reset()
            assert np.array_equal(SA, sa) and np.array_equal(LCP, lcp), (name, path)
            assert races == 0, (name, path, races)


@pytest.mark.slow
def test_poisoned_lds_and_registers_change_nothing(oracle, golden_cases, monkeypatch):
    E, _ = _load("libcaps_sa_emul_small_poison.so")          # (also: the threads of every phase in a scattered order)
    for name, T, p, bits in _texts():
        sa, lcp = oracle.build_sa_lcp(T, p=p, idx_bits=bits)[:2]
        for path in ("auto", "classic"):
            monkeypatch.setenv("CAPS_SA_PATH", path)
            SA, LCP, _ = E.build(T, p=p, idx_bits=bits)
            assert np.array_equal(SA, sa) and np.array_equal(LCP, lcp), (name, path)
    for c in golden_cases:                                    # the reference-made vectors
        T = text_bytes(c["text"])
        if T.size < 32:
            continue
        SA, LCP, _ = E.build(T, p=0)
        assert SA.tolist() == c["sa"] and LCP.tolist() == c["lcp"], c["name"]
