"""CPU: the oracle (C restatement of the reference algorithm) against the golden vectors
recorded from the reference's own chatgpt_baseline.py (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import LARGE_GOLDEN, large_golden, text_bytes


def _cases(golden_cases):
    return {c["name"]: c for c in golden_cases}


def test_golden_file_has_reference_data_case(golden_cases):
    names = {c["name"] for c in golden_cases}
    assert {"simpletest2_cli", "banana_cli", "unary_1000", "lowercase_5000"} <= names


@pytest.mark.parametrize("p", [0, 2, 3, 7, 16])
def test_oracle_matches_golden(oracle, golden_cases, p):
    ran = 0
    for c in golden_cases:
        T = text_bytes(c["text"])
        n = T.size
        p_eff = min(p if p else 8192, n // 16)
        if n < 32 or p_eff < 2:
            with pytest.raises(ValueError):      # reference: SIGFPE / undefined
                oracle.build_sa_lcp(T, p=p)
            continue
        SA, LCP = oracle.build_sa_lcp(T, p=p)
        assert SA.tolist() == c["sa"], c["name"]
        assert LCP.tolist() == c["lcp"], c["name"]
        ran += 1
    assert ran >= 15


@pytest.mark.parametrize("name", LARGE_GOLDEN)
def test_oracle_matches_large_golden(oracle, name):
    """The 136k .. 250k-char reference-made vectors (incl. bytes >= 0x80: the signed-char order of src/Suffix_Array.cpp:75-77
    now has a reference-held pin, see make_golden.py): oracle, default p and p = 8000-style small subarrays."""
    T, sa, lcp = large_golden(name)
    for p in (0, 37):
        SA, LCP = oracle.build_sa_lcp(T, p=p)
        assert np.array_equal(SA, sa), (name, p)
        assert np.array_equal(LCP, lcp), (name, p)


def test_naive_matches_golden_everywhere(oracle, golden_cases):
    for c in golden_cases:
        T = text_bytes(c["text"])
        SA, LCP = oracle.naive_sa_lcp(T)
        assert SA.tolist() == c["sa"], c["name"]
        assert LCP.tolist() == c["lcp"], c["name"]
        assert oracle.check(T, SA, LCP) == 0


def test_u64_variant_matches(oracle, golden_cases):
    c = _cases(golden_cases)["gen_rand_seq_7_4096_cli"]
    T = text_bytes(c["text"])
    SA, LCP = oracle.build_sa_lcp(T, p=13, idx_bits=64)
    assert SA.dtype == np.uint64
    assert SA.tolist() == c["sa"] and LCP.tolist() == c["lcp"]


def test_simpletest2_input_file_is_the_reference_data(oracle, golden_cases):
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    raw = open(os.path.join(here, "golden", "simpletest2.input"), "rb").read()
    assert len(raw) == 118
    T = oracle.remap(raw)
    assert T.tobytes().decode() == _cases(golden_cases)["simpletest2_cli"]["text"]


def test_closed_forms(oracle):
    n = 5000
    T = np.full(n, ord("a"), dtype=np.uint8)
    SA, LCP = oracle.build_sa_lcp(T, p=16)
    assert np.array_equal(SA, np.arange(n - 1, -1, -1, dtype=np.uint32))
    assert np.array_equal(LCP, np.arange(n, dtype=np.uint32))


def test_signed_char_order(oracle):
    # bytes >= 0x80 sort BEFORE ASCII (char is signed: Suffix_Array.cpp:77,289)
    rs = np.random.RandomState(5)
    T = rs.choice(np.array([0x41, 0x7F, 0x80, 0xFF, 0x00], dtype=np.uint8), size=2000)
    SA, LCP = oracle.build_sa_lcp(T, p=5)
    SA2, LCP2 = oracle.naive_sa_lcp(T)
    assert np.array_equal(SA, SA2) and np.array_equal(LCP, LCP2)
    first = T[SA[0]]
    assert first in (0x80, 0xFF)        # a negative char leads
    assert oracle.check(T, SA, LCP) == 0


def test_unit_entry_points(oracle):
    rs = np.random.RandomState(1)
    T = rs.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=3000)
    idx = rs.permutation(3000)[:1500].astype(np.uint32)
    Y, L = oracle.merge_sort(T, idx)
    SAn, _ = oracle.naive_sa_lcp(T)
    keep = np.isin(SAn, idx)
    assert np.array_equal(Y, SAn[keep])
    for i in range(1, Y.size):
        assert L[i] == oracle.lcp(T, int(Y[i - 1]), int(Y[i]))
    # merge of two sorted halves
    a, la = oracle.merge_sort(T, idx[:700])
    b, lb = oracle.merge_sort(T, idx[700:])
    Z, LZ = oracle.merge(T, a, b, la, lb)
    assert np.array_equal(Z, Y) and np.array_equal(LZ, L)
    # upper_bound
    rank = {int(s): r for r, s in enumerate(SAn)}
    for piv in (0, 17, 2999, int(Y[10])):
        ub = oracle.upper_bound(T, Y, piv)
        assert ub == sum(1 for s in Y if rank[int(s)] <= rank[piv])
