"""CPU: pin the oracle to outputs of the REAL reference.

The digests below are sha256 of dump files (Suffix_Array::dump format,
src/Suffix_Array.cpp:497-509) that the unmodified reference produced during the survey
session for inputs fed through its CLI.  The inputs are
re-created here bit for bit (utils/gen_rand_seq.py stream + CLI remap)."""
import hashlib

import pytest

PINS = [  # (seed, N, sha256(input file)[:16], sha256(dump)[:16])
    (1, 1000, "4bc72233db59c623", "fb18c177a9caa2ae"),
    (7, 4096, "bbfc75cad6b19512", "af073a56da834b57"),
    (123, 100000, "4d9e5f2f44301281", "24db5e1b31a804d4"),
]


def test_generator_matches_cpython_random(oracle):
    import random
    for seed, N in [(0, 50), (1, 1000), (12345, 3000), (32767, 777)]:
        random.seed(seed)
        want = "".join(random.choice(["A", "C", "G", "T"]) for _ in range(N)) + "\n"
        assert oracle.gen_rand_seq(seed, N).tobytes().decode() == want


@pytest.mark.parametrize("seed,N,file_sha,dump_sha", PINS)
def test_dump_digest_pins(oracle, seed, N, file_sha, dump_sha):
    raw = oracle.gen_rand_seq(seed, N)
    assert hashlib.sha256(raw.tobytes()).hexdigest()[:16] == file_sha
    T = oracle.remap(raw)
    for p in (0, 64):
        SA, LCP = oracle.build_sa_lcp(T, p=p)
        assert oracle.dump_sha256(SA, LCP)[:16] == dump_sha


def test_simpletest2_dump_pin(oracle):
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    raw = open(os.path.join(here, "golden", "simpletest2.input"), "rb").read()
    SA, LCP = oracle.build_sa_lcp(oracle.remap(raw))
    assert len(oracle.dump_bytes(SA, LCP)) == 952
    assert oracle.dump_sha256(SA, LCP)[:16] == "36c1179e82ddbc8d"
    assert SA[:16].tolist() == [43, 82, 44, 34, 74, 29, 83, 45, 5, 0, 35, 116, 12, 100, 75, 30]
    assert LCP[:16].tolist() == [0, 3, 6, 3, 2, 3, 6, 5, 4, 3, 2, 1, 2, 3, 4, 2]


@pytest.mark.slow
def test_16mi_full_digest_pin(oracle):
    raw = oracle.gen_rand_seq(42, 16 * 1024 * 1024)
    assert hashlib.sha256(raw.tobytes()).hexdigest() == \
        "289f1c7cc75c8cd1f3308b5d7cd5c873ff7e142ddc611bd3a21ca4060d404d72"
    SA, LCP = oracle.build_sa_lcp(oracle.remap(raw), p=8000)
    assert oracle.dump_sha256(SA, LCP) == \
        "8feac4aca81da6d0457628f62282ea6225837697123508e7359b9dacf5abf64e"
