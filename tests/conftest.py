import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes tens of seconds")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O  # oracle/oracle.py (test infrastructure)
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden_cases():
    with open(os.path.join(ROOT, "tests", "golden", "cases.json")) as f:
        return json.load(f)["cases"]


LARGE_GOLDEN = ["dna_cli_140k", "two_letters_skewed_150k", "planted_repeat_145k", "latin1_signed_136k", "dna_cli_200k",
                "markov_skewed_250k"]


def large_golden(name: str):
    """(text, sa, lcp) of one large reference-made fixture (tests/golden/large_<name>.npz: outputs of the reference's
    chatgpt_baseline.py, recorded by tests/golden/make_golden.py).  Long enough for the DEFAULT construction (direct path)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", f"large_{name}.npz"))
    return np.ascontiguousarray(z["text"]), z["sa"], z["lcp"]


def text_bytes(s: str) -> np.ndarray:
    return np.frombuffer(s.encode("latin-1"), dtype=np.uint8)


@pytest.fixture(params=["classic", "auto"])
def sa_path(request, monkeypatch):
    """Which construction a build takes (csrc/pipeline.h Builder::run): the samplesort path forced, or the default
    (direct path whenever the input's shape allows it).  The library reads CAPS_SA_PATH at every build."""
    monkeypatch.setenv("CAPS_SA_PATH", request.param)
    return request.param
