#!/usr/bin/env python3
"""Generate tests/golden/cases.json from the reference's own correctness tool.

Run ONLY in the dev container (needs /root/reference): it imports the reference's
chatgpt_baseline.py (naive suffix array, chatgpt_baseline.py:5-10, and Kasai LCP,
chatgpt_baseline.py:12-28) -- the "true" program of the reference's correctness recipe
(utils/test-correctness.sh) -- and records its outputs for a fixed list of inputs.
The committed cases.json is data (inputs + expected outputs); nothing at test time
reads /root/reference.

    python tests/golden/make_golden.py
"""
import importlib.util
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("CAPS_SA_REFERENCE", "/root/reference")

spec = importlib.util.spec_from_file_location("chatgpt_baseline", os.path.join(REF, "chatgpt_baseline.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def cli_remap(raw: bytes) -> str:
    # src/main.cpp:61-70
    lookup = "ACTG"
    return "".join(lookup[(ord(chr(c).upper()) & 0x6) >> 1] if c < 128 else lookup[(c & 0x6) >> 1] for c in raw)


def rand_str(seed, n, alphabet):
    rs = random.Random(seed)
    return "".join(rs.choice(alphabet) for _ in range(n))


def gen_rand_seq(seed, n):
    # utils/gen_rand_seq.py:9-16 (print() appends the newline)
    random.seed(seed)
    return "".join(random.choice(["A", "C", "G", "T"]) for _ in range(n)) + "\n"


cases = []


def add(name, text, note=""):
    sa = ref.suffix_array(text)
    lcp = ref.lcp_array(text, sa)
    cases.append({"name": name, "note": note, "text": text, "sa": sa, "lcp": lcp})


with open(os.path.join(REF, "data", "simpletest2"), "rb") as f:
    add("simpletest2_cli", cli_remap(f.read()), "reference data/simpletest2 through the CLI remap")
with open(os.path.join(REF, "data", "banana"), "rb") as f:
    raw = f.read()
add("banana_cli", cli_remap(raw), "data/banana through the CLI remap (n=7: outside the reference's domain)")
add("banana_raw", raw.decode().strip(), "data/banana as chatgpt_baseline.py reads it (stripped)")
with open(os.path.join(REF, "data", "simpletest"), "rb") as f:
    add("simpletest_cli", cli_remap(f.read()), "data/simpletest through the CLI remap (n=5)")
for seed, n in [(1, 1000), (7, 4096), (3, 31), (4, 32), (5, 33), (6, 64), (8, 100), (9, 257)]:
    add(f"gen_rand_seq_{seed}_{n}_cli", cli_remap(gen_rand_seq(seed, n).encode()),
        "utils/gen_rand_seq.py output through the CLI remap")
add("unary_64", "A" * 64, "a^n: SA[i]=n-1-i, LCP[i]=i")
add("unary_1000", "A" * 1000)
add("ac_500", "AC" * 500, "(AC)^k")
add("period37_2000", (rand_str(11, 37, "ACGT") * 60)[:2000], "period-37 string")
add("two_letters_3000", rand_str(12, 3000, "AT"))
add("three_letters_2049", rand_str(13, 2049, "CGT"))
add("lowercase_5000", rand_str(14, 5000, "abcdefghijklmnopqrstuvwxyz"), "26-letter alphabet (8-bit path)")
add("ascii_mixed_3000", rand_str(15, 3000, "ab ,.\n01XYZ~"), "mixed ASCII incl. newline (8-bit path)")
add("long_repeat_4099", (rand_str(16, 700, "ACGT") * 6)[:4099], "long repeats (deep LCPs)")
add("single_char", "G")
add("two_chars", "GA")

with open(os.path.join(HERE, "cases.json"), "w") as f:
    json.dump({"generator": "tests/golden/make_golden.py", "source": "chatgpt_baseline.py (reference)",
               "cases": cases}, f, separators=(",", ":"))
print(f"wrote {len(cases)} cases")
