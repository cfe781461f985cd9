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

# ---------------------------------------------------------------------------------------------------------------
# Large cases (round 3): long enough (>= 32 tiles of 4096 suffixes) for the library's DEFAULT construction -- the
# direct path, csrc/pipeline.h Builder::run_direct -- so that reference-made vectors drive it, not only the chain
# fixtures -> oracle -> HIP.  chatgpt_baseline.suffix_array materialises every suffix (n^2 / 2 bytes: 10 GB at
# n = 140,000), so the sizes stay just above that threshold and the cases are made one at a time.  Stored as
# tests/golden/large_<name>.npz (text, sa, lcp as arrays; data only).
#
# latin1_signed: the reference compares `char`s, which are SIGNED on its platform (src/Suffix_Array.cpp:75-77,289), while
# Python orders str by code point.  The text handed to chatgpt_baseline is therefore the order-isomorphic string
# chr(byte ^ 0x80) (signed order of the bytes = code point order of that string); SA and LCP are position-based, so they
# are the expected arrays of the byte text itself.
# ---------------------------------------------------------------------------------------------------------------
import gc
import numpy as np


def add_large(name, raw: bytes, note, signed_order=False):
    text = "".join(chr(b ^ 0x80) for b in raw) if signed_order else raw.decode("latin-1")
    sa = ref.suffix_array(text)
    gc.collect()
    lcp = ref.lcp_array(text, sa)
    np.savez_compressed(os.path.join(HERE, f"large_{name}.npz"), text=np.frombuffer(raw, dtype=np.uint8),
                        sa=np.asarray(sa, dtype=np.uint32), lcp=np.asarray(lcp, dtype=np.uint32),
                        note=np.asarray(note), signed_order=np.asarray(signed_order))
    print(f"wrote large_{name}.npz  n={len(raw)}  max lcp {max(lcp)}")
    del sa, lcp, text
    gc.collect()


if os.environ.get("CAPS_GOLDEN_LARGE", "1") != "0":
    add_large("dna_cli_140k", cli_remap(gen_rand_seq(21, 140_000).encode()).encode(),
              "utils/gen_rand_seq.py 21 140000 through the CLI remap (2-bit path, uniform keys)")
    add_large("two_letters_skewed_150k", "".join(random.Random(22).choices("AT", weights=[0.85, 0.15], k=150_000)).encode(),
              "two letters, 85 % / 15 % (skewed keys: quantile buckets)")
    rs = random.Random(23)
    body = [rs.choice("ACGT") for _ in range(145_000)]
    rep = body[10_000:15_000]
    body[70_000:75_000] = rep                                            # an exact 5-kb duplicate
    mut = list(rep)
    for i in rs.sample(range(5000), 50):
        mut[i] = rs.choice("ACGT")
    body[120_000:125_000] = mut                                          # and a copy with 1 % point mutations
    body[40_000:40_600] = "G" * 600                                      # an N-block stand-in (src/main.cpp:61-68 maps N to G)
    add_large("planted_repeat_145k", "".join(body).encode(), "random DNA + exact and mutated 5-kb repeat + a 600-long G run (deep LCPs)")
    rs = random.Random(24)
    add_large("latin1_signed_136k", bytes(rs.choice([0x00, 0x41, 0x61, 0x7F, 0x80, 0x81, 0xC3, 0xE9, 0xFF]) for _ in range(136_000)),
              "bytes on both sides of 0x80: signed-char order (8-bit path); see the comment in make_golden.py", signed_order=True)
    rs = random.Random(26)
    trans = {a + b: [rs.random() ** 3 + 0.01 for _ in range(4)] for a in "ACGT" for b in "ACGT"}     # order-2 chain, skewed rows
    out = ["A", "C"]
    for _ in range(250_000 - 2):
        out.append(rs.choices("ACGT", weights=trans[out[-2] + out[-1]])[0])
    add_large("markov_skewed_250k", "".join(out).encode(), "order-2 Markov chain with skewed transitions (genome-like k-mer skew; 61 tiles)")
    add_large("dna_cli_200k", cli_remap(gen_rand_seq(25, 200_000).encode()).encode(),
              "utils/gen_rand_seq.py 25 200000 through the CLI remap (48 tiles)")
