"""Randomised parity stress on the GPU: random sizes, subproblem counts, alphabets and text shapes,
each build compared bit for bit with the oracle (test infrastructure).  usage: stress_gpu.py [seconds] [seed]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: F401  (first: one HIP runtime per process)
import caps_sa_amd
import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
if os.environ.get("STRESS_EMUL"):             # same random sequence through the host emulation (debugging a failure)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emul_util import emul, emul_small
    L = emul_small() if os.environ["STRESS_EMUL"] == "small" else emul()
else:
    L = caps_sa_amd.lib()
DNA = np.frombuffer(b"ACGT", dtype=np.uint8)


def text(kind, n):
    if kind == "uniform":
        return rs.choice(DNA, size=n)
    if kind == "skew":
        w = rs.dirichlet([0.4] * 4)
        return rs.choice(DNA, size=n, p=w)
    if kind == "two":
        return rs.choice(np.frombuffer(b"AT", dtype=np.uint8), size=n, p=[0.8, 0.2])
    if kind == "bytes":
        k = int(rs.randint(5, 200))
        return rs.randint(0, k, size=n).astype(np.uint8) + np.uint8(rs.randint(0, 256 - k))
    if kind == "periodic":
        per = int(rs.randint(1, 200))
        return np.tile(rs.choice(DNA, size=per), n // per + 1)[:n]
    if kind == "planted":
        T = rs.choice(DNA, size=n)
        for _ in range(int(rs.randint(1, 6))):
            ln = int(rs.randint(10, max(11, n // 8)))
            a, b = rs.randint(0, n - ln, size=2)
            T[b:b + ln] = T[a:a + ln]
        return T
    if kind == "runs":
        T = rs.choice(DNA, size=n)
        a = int(rs.randint(0, n // 2)); ln = int(rs.randint(1, n // 3 + 2))
        T[a:a + ln] = T[a]
        return T
    if kind == "stretches":                      # periodic stretches (run table, csrc/text.h): periods around the table's
        sym = DNA if rs.rand() < 0.7 else np.frombuffer(b"abcdefg", dtype=np.uint8)        # limit, lengths around RUN_LONG,
        T = rs.choice(sym, size=n)                                                          # any alignment, text start / end
        for _ in range(int(rs.randint(1, 7))):
            per = int(rs.choice([1, 1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 20, 33]))
            ln = int(rs.choice([rs.randint(40, 200), rs.randint(900, 1200), rs.randint(1200, max(1201, n // 3 + 1202))]))
            ln = min(ln, n)
            where = rs.rand()
            a = 0 if where < 0.15 else (n - ln if where < 0.3 else int(rs.randint(0, n - ln + 1)))
            T[a:a + ln] = np.tile(rs.choice(sym, size=per), ln // per + 1)[:ln]
        if rs.rand() < 0.05:
            T[:] = T[0]
        return T
    raise ValueError(kind)


kinds = ["uniform", "skew", "two", "bytes", "periodic", "planted", "runs", "stretches", "stretches"]
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sa_check import sa_lcp                 # independent prefix-doubling construction: not quadratic on stretches
t0 = time.time(); done = 0; oom = 0; last_note = t0
stats = {"slot_splits": 0, "slot_splits_redone": 0, "long_runs": 0, "path_direct": 0, "direct_quantile": 0, "knot_slot_splits": 0,
         "knot_slot_splits_redone": 0}
fallbacks = {}
while time.time() - t0 < budget:
    kind = kinds[rs.randint(len(kinds))]
    big = rs.rand() < 0.25
    n = int(rs.randint(33, int(os.environ.get("STRESS_MAX_N", 3_000_000)) if big else 60_000))
    if os.environ.get("STRESS_MIN_N"):           # sizes the direct path takes (>= 32 tiles), for the kinds the oracle is not quadratic on
        n = int(rs.randint(int(os.environ["STRESS_MIN_N"]), int(os.environ.get("STRESS_MAX_N", 3_000_000))))
        while kind in ("periodic", "runs"):
            kind = kinds[rs.randint(len(kinds))]
    if kind == "periodic" or kind == "runs":
        n = min(n, 40_000)                       # quadratic in the LCP, for the oracle too
    if kind == "stretches":
        n = max(64, min(n, 400_000))             # checked by sa_check (Python Kasai loop)
    p = int(rs.choice([0, 2, 3, 7, 16, 50, 333, 1000, 8000, max(2, n // 16), max(2, n // 40)]))
    p_mem = min(p, 20000)                        # the p x p matrices (like the reference's) must fit
    if os.environ.get("STRESS_CAP_P"): p = p_mem
    bits = 64 if rs.rand() < 0.2 else 32
    T = text(kind, n)
    # which construction: the default choice, or forced (the library reads these at every build)
    os.environ["CAPS_SA_PATH"] = str(rs.choice(["auto", "auto", "auto", "classic"]))
    mode = str(rs.choice(["auto", "auto", "linear", "quantile"]))
    if mode == "auto": os.environ.pop("CAPS_SA_DIRECT_MODE", None)
    else: os.environ["CAPS_SA_DIRECT_MODE"] = mode
    if rs.rand() < 0.3: os.environ["CAPS_SA_DIRECT_SUB"] = str(rs.choice([1, 2, 8]))
    else: os.environ.pop("CAPS_SA_DIRECT_SUB", None)
    # quantile level B without its count pass: slots smaller than they would be (more buckets outgrow them, more of the stream)
    if rs.rand() < 0.35: os.environ["CAPS_SA_TEST_SPILL_SLOT"] = str(rs.choice([3584, 2816, 1024] if not os.environ.get("STRESS_EMUL") else [224, 160, 64]))
    else: os.environ.pop("CAPS_SA_TEST_SPILL_SLOT", None)
    if os.environ.get("STRESS_ONLY_N") and n != int(os.environ["STRESS_ONLY_N"]):
        continue                                 # replaying one case of a sequence: same random draws, nothing built
    if os.environ.get("STRESS_TRACE"):           # the parameters of every build, before it runs (finding the one that faults)
        print(json.dumps({"next": done, "kind": kind, "n": n, "p": p, "bits": bits, "path": os.environ.get("CAPS_SA_PATH"),
                          "mode": os.environ.get("CAPS_SA_DIRECT_MODE"), "sub": os.environ.get("CAPS_SA_DIRECT_SUB")}), flush=True)
    try:
        if os.environ.get("STRESS_MULTI"):       # the sharded build: 2 .. 8 ranks (all on device 0), with and without the exchange
            world = int(rs.choice([2, 3, 4, 8]))
            if rs.rand() < 0.3: os.environ["CAPS_SA_SHARD_EXCHANGE"] = "1"
            else: os.environ.pop("CAPS_SA_SHARD_EXCHANGE", None)
            if os.environ.get("STRESS_TRACE"):
                print(json.dumps({"world": world, "exchange": os.environ.get("CAPS_SA_SHARD_EXCHANGE")}), flush=True)
            SA, LCP, st = L.build_multi(T, [0] * world, p=p, idx_bits=bits)
        else:
            SA, LCP, st = L.build(T, p=p, idx_bits=bits)
    except caps_sa_amd.CapsSaError as e:         # p^2 matrices beyond the device memory (as in the reference: p^2 on the host)
        if e.code != -4:
            raise
        oom += 1
        continue
    SAo, LCPo = sa_lcp(T, bits) if kind == "stretches" else O.build_sa_lcp(T, p=p, idx_bits=bits)[:2]
    ok = np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo)
    if not ok:
        np.save(os.environ.get("STRESS_DUMP", "/tmp/stress_fail_T.npy"), T)
        print(json.dumps({"FAIL": True, "kind": kind, "n": n, "p": p, "bits": bits, "path": os.environ.get("CAPS_SA_PATH"),
                          "mode": os.environ.get("CAPS_SA_DIRECT_MODE"), "sub": os.environ.get("CAPS_SA_DIRECT_SUB"),
                          "spill_slot": os.environ.get("CAPS_SA_TEST_SPILL_SLOT"), "multi": os.environ.get("STRESS_MULTI"), "exchange": os.environ.get("CAPS_SA_SHARD_EXCHANGE"),
                          "stats": {k: st[k] for k in ("path_direct", "path_fallback", "direct_quantile", "direct_groups")}}))
        sys.exit(1)
    for k in stats: stats[k] += st[k]
    if st["path_fallback"]: fallbacks[st["path_fallback"]] = fallbacks.get(st["path_fallback"], 0) + 1
    done += 1
    if time.time() - last_note > 30:             # a sign of life (a silent GPU job is taken to be hung)
        last_note = time.time()
        print(json.dumps({"progress": done, "seconds": round(time.time() - t0, 1), "path_direct": stats["path_direct"]}), flush=True)
print(json.dumps({"builds": done, "out_of_memory": oom, "seconds": round(time.time() - t0, 1), **stats, "fallbacks_by_code": fallbacks}))
