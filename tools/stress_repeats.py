"""Randomised stress of the deferred-ties path (kernels.h "Deferred ties", pipeline.h msd_refine): repeat-rich random texts -- tandem
arrays of random unit length / copy number / divergence, higher-order arrays, repeat families, exact duplicates, single-letter
blocks, texts that end inside a repeat -- against tests/sa_check.py (prefix doubling + Kasai: independent of the oracle, and not
quadratic on such texts).  Every build bit for bit; the last line counts builds, builds that deferred, and the largest tie_levels.

    python3 tools/stress_repeats.py <builds> <seed> [gpu | emul | emul_small]     (STRESS_MAX_N: largest text, default 2,500,000;
                                                                                   STRESS_MULTI=1: 2 - 8 ranks of the sharded build)

gpu: libcaps_sa_hip.so on cuda:0 (run on the GPU box); emul / emul_small: the host emulation (4096- / 256-element tiles)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from sa_check import sa_lcp  # noqa: E402

DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
BYTES = np.frombuffer(b"acgtn\x80\xfe", dtype=np.uint8)


def mutate(rs, seg, rate, alphabet):
    seg = seg.copy()
    if rate > 0:
        m = rs.rand(seg.size) < rate
        seg[m] = rs.choice(alphabet, size=int(m.sum()))
    return seg


def make_text(rs, n, alphabet):
    T = rs.choice(alphabet, size=n, p=rs.dirichlet(np.ones(alphabet.size) * rs.choice([0.5, 2.0, 20.0])))
    budget = int(n * rs.uniform(0.02, 0.25))            # chars of planted repeat content (more would switch deferral off)
    shapes = []
    while budget > 200:
        kind = rs.choice(["tandem", "hor", "family", "dup", "block"])
        if kind == "tandem":
            unit = rs.choice(alphabet, size=int(rs.choice([1, 2, 3, 5, 17, 23, 57, 171, 300])))
            span = int(min(budget, rs.choice([500, 5_000, 40_000, 200_000])))
            seg = mutate(rs, np.tile(unit, span // unit.size + 1)[:span], rs.choice([0.0, 0.001, 0.01, 0.03]), alphabet)
        elif kind == "hor":
            mono = rs.choice(alphabet, size=int(rs.choice([31, 57, 171])))
            unit = np.concatenate([mutate(rs, mono, 0.2, alphabet) for _ in range(int(rs.randint(2, 9)))])
            span = int(min(budget, rs.choice([5_000, 60_000, 150_000])))
            seg = mutate(rs, np.tile(unit, span // unit.size + 1)[:span], rs.choice([0.0, 0.005, 0.01]), alphabet)
        elif kind == "family":
            cons = rs.choice(alphabet, size=int(rs.choice([40, 120, 300])))
            copies = int(min(budget // cons.size, rs.choice([20, 300, 3000])))
            for pos in rs.randint(0, max(1, n - cons.size), size=copies):
                T[pos:pos + cons.size] = mutate(rs, cons, rs.choice([0.0, 0.02, 0.1]), alphabet)
            budget -= copies * cons.size
            shapes.append(kind)
            continue
        elif kind == "dup":
            span = int(min(budget, rs.choice([300, 3_000, 30_000])))
            src = int(rs.randint(0, max(1, n - span)))
            seg = T[src:src + span].copy()
        else:
            span = int(min(budget, rs.choice([100, 2_000, 30_000])))
            seg = np.full(span, rs.choice(alphabet), dtype=np.uint8)
        at = int(rs.randint(0, max(1, n - seg.size)))
        if rs.rand() < 0.15:
            at = n - seg.size                            # the text ends inside the repeat
        T[at:at + seg.size] = seg
        budget -= seg.size
        shapes.append(kind)
    return T, shapes


def main():
    builds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    backend = sys.argv[3] if len(sys.argv) > 3 else "gpu"
    max_n = int(os.environ.get("STRESS_MAX_N", "2500000"))
    if backend == "gpu":
        import torch  # noqa: F401  (device memory and streams for the library)
        import caps_sa_amd
        L = caps_sa_amd.lib()
    else:
        import emul_util
        L = emul_util.emul_small() if backend == "emul_small" else emul_util.emul()
    rs = np.random.RandomState(seed)
    t0 = time.time()
    tot = {"builds": 0, "deferred": 0, "direct": 0, "max_tie_levels": 0, "tie_groups": 0, "mismatches": 0}
    for b in range(builds):
        lo = 40_000 if backend == "emul_small" else 300_000
        n = int(rs.randint(lo, max(lo + 1, max_n))) + int(rs.randint(0, 3))
        alphabet = DNA if rs.rand() < 0.75 else BYTES
        bits = 32 if rs.rand() < 0.8 else 64
        p = int(rs.choice([0, 0, 16, 200, 3000]))
        mode = rs.choice(["", "", "quantile", "linear"])
        T, shapes = make_text(rs, n, alphabet)
        if mode:
            os.environ["CAPS_SA_DIRECT_MODE"] = mode
        else:
            os.environ.pop("CAPS_SA_DIRECT_MODE", None)
        SAo, LCPo = sa_lcp(T, idx_bits=bits)
        world = int(rs.choice([2, 3, 4, 8])) if os.environ.get("STRESS_MULTI") == "1" else 1
        # STRESS_MULTI=1: the sharded build (caps_sa_hip_build_multi_*), all ranks on device 0 -- the ranks defer and retry on their own
        SA, LCP, st = L.build_multi(T, [0] * world, p=p, idx_bits=bits) if world > 1 else L.build(T, p=p, idx_bits=bits)
        ok = bool(np.array_equal(SA, SAo) and np.array_equal(LCP, LCPo))
        tot["builds"] += 1
        tot["direct"] += int(st["path_direct"])
        tot["deferred"] += int(st["tie_groups_deferred"] > 0)
        tot["tie_groups"] += int(st["tie_groups_deferred"])
        tot["max_tie_levels"] = max(tot["max_tie_levels"], int(st["tie_levels"]))
        if not ok:
            tot["mismatches"] += 1
            bad = int(np.argmax((SA != SAo) | (LCP != LCPo)))
            print(json.dumps({"MISMATCH": b, "seed": seed, "n": n, "bits": bits, "p": p, "mode": mode, "first_bad": bad, "shapes": shapes[:12],
                              "direct": st["path_direct"], "groups": st["tie_groups_deferred"]}), flush=True)
        if (b + 1) % 10 == 0:
            print(json.dumps({"progress": b + 1, "seconds": round(time.time() - t0, 1), **tot}), flush=True)
    tot["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(tot), flush=True)
    sys.exit(1 if tot["mismatches"] else 0)


if __name__ == "__main__":
    main()
