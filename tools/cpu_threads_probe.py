"""How many threads should the CPU baseline use on this host?  The GPU box shows every core of the node (os.cpu_count() = 256) but
runs the command under a CPU quota (cgroup cpu.max: 16 cores' worth); the oracle on a 64 Mi text at several thread counts."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O

def quota():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else int(q) / int(per)
    except (OSError, ValueError):
        return None

n = int(sys.argv[1]) if len(sys.argv) > 1 else (64 << 20) + 1
T = np.frombuffer(np.random.RandomState(3).bytes(n), dtype=np.uint8) & 3
T = np.frombuffer(b"ACGT", dtype=np.uint8)[T]
print(json.dumps({"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_quota_cores": quota()}), flush=True)
for th in [int(x) for x in (sys.argv[2:] or [16, 24, 32, 64, 128, 256])]:
    tm = {}
    t0 = time.time(); O.build_sa_lcp(T, p=8000, threads=th, timings=tm)
    print(json.dumps({"threads": th, "construct_s": round(tm["total"], 2), "wall_s": round(time.time() - t0, 2), "M_suffixes_per_s": round(n / tm["total"] / 1e6, 2)}), flush=True)
