#!/usr/bin/env python3
"""bench.py -- suffixes/sec of the SA+LCP build on a synthetic DNA text (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one complete construction (text already resident in HBM -> SA and LCP resident
in HBM): the interval the reference prints as "Constructed the suffix array"
(src/Suffix_Array.cpp:469,492-493).  Default workload = the configuration the metric is
quoted on, C3: 3,000,000,000 random bases + the trailing newline the CLI remaps to 'C'
(src/main.cpp:61-70), 32-bit indices, 8000 subproblems.  Prints ONE JSON line (rank 0).

The result of the LAST timed step is always checked by the exact device verifier
(permutation + adjacent order + exact LCP on the raw bytes): `verify_errors` in the line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
WORKLOADS = {
    # name: (bases, kind, description)
    # "uniform" = the reference's generator itself: utils/gen_rand_seq.py <seed> <N> (CPython's MT19937 stream, reproduced bit for bit by
    # caps_sa_hip_gen_rand_seq) + the newline its print() appends, which the CLI remaps to 'C' (src/main.cpp:61-70) -- SURVEY 8d's inputs
    "c3": (3_000_000_000, "uniform", "C3: gen_rand_seq.py 42 3000000000 (MT19937 stream) + remapped trailing newline, u32 indices, p=8000"),
    "c2": (268_435_456, "uniform", "C2: gen_rand_seq.py 42 268435456 (MT19937 stream) + remapped trailing newline, u32 indices, p=8000"),
    # the same shapes from torch.randint (Philox): what rounds 1 and 2 benched; statistically the same text
    "c3t": (3_000_000_000, "uniform-torch", "3e9 random DNA bases (torch.randint) + remapped trailing newline, u32 indices, p=8000"),
    "c2t": (268_435_456, "uniform-torch", "256 Mi random DNA bases (torch.randint) + remapped trailing newline, u32 indices, p=8000"),
    # GRCh38 is not available offline: order-5 Markov chain with Dirichlet(0.5) transitions + 2 % planted mutated
    # repeats (tools/genome_like.py, seeded), and the same with N-block stand-ins (the CLI maps N to G: src/main.cpp:61-68)
    "g3": (3_000_000_000, "genome", "genome-like (tools/genome_like.py seed 7): 3e9 bases, u32 indices, p=8000"),
    "g2": (268_435_456, "genome", "genome-like (tools/genome_like.py seed 7): 256 Mi bases, u32 indices, p=8000"),
    "g3n": (3_000_000_000, "genome+n", "genome-like + single-letter blocks (1 x 2e6, 5 x 5e5, 100 x 5e4): 3e9 bases, u32, p=8000"),
    # ... and with GRCh38-shaped repeats: satellite tandem arrays (period 171 / 2052), a 300-base family in 1e5 copies,
    # segmental duplications incl. an exact 50-kb one (tools/genome_like.py plant_genome_repeats)
    "g3r": (3_000_000_000, "genome+r", "genome-like + satellite arrays, 1e5-copy 300-base family, segmental duplications: 3e9 bases, u32, p=8000"),
    "g2r": (268_435_456, "genome+r", "genome-like + satellite arrays, repeat family, segmental duplications: 256 Mi bases, u32, p=8000"),
    # BASELINE config 4's shape: needs --gpus 8 (one GPU cannot hold a whole build of it; tools/shard_probe.py c4 times one rank)
    "c4": (8 << 30, "uniform", "C4: gen_rand_seq.py 42 8589934592 + remapped trailing newline, u64 indices, p=8000 (multi-GPU only)"),
}


def make_text(torch, n_bases, seed, device, kind="uniform"):
    """The workload's text on the device, plus the CLI's remapped trailing newline ('C')."""
    n = n_bases + 1
    if kind.startswith("genome"):
        from genome_like import markov_dna
        T = torch.empty(n, dtype=torch.uint8, device=device)
        T[:n_bases] = markov_dna(n_bases, seed=7, device=device)
        if kind == "genome+n":
            g = torch.Generator(device=device)
            g.manual_seed(99)
            for ln, cnt in ((2_000_000, 1), (500_000, 5), (50_000, 100)):
                if ln * 4 < n_bases:
                    for a in torch.randint(0, n_bases - ln, (cnt,), device=device, generator=g).tolist():
                        T[a:a + ln] = ord("G")
        if kind == "genome+r":
            from genome_like import plant_genome_repeats
            plant_genome_repeats(T, n_bases)
        T[n - 1] = ord("C")
        return T
    if kind == "uniform":
        # host generator into page-locked memory (already touched: fresh pageable pages cost 3 x the generator itself), one copy up
        import caps_sa_amd
        L = caps_sa_amd.lib()
        raw = L.pinned_empty(n, "uint8") if device.type == "cuda" else __import__("numpy").empty(n, dtype="uint8")
        L.gen_rand_seq(seed, n_bases, raw)
        raw[n - 1] = ord("C")
        T = torch.empty(n, dtype=torch.uint8, device=device)
        T.copy_(torch.from_numpy(raw))
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        del raw
        return T
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    T = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 28
    for o in range(0, n_bases, step):
        m = min(step, n_bases - o)
        T[o:o + m] = lut[torch.randint(0, 4, (m,), device=device, generator=g, dtype=torch.int64)]
    T[n - 1] = ord("C")
    return T


def cpu_quota_cores():
    """CPU cores this process may use: the cgroup's quota when there is one (the GPU box shows all 256 hardware threads of its node but
    runs the command under cpu.max = 16 cores' worth), else the affinity mask."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()
            if q != "max":
                return max(1.0, min(float(aff), int(q) / int(per)))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1.0, min(float(aff), q / per))
    except (OSError, ValueError):
        pass
    return float(aff)


def cpu_threads():
    """Threads for the oracle: two per core of the quota (measured on the GPU box, 64 Mi text, quota 16: 16 / 24 / 32 / 64 / 128 /
    256 threads -> 28.8 / 29.9 / 31.3 / 27.5 / 23.1 / 16.5 M suffixes/s; tools/cpu_threads_probe.py)."""
    env = os.environ.get("CAPS_BENCH_CPU_THREADS")
    if env:
        return max(1, int(env))
    return max(1, int(round(2 * cpu_quota_cores())))


def cpu_baseline(T_dev, n_sample, p, n_full):
    """The oracle (C restatement of the reference algorithm: kind 'port') timed on this host's
    cores on a bounded sample of the same text."""
    import oracle as O
    T = T_dev[:n_sample].cpu().numpy()
    tm = {}
    t0 = time.time()
    O.build_sa_lcp(T, p=p, threads=cpu_threads(), timings=tm)
    wall = time.time() - t0
    rate = n_sample / tm["total"]
    cores = cpu_quota_cores()
    return {
        "value": rate,
        "unit": "suffixes/s",
        "cores": int(round(cores)),
        "threads": int(tm["threads"]),
        "kind": "port",
        "seconds": tm["total"],
        "sample": f"first {n_sample} chars of the same text, p={p}: oracle construct() interval {tm['total']:.2f} s "
                  f"(wall {wall:.2f} s) on {int(tm['threads'])} threads over the {cores:g} cores this process may use (cgroup quota; the "
                  f"node shows {os.cpu_count()} hardware threads); at that rate the full {n_full}-char text would take "
                  f"{n_full / rate:.0f} s (an underestimate: merge depth grows with n)",
    }


def host_memory_available_gb():
    """What this process may still allocate: MemAvailable, capped by the cgroup's limit minus its usage (the GPU box runs the
    command in a container)."""
    avail = float("inf")
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024 / 1e9
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            m = open(lim).read().strip()
            if m != "max":
                avail = min(avail, (int(m) - int(open(cur).read().strip())) / 1e9)
        except (OSError, ValueError):
            pass
    return avail


def cpu_baseline_full(torch, T_dev, n, p, idx_bits, SA_dev, LCP_dev):
    """--cpu-full: the oracle's construct() on the WHOLE text on all host threads (the interval the reference prints as
    "Constructed the suffix array", src/Suffix_Array.cpp:469,492-493), in this process, and its SA / LCP compared entry by
    entry with the arrays the GPU build left in HBM -- a size-matched parity check on top of the device verifier.  Needs
    ~(1 + 4 w) n bytes of host memory (C3: 51 GB) and a few minutes; refused when the host cannot hold it."""
    import numpy as np
    import oracle as O
    w = idx_bits // 8
    need_gb = (1 + 4 * w) * n / 1e9 + 4.0
    have_gb = host_memory_available_gb()
    if have_gb < need_gb * 1.15:
        return {"skipped": f"host memory: {have_gb:.0f} GB available, {need_gb:.0f} GB needed for the oracle at n = {n}"}
    T = T_dev.cpu().numpy()
    tm = {}
    t0 = time.time()
    SAo, LCPo = O.build_sa_lcp(T, p=p, idx_bits=idx_bits, threads=cpu_threads(), timings=tm)
    wall = time.time() - t0
    cores = cpu_quota_cores()
    dt = torch.int32 if idx_bits == 32 else torch.int64
    sa_diff = lcp_diff = 0
    step = 1 << 28
    for o in range(0, n, step):                     # chunk by chunk through HBM: no second copy of the arrays on the host
        m = min(step, n - o)
        a = torch.from_numpy(SAo[o:o + m].view(np.int32 if idx_bits == 32 else np.int64)).to(SA_dev.device)
        sa_diff += int((a != SA_dev[o:o + m].view(dt)).sum().item())
        a = torch.from_numpy(LCPo[o:o + m].view(np.int32 if idx_bits == 32 else np.int64)).to(SA_dev.device)
        lcp_diff += int((a != LCP_dev[o:o + m].view(dt)).sum().item())
        del a
    del SAo, LCPo
    return {
        "value": n / tm["total"], "unit": "suffixes/s", "cores": int(round(cores)), "threads": int(tm["threads"]), "kind": "port",
        "seconds": tm["total"],
        "sample": f"full text ({n} chars), p={p}: oracle construct() interval {tm['total']:.1f} s (wall {wall:.1f} s) on "
                  f"{int(tm['threads'])} threads over the {cores:g} cores this process may use (cgroup quota; the node shows "
                  f"{os.cpu_count()} hardware threads); its SA and LCP compared entry by entry with the GPU build's",
        "phases_s": {k: v for k, v in tm.items() if k != "threads"},
        "sa_mismatches_vs_gpu": sa_diff, "lcp_mismatches_vs_gpu": lcp_diff,
    }


def kernel_sources_sha256():
    """Hash of the sources the kernels are built from: profiles/traffic.json carries the hash of the build its PMC passes
    profiled, and its figures are used only while the two agree (the GPU box has no git history to compare commits with)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "caps-sa_amd", "csrc")
    for f in ("kernel_lang.h", "text.h", "kernels.h", "pipeline.h"):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def roofline(kernels, w, traffic_key, build_ms_total):
    """kernels: {family: (ms summed over its launches, launches, elements summed, algorithmic bytes per element or None)} over the
    timed steps.  The streaming passes over the suffixes (level A, level B count and scatter, tile sort, merge passes) are priced
    at the algorithmic 4w bytes per suffix (read + write of an SA- and an LCP-sized payload: SURVEY 8d's per-pass figure); the
    small families at what they must move (pack: text in, packed text + run table out; deferred ties: key + index in and out per
    level).  The dominant (largest summed time) family is the headline; all of them are listed, and `coverage_of_build` says what
    share of the builds' device time they explain."""
    moved = {                               # what a pass really moves per element (64-bit keys travel with the indices)
        "pack": 1.0 + 0.25 + 0.25,                               # text in, 2-bit text + run table out
        "level_a_scatter": 0.25 + (8 + w),                       # packed text in, (key, sa) out
        "level_b_count": 8 + 2,                                  # keys in, 16-bit bucket ids out
        "level_b_scatter": 2 * (8 + w),                          # (key, sa) in and out
        "bucket_scatter_kernel": 2 * (8 + w),
        "tile_sort_kernel": (8 + w) + 2 * w,                     # (key, sa) in, SA + LCP out
        "merge_pass_kernel": 2 * (8 + w),
        "deferred_ties": 2 * (8 + w),
    }
    traffic, traffic_note = {}, "no profiles/traffic.json"
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        if tj.get("kernel_sources_sha256") == kernel_sources_sha256():
            traffic = tj.get(traffic_key, {})
            traffic_note = "profiles/traffic.json (rocprofv3 PMC passes of this build and workload, FETCH_SIZE x2 + WRITE_SIZE; tools/pmc_traffic.py)"
        else:
            traffic_note = ("profiles/traffic.json was measured on other kernel sources (its kernel_sources_sha256 differs): not used; "
                            "re-run tools/pmc.sh + tools/pmc_traffic.py")
    except OSError:
        pass
    rows = {}
    for name, (ms, launches, elems, alg_per_elem) in kernels.items():
        if not launches or ms <= 0:
            continue
        avg_ms = ms / launches
        alg = (alg_per_elem if alg_per_elem is not None else 4 * w) * (elems / launches)
        ach = alg / (avg_ms * 1e-3) / 1e9
        small = ms < 0.01 * build_ms_total                     # under 1 % of the builds: listed for the coverage, not rated
        rows[name] = {"avg_launch_ms": avg_ms, "launches": launches, "achieved": None if small else ach,
                      "frac": None if small else ach / HBM_PEAK_GBS,
                      "algorithmic_bytes_per_launch": alg,
                      "moved_bytes_per_launch_incl_keys": moved.get(name, 0) * (elems / launches),
                      "traffic": (traffic.get(name) or {}).get("traffic_bytes_per_launch")}
    if not rows:
        return None
    dom = max(rows, key=lambda k: rows[k]["avg_launch_ms"] * rows[k]["launches"])
    d = rows[dom]
    explained = sum(r["avg_launch_ms"] * r["launches"] for r in rows.values())
    return {"bound": "hbm", "kernel": dom, "achieved": d["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
            "traffic": d["traffic"], "avg_launch_ms": d["avg_launch_ms"],
            "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
            "moved_bytes_per_launch_incl_keys": d["moved_bytes_per_launch_incl_keys"],
            "traffic_source": traffic_note,
            "coverage_of_build": explained / build_ms_total if build_ms_total > 0 else None,
            "coverage_note": "sum of the families' device times (HIP events on the build's stream) / sum of the builds' device times (ms_total)",
            "kernels": rows}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--prewarm-s", type=float, default=1.0,
                    help="untimed builds before the W warm-up steps until this many seconds have passed (first use of the 99 GB workspace, "
                         "clocks); 0 = off.  It does NOT remove the run-to-run spread of the level-A scatter (12.0 or 13.7-13.9 ms per C3 "
                         "build, the same for all builds of a process, whatever ran before: 47.8 or 50 ms per build)")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--bases", type=int, default=0, help="override the number of bases (debugging)")
    ap.add_argument("--p", type=int, default=8000)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-sample", type=int, default=256 * 1024 * 1024 + 1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true",
                    help="time the oracle on the WHOLE text (minutes, ~(1 + 4w) n bytes of host memory) and compare its SA / LCP with the "
                         "GPU's entry by entry, whatever the projection says (the default does this when host memory and time allow)")
    ap.add_argument("--cpu-sample-only", action="store_true",
                    help="CPU baseline on the --cpu-sample prefix only (also: CAPS_BENCH_CPU_FULL=0)")
    ap.add_argument("--cpu-full-budget-s", type=float, default=float(os.environ.get("CAPS_BENCH_CPU_FULL_BUDGET_S", "330")),
                    help="the default run times the oracle on the whole text only when the sample's rate projects it under this many seconds")
    ap.add_argument("--no-verify", action="store_true", help="skip the device verifier after the timed steps")
    ap.add_argument("--verify", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--no-host-path", action="store_true", help="skip timing the host-buffer entry point (PCIe inclusive)")
    ap.add_argument("--shard-mode", default="auto", choices=["local", "exchange", "auto"],
                    help="N > 1: how the direct path is sharded (caps_sa_dist.bench_main); auto times both in the warm-up and keeps the faster")
    ap.add_argument("--text-file", default="", help="read the workload's text from this .npy file (tools/make_text.py wrote it) instead of "
                    "generating it: the genome-like generators launch ~1e6 tiny kernels, which rocprofv3 --pmc does not survive")
    return ap.parse_args(argv)


def main():
    args = parse_args()

    import torch  # first: the HIP runtime of this process is torch's
    import caps_sa_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if args.workload == "c4" and world < 8 and not args.bases:
        raise SystemExit("--workload c4 (8 Gi + 1 bases, 64-bit indices) needs --gpus 8: one GPU cannot hold a whole build of it "
                         "(tools/shard_probe.py c4 times one rank of eight on one GPU)")
    if world > 1 or os.environ.get("CAPS_SA_FORCE_SHARDED") == "1":
        # sharded path (one process per GPU, RCCL exchange); CAPS_SA_FORCE_SHARDED=1 runs it at world
        # size 1 too (smoke test of the multi-GPU driver on a 1-GPU box)
        import caps_sa_dist
        return caps_sa_dist.bench_main(args, rank, local_rank, world)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    L = caps_sa_amd.lib()

    n_bases, kind, desc = WORKLOADS[args.workload]
    if args.bases:
        n_bases, desc = args.bases, f"custom: {args.bases} bases of kind {kind} + remapped newline, p={args.p}"
    n = n_bases + 1
    idx_bits = 32 if n <= 0xFFFFFFFF else 64
    if args.text_file:
        import numpy as np
        T = torch.from_numpy(np.load(args.text_file, mmap_mode="r")[:n].copy()).to(dev)
        assert T.numel() == n, "the text file belongs to another workload"
    else:
        T = make_text(torch, n_bases, args.seed, dev, kind)
    dt = torch.int32 if idx_bits == 32 else torch.int64
    SA = torch.empty(n, dtype=dt, device=dev)
    LCP = torch.empty(n, dtype=dt, device=dev)
    ws_bytes = L.workspace_bytes(n, args.p, idx_bits, bits_per_char=2)       # every workload here is DNA: the 2-bit text arena
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        return L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=args.p, idx_bits=idx_bits,
                              workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=stream)

    prewarm_steps = 0
    tp = time.perf_counter()
    while args.prewarm_s > 0 and time.perf_counter() - tp < args.prewarm_s and prewarm_steps < 64:
        step()
        torch.cuda.synchronize()
        prewarm_steps += 1
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms_per_step = 1e3 * elapsed / args.steps
    value = n / (elapsed / args.steps)
    verify_errors = None if args.no_verify else L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), idx_bits=idx_bits)

    # kernel families of the timed steps (HIP events on the build's stream around every launch, summed in caps_sa_stats)
    w = idx_bits // 8
    direct = all(s["path_direct"] for s in stats)
    fams = {}
    S = len(stats)
    tot = lambda k: sum(s[k] for s in stats)                                               # noqa: E731
    bits = stats[-1]["bits_per_char"]
    fams["pack"] = (tot("ms_pack"), S, n * S, 1.0 + bits / 8.0 + bits / 8.0)             # text in; packed text + run table out
    # samples -> sorted -> pivots / knots / group keys: its elements are the samples, a few per mille of the suffixes
    m_samples = stats[-1]["p_eff"] * stats[-1]["ppp"]
    fams["sample_pivots"] = (tot("ms_select_pivots"), S, m_samples * S, 2.0 * (8 + w))
    if direct:      # the first scatter launch of a direct build is level A (text -> groups), the second level B
        fams["level_a_scatter"] = (tot("level_a_ms"), S, n * S, None)
        lb = sum(s["bucket_scatter_ms"] - s["level_a_ms"] for s in stats)
        fams["level_b_scatter"] = (lb, sum(s["bucket_scatter_launches"] - 1 for s in stats), n * S, None)
    else:
        fams["bucket_scatter_kernel"] = (tot("bucket_scatter_ms"), tot("bucket_scatter_launches"), tot("bucket_scatter_elems"), None)
    if tot("bucket_count_ms") > 0:
        fams["level_b_count"] = (tot("bucket_count_ms"), S, n * S, None)
    fams["tile_sort_kernel"] = (tot("tile_sort_ms"), tot("tile_sort_launches"), tot("tile_sort_elems"), None)
    fams["merge_pass_kernel"] = (tot("merge_pass_ms"), tot("merge_pass_launches"), tot("merge_pass_elems"), None)
    if tot("collate_ms") > 0:
        fams["collate"] = (tot("collate_ms"), S, n * S, None)
    if tot("ms_sort_subarrays") > 0 or tot("ms_locate_pivots") > 0:                        # samplesort path: the located pivots (a8)
        fams["locate_pivots"] = (tot("ms_locate_pivots"), S, n * S, None)
    fams["finish_gather_head_lcps"] = (tot("finish_ms"), S, n * S, 2.0 * w)
    if tot("run_bucket_ms") > 0:
        fams["run_buckets"] = (tot("run_bucket_ms"), S, max(1, n * S), None)
    if tot("msd_ms") > 0:
        te = max(1, tot("tie_elems_deferred"))
        fams["deferred_ties"] = (tot("msd_ms"), S, te, 2.0 * (8 + w) * max(1, stats[-1]["tie_levels"]))
    roof = roofline(fams, w, args.workload if not args.bases else "", tot("ms_total"))
    last = stats[-1]
    phases = {k: last[k] for k in ("ms_total", "ms_pack", "ms_sort_subarrays", "ms_select_pivots", "ms_locate_pivots",
                                   "ms_partition", "ms_merge_partitions", "ms_boundary_lcp", "ms_output",
                                   "merge_pass_ms", "tile_sort_ms", "bucket_scatter_ms", "bucket_count_ms", "collate_ms",
                                   "level_a_ms", "finish_ms", "run_bucket_ms", "msd_ms")}
    out = {
        "metric": "suffixes/sec (SA+LCP build)", "value": value, "unit": "suffixes/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "prewarm_steps": prewarm_steps, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": f"u{idx_bits}", "data": "synthetic",
        "config": {"workload": desc, "n": n, "subproblems": last["p_eff"], "bits_per_char": last["bits_per_char"],
                   "construction": "direct (pivots from the text, one scatter into groups, per-group sort)" if direct
                                   else f"samplesort (fallback reason {last['path_fallback']})",
                   "groups": last["direct_groups"], "slot_splits": [last["slot_splits"], last["slot_splits_redone"]],
                   "knot_slot_splits": {"kept": last["knot_slot_splits"], "redone_with_count_pass": last["knot_slot_splits_redone"],
                                        "elements_on_the_spill_stream": last["spill_entries"]},
                   "merge_passes": [last["merge_passes_phase1"], last["merge_passes_phase2"]],
                   "max_partition": last["max_partition"], "workspace": "preallocated",
                   "workspace_gb": ws_bytes / 1e9, "parallelism": "1 GPU"},
        "phases_ms": phases,
        "whole_build_floor": {"bytes": (1 + 2 * w) * n, "frac_of_peak": (1 + 2 * w) * n / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": "compulsory traffic (read T once, write SA and LCP once) / ms_per_step / 8 TB/s"},
        "bytes_per_suffix": {"compulsory_floor": 1 + 2 * w,
                             "moved_by_the_big_kernels": (sum(r["moved_bytes_per_launch_incl_keys"] * r["launches"]
                                                              for r in roof["kernels"].values()) / (n * len(stats))) if roof else None},
        "roofline": roof,
        "verify_errors": verify_errors,
    }
    parity_errors = 0
    if not args.no_cpu_baseline:
        # The CPU path on the WHOLE text, its arrays compared with the GPU's entry by entry (north_star: "bit-exact ... with that CPU
        # path timed on the GPU box's own host cores in the same run") -- by default, when the host can hold it and the sample's rate
        # projects it inside the budget; the sample alone otherwise (and with --cpu-sample-only / CAPS_BENCH_CPU_FULL=0).
        sample = cpu_baseline(T, min(args.cpu_sample, n), args.p, n)
        want_full = args.cpu_full or not (args.cpu_sample_only or os.environ.get("CAPS_BENCH_CPU_FULL") == "0")
        projected = 1.3 * n / sample["value"]              # merge depth grows with n: C3 measured 1.25 x the sample's projection
        if want_full and n > args.cpu_sample and (args.cpu_full or projected <= args.cpu_full_budget_s):
            del ws
            torch.cuda.empty_cache()
            full = cpu_baseline_full(torch, T, n, args.p, idx_bits, SA, LCP)
            ws = None
            if "skipped" in full:
                out["cpu_baseline_full_skipped"] = full["skipped"]
                out["cpu_baseline"] = sample
            else:
                parity_errors = full.get("sa_mismatches_vs_gpu", 0) + full.get("lcp_mismatches_vs_gpu", 0)
                out["cpu_baseline"] = full
                out["cpu_baseline_sample"] = {k: sample[k] for k in ("value", "seconds", "sample")}
        else:
            if want_full and n > args.cpu_sample:
                out["cpu_baseline_full_skipped"] = (f"projected {projected:.0f} s for the whole text exceeds the budget of "
                                                    f"{args.cpu_full_budget_s:.0f} s (--cpu-full forces it)")
            out["cpu_baseline"] = sample
    if not args.no_host_path:
        del ws, SA, LCP
        torch.cuda.empty_cache()
        out["pcie_inclusive"] = host_path(L, torch, T, n, args.p, idx_bits)
    print(json.dumps(out))
    host_errors = (out.get("pcie_inclusive") or {}).get("verify_errors") or 0
    if verify_errors or parity_errors or host_errors:           # a wrong result is not a benchmark result
        raise SystemExit(f"verify_errors = {verify_errors}, mismatches against the oracle = {parity_errors}, "
                         f"verify_errors of the host path = {host_errors}")


def host_path(L, torch, T_dev, n, p, idx_bits):
    """What construct() callers see: caps_sa_hip_build_* on host buffers (H2D of T, build, D2H of SA and LCP) -- the
    PCIe-inclusive interval.  Never `value`.  Result arrays page-locked (what the class mirror allocates in its
    constructor) and pageable; the entry point keeps its device block between calls, so the first call also pays for it."""
    import numpy as np
    T = T_dev.cpu().numpy()
    sfx_dt = np.uint32 if idx_bits == 32 else np.uint64
    res = {"n": n}
    for label, pinned in (("pinned_results", True), ("pageable_results", False)):
        SA = L.pinned_empty(n, sfx_dt) if pinned else np.empty(n, dtype=sfx_dt)
        LCP = L.pinned_empty(n, sfx_dt) if pinned else np.empty(n, dtype=sfx_dt)
        times, st = [], None
        for _ in range(4):        # (the LCP bytes' staging is allocated by the second host build of a process: the third is steady)
            t0 = time.perf_counter()
            st = L.build_into(T, SA, LCP, p=p, idx_bits=idx_bits)
            times.append(1e3 * (time.perf_counter() - t0))
        res[label] = {"first_call_ms": times[0], "calls_ms": times, "steady_ms": min(times[1:]), "ms_h2d": st["ms_h2d"], "ms_d2h": st["ms_d2h"],
                      "ms_build": st["ms_total"], "result_waves": st.get("result_waves", 1), "lcp_bytes_on_link": st.get("lcp_bytes_on_link"),
                      "suffixes_per_s": n / (min(times[1:]) * 1e-3)}
        if pinned:
            # what came back over the link (in waves, on a second stream, while later groups were sorted) checked ONCE by the exact
            # device verifier: back up in chunks, no second host copy
            L.release_cache()
            dt = torch.int32 if idx_bits == 32 else torch.int64
            npdt = np.int32 if idx_bits == 32 else np.int64
            dSA = torch.empty(n, dtype=dt, device=T_dev.device)
            dLCP = torch.empty(n, dtype=dt, device=T_dev.device)
            for o in range(0, n, 1 << 28):
                m = min(1 << 28, n - o)
                dSA[o:o + m] = torch.from_numpy(SA[o:o + m].view(npdt)).to(T_dev.device)
                dLCP[o:o + m] = torch.from_numpy(LCP[o:o + m].view(npdt)).to(T_dev.device)
            res["verify_errors"] = L.verify_device(T_dev.data_ptr(), n, dSA.data_ptr(), dLCP.data_ptr(), idx_bits=idx_bits)
            del dSA, dLCP
            torch.cuda.empty_cache()
        del SA, LCP
    L.release_cache()
    return res


if __name__ == "__main__":
    main()
