#!/usr/bin/env python3
"""bench.py -- suffixes/sec of the SA+LCP build on a synthetic DNA text (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one complete construction (text already resident in HBM -> SA and LCP resident
in HBM): the interval the reference prints as "Constructed the suffix array"
(src/Suffix_Array.cpp:469,492-493).  Default workload = the configuration the metric is
quoted on, C3: 3,000,000,000 random bases + the trailing newline the CLI remaps to 'C'
(src/main.cpp:61-70), 32-bit indices, 8000 subproblems.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
WORKLOADS = {
    # name: (bases, description)
    "c3": (3_000_000_000, "C3: 3e9 random DNA bases + remapped trailing newline, u32 indices, p=8000"),
    "c2": (268_435_456, "C2: 256 Mi random DNA bases + remapped trailing newline, u32 indices, p=8000"),
}


def make_text(torch, n_bases, seed, device):
    """i.i.d. uniform ACGT on the device, plus the CLI's remapped trailing newline ('C')."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    n = n_bases + 1
    T = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 28
    for o in range(0, n_bases, step):
        m = min(step, n_bases - o)
        T[o:o + m] = lut[torch.randint(0, 4, (m,), device=device, generator=g, dtype=torch.int64)]
    T[n - 1] = ord("C")
    return T


def cpu_baseline(T_dev, n_sample, p):
    """The oracle (C restatement of the reference algorithm: kind 'port') timed on this host's
    cores on a bounded sample of the same text."""
    import oracle as O
    T = T_dev[:n_sample].cpu().numpy()
    tm = {}
    t0 = time.time()
    O.build_sa_lcp(T, p=p, timings=tm)
    wall = time.time() - t0
    return {
        "value": n_sample / tm["total"],
        "unit": "suffixes/s",
        "cores": int(tm["threads"]),
        "kind": "port",
        "sample": f"first {n_sample} chars of the same text, p={p}, oracle construct() interval "
                  f"{tm['total']:.2f} s (wall {wall:.2f} s), host cpus={os.cpu_count()}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--bases", type=int, default=0, help="override the number of bases (debugging)")
    ap.add_argument("--p", type=int, default=8000)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-sample", type=int, default=16 * 1024 * 1024 + 1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true", help="run the device verifier on the last result")
    args = ap.parse_args()

    import torch  # first: the HIP runtime of this process is torch's
    import caps_sa_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
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

    n_bases, desc = WORKLOADS[args.workload]
    if args.bases:
        n_bases, desc = args.bases, f"custom: {args.bases} random DNA bases + remapped newline, u32, p={args.p}"
    n = n_bases + 1
    idx_bits = 32 if n <= 0xFFFFFFFF else 64
    T = make_text(torch, n_bases, args.seed, dev)
    dt = torch.int32 if idx_bits == 32 else torch.int64
    SA = torch.empty(n, dtype=dt, device=dev)
    LCP = torch.empty(n, dtype=dt, device=dev)
    ws_bytes = L.workspace_bytes(n, args.p, idx_bits)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        return L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=args.p, idx_bits=idx_bits,
                              workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=stream)

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

    # Roofline of the DOMINANT kernel family of the timed steps (HIP events on the build's stream
    # around every launch, summed in caps_sa_stats).  Every candidate is one streaming pass over
    # the suffixes; its algorithmic bytes are 4w per suffix (read + write of SA and LCP-sized
    # payload: SURVEY 8d's per-pass figure).  What the kernel really moves (64-bit keys travel
    # with the indices) is reported next to it.
    w = idx_bits // 8
    fams = {
        "tile_sort_kernel": ("tile_sort_ms", "tile_sort_launches", "tile_sort_elems", 2 * (8 + w) + w / 2),
        "merge_pass_kernel": ("merge_pass_ms", "merge_pass_launches", "merge_pass_elems", 2 * (8 + w)),
        "bucket_scatter_kernel": ("bucket_scatter_ms", "bucket_scatter_launches", "bucket_scatter_elems", 1.5 * (8 + w)),
    }
    tot = {k: sum(s[v[0]] for s in stats) for k, v in fams.items()}
    dom = max(tot, key=tot.get)
    f_ms, f_l, f_e, moved_per_elem = fams[dom]
    launches = sum(s[f_l] for s in stats)
    ms = sum(s[f_ms] for s in stats)
    elems = sum(s[f_e] for s in stats)
    roof = None
    if launches and ms > 0:
        avg_ms = ms / launches
        alg_bytes = 4 * w * (elems / launches)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch measured with rocprofv3 PMC passes (cannot be collected inside this
        # process): taken from the committed profile of the same workload, if there is one
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic = json.load(f).get(args.workload if not args.bases else "", {}).get(dom, {}).get("traffic_bytes_per_launch")
        except OSError:
            pass
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_ms": avg_ms, "launches_per_step": launches / args.steps,
                "algorithmic_bytes_per_launch": alg_bytes,
                "moved_bytes_per_launch_incl_keys": moved_per_elem * (elems / launches),
                "kernel_ms_per_step": {k: v / args.steps for k, v in tot.items()}}
    last = stats[-1]
    phases = {k: last[k] for k in ("ms_total", "ms_pack", "ms_sort_subarrays", "ms_select_pivots", "ms_locate_pivots",
                                   "ms_partition", "ms_merge_partitions", "ms_boundary_lcp", "ms_output",
                                   "merge_pass_ms", "tile_sort_ms", "bucket_scatter_ms", "bucket_count_ms", "collate_ms")}
    out = {
        "metric": "suffixes/sec (SA+LCP build)", "value": value, "unit": "suffixes/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": f"u{idx_bits}", "data": "synthetic",
        "config": {"workload": desc, "n": n, "subproblems": last["p_eff"], "bits_per_char": last["bits_per_char"],
                   "merge_passes": [last["merge_passes_phase1"], last["merge_passes_phase2"]],
                   "max_partition": last["max_partition"], "workspace": "preallocated",
                   "workspace_gb": ws_bytes / 1e9, "parallelism": "1 GPU"},
        "phases_ms": phases,
        "roofline": roof,
    }
    if args.verify:
        out["verify_errors"] = L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), idx_bits=idx_bits)
    if not args.no_cpu_baseline:
        del ws
        out["cpu_baseline"] = cpu_baseline(T, min(args.cpu_sample, n), args.p)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
