"""caps_sa_dist.py -- multi-GPU SA+LCP construction: one process per GPU, torch.distributed.

Host driver of the sharded path.  The kernels run inside
libcaps_sa_hip.so through the caps_sa_hip_shard_* entry points; this module only issues the
collectives between them, with backend "nccl" (= RCCL over xGMI) on GPUs:

    phase1 (local)  -> all_gather(samples)          small: p*ppp*(8+w) bytes in total
    pivots (local)  -> all_gather(partition sizes)  p u64 per rank
    collate (local) -> all_to_all_v(key, sa)        THE exchange: (8+w)*n*(G-1)/G bytes, direct p2p
    phase2 (local)  -> all_gather(last SA of slice) 1 idx per rank
    fix first LCP of the slice (local)

The text is replicated; rank r ends up with the contiguous slice
[slice_off, slice_off + slice_len) of the global SA and LCP arrays in its own HBM.
`lib` is a CapsLib binding: the product library on GPUs; the CPU tests (gloo, world size 2)
pass the host emulation of the same sources instead.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist


def _idx_dtype(idx_bits: int):
    return torch.int32 if idx_bits == 32 else torch.int64


def _all_gather_var(t: torch.Tensor, counts: list[int], max_elems: int = 1 << 25) -> torch.Tensor:
    """all_gather of per-rank tensors of different lengths, in rounds of at most max_elems
    elements per rank (collective payloads are kept far below RCCL's safe message size)."""
    world = dist.get_world_size()
    total = sum(counts)
    out = torch.empty(total, dtype=t.dtype, device=t.device)
    offs = [sum(counts[:r]) for r in range(world)]
    mx = max(counts) if counts else 0
    for lo in range(0, max(mx, 1), max_elems):
        step = min(max_elems, mx - lo) if mx else 0
        if step <= 0:
            break
        pad = torch.zeros(step, dtype=t.dtype, device=t.device)
        mine = t[lo:lo + step]
        pad[:mine.numel()] = mine
        buf = torch.empty(step * world, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf, pad)
        for r in range(world):
            c = max(0, min(step, counts[r] - lo))
            if c:
                out[offs[r] + lo: offs[r] + lo + c] = buf[r * step: r * step + c]
    return out


# Largest message (bytes) handed to one collective / point-to-point call per peer.  Measured on
# this stack (RCCL 2.26.6 / torch 2.10, MI355X): a message above 2**30 bytes is silently
# truncated to about half (tools/a2a_probe.py), so larger sub-subarray blocks go in rounds.
A2A_MAX_BYTES = (1 << 30) - (1 << 16)


def exchange_plan(rc: list[int], sc: list[int], elem_bytes: int, device, max_bytes: int | None = None):
    """(rounds, elements per round and peer) for an all-to-all-v of blocks of rc/sc elements whose
    widest element has elem_bytes bytes; agreed on by all ranks (one tiny all_reduce)."""
    max_bytes = max_bytes or A2A_MAX_BYTES
    cmax = max(1, max_bytes // elem_bytes)
    big = torch.tensor([max(sc + rc) if sc else 0], dtype=torch.int64, device=device)
    dist.all_reduce(big, op=dist.ReduceOp.MAX)
    return max(1, -(-int(big.item()) // cmax)), cmax


def all_to_all_v(recv: torch.Tensor, send: torch.Tensor, rc: list[int], sc: list[int], plan=None):
    """all-to-all-v of 1-D tensors: send[sum(sc[:r]) : +sc[r]] goes to rank r, recv gets rc[r]
    elements from rank r.  One all_to_all_single when every block fits the message cap (C3 on 8
    GPUs: 375 MB per pair); otherwise rounds of grouped point-to-point sends/receives on views of
    the buffers -- direct peer-to-peer over xGMI, all links busy, no staging copy."""
    world, rank = dist.get_world_size(), dist.get_rank()
    rounds, cmax = plan if plan is not None else exchange_plan(rc, sc, send.element_size(), send.device)
    if rounds == 1:
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc)
        return
    so = [sum(sc[:r]) for r in range(world)]
    ro = [sum(rc[:r]) for r in range(world)]
    recv[ro[rank]: ro[rank] + rc[rank]] = send[so[rank]: so[rank] + sc[rank]]       # own block: local copy
    for i in range(rounds):
        ops = []
        for d in range(1, world):                                  # staggered peers
            to, frm = (rank + d) % world, (rank - d) % world
            ns = max(0, min(cmax, sc[to] - i * cmax))
            nr = max(0, min(cmax, rc[frm] - i * cmax))
            if ns:
                ops.append(dist.P2POp(dist.isend, send[so[to] + i * cmax: so[to] + i * cmax + ns], to))
            if nr:
                ops.append(dist.P2POp(dist.irecv, recv[ro[frm] + i * cmax: ro[frm] + i * cmax + nr], frm))
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()


class ShardBuffers:
    """Exchange and result buffers of one rank, allocated once for a given shard (torch tensors on
    the text's device) and re-used by every build."""

    def __init__(self, info: dict, dev, dt):
        cap, loc = max(info["capacity"], 1), max(info["send_capacity"], info["local_elems"], 1)
        self.report = torch.zeros(info["n_streams"] + 2, dtype=torch.int64, device=dev)
        self.sk = torch.empty(max(info["m_local"], 1), dtype=torch.int64, device=dev)
        self.ss = torch.empty(max(info["m_local"], 1), dtype=dt, device=dev)
        self.sizes = torch.empty(info["p"], dtype=torch.int64, device=dev)
        self.send_k = torch.empty(loc, dtype=torch.int64, device=dev)
        self.send_s = torch.empty(loc, dtype=dt, device=dev)
        self.recv_k = torch.empty(cap, dtype=torch.int64, device=dev)
        self.recv_s = torch.empty(cap, dtype=dt, device=dev)
        self.SA = torch.empty(cap, dtype=dt, device=dev)
        self.LCP = torch.empty(cap, dtype=dt, device=dev)


def build_sharded(lib, T: torch.Tensor, p: int = 0, idx_bits: int | None = None, stream: int = 0, shard=None, bufs=None):
    """Collective: every rank passes the same text T (uint8 tensor on its device).
    `shard`: a Shard created earlier for the same (T, p, idx_bits) -- its device arrays are
    allocated once and re-used by every build (allocating tens of GB costs seconds); `bufs`: the
    matching ShardBuffers (the returned SA/LCP are then views of bufs.SA / bufs.LCP).

    Returns (SA_slice, LCP_slice, slice_off, info dict).  SA/LCP slices are tensors on T's
    device with the signed dtype of the same width as the unsigned indices."""
    rank, world = dist.get_rank(), dist.get_world_size()
    n = T.numel()
    idx_bits = idx_bits or (32 if n <= 0xFFFFFFFF else 64)
    dt = _idx_dtype(idx_bits)
    dev = T.device
    sh = shard if shard is not None else lib.shard(T.data_ptr(), n, p, idx_bits, rank, world, stream)
    prof = {} if os.environ.get("CAPS_SA_DIST_PROFILE") else None
    tlast = [time.perf_counter()]

    def lap(name):
        if prof is not None:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            now = time.perf_counter()
            prof[name] = prof.get(name, 0.0) + 1e3 * (now - tlast[0])
            tlast[0] = now
    try:
        inf = sh.info()
        P = inf["p"]
        B = bufs if bufs is not None else ShardBuffers(inf, dev, dt)
        # ---- direct path: same pivots on every rank, level A on the rank's tiles, ONE exchange, level B + tile sort
        attempt = 0
        while inf["direct_fallback"] == 0:
            sh.scatter(B.send_k.data_ptr(), B.send_s.data_ptr(), B.report.data_ptr())
            lap("scatter")
            reports = torch.empty(world * B.report.numel(), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(reports, B.report)
            reports_h = reports.cpu().numpy().astype(np.uint64).reshape(world, B.report.numel())
            code, sc, rc = sh.plan(reports_h)
            lap("plan")
            if code != 0:
                break
            sc = [int(x) for x in sc]
            rc = [int(x) for x in rc]
            now = sh.info()
            kb = now["key_bytes"]                                 # 4: 32-bit keys travel (2-bit texts, exchange mode), else 8
            send_k = B.send_k.view(torch.int32) if kb == 4 else B.send_k
            recv_k = B.recv_k.view(torch.int32) if kb == 4 else B.recv_k
            t0 = time.perf_counter()
            if world == 1 or not now["exchange"]:
                recv_k, recv_s = send_k, B.send_s                 # nothing travels: every rank scattered the whole text and kept
                                                                  # its own groups; level B reads the streams in place
            else:
                recv_s = B.recv_s
                plan = exchange_plan(rc, sc, max(kb, idx_bits // 8), dev)
                all_to_all_v(recv_k[:sum(rc)], send_k[:sum(sc)], rc, sc, plan)
                all_to_all_v(recv_s[:sum(rc)], B.send_s[:sum(sc)], rc, sc, plan)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            ms_exchange = 1e3 * (time.perf_counter() - t0)
            lap("exchange")
            mine = sh.sort_owned(recv_k.data_ptr(), recv_s.data_ptr(), B.SA.data_ptr(), B.LCP.data_ptr())
            worst = torch.tensor([mine], dtype=torch.int64, device=dev)
            if world > 1:                                         # (CAPS_SA_KEYS=32 gives the no-exchange mode 32-bit keys too)
                dist.all_reduce(worst, op=dist.ReduceOp.MAX)      # a slot overflow on ANY rank sends every rank round again
            lap("sort")
            if int(worst.item()) != 0:
                if attempt:
                    raise RuntimeError("the sharded direct path failed with 64-bit keys")
                attempt += 1
                sh.set_key_bits(64)
                continue
            info = _finish(sh, B.LCP, dev, rank, world)
            lap("boundary")
            total = info["recv_total"]
            info["ms_exchange"] = ms_exchange
            info["path"] = "direct"
            info["key_retry"] = attempt
            info["exchange_elems_sent"] = sum(sc) - sc[rank] if info["exchange"] else 0
            if prof is not None:
                info["host_profile_ms"] = prof
            return B.SA[:total], B.LCP[:total], info["slice_off"], info
        if inf["direct_fallback"] == 0:
            inf = dict(inf, direct_fallback=code)
        # ---- samplesort path: phase 1 + samples
        sk, ss = B.sk, B.ss
        lap("alloc_samples")
        sh.phase1(sk.data_ptr(), ss.data_ptr())
        lap("phase1")
        counts = [((r + 1) * P // world - r * P // world) * inf["ppp"] for r in range(world)]
        all_k = _all_gather_var(sk[:inf["m_local"]], counts)
        all_s = _all_gather_var(ss[:inf["m_local"]], counts)
        assert all_k.numel() == inf["m_total"]
        lap("gather_samples")
        # ---- pivots + local partition sizes
        sizes = B.sizes
        sh.pivots(all_k.data_ptr(), all_s.data_ptr(), sizes.data_ptr())
        all_sizes = torch.empty(world * P, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(all_sizes, sizes)
        all_sizes_h = all_sizes.cpu().numpy().astype(np.uint64).reshape(world, P)
        lap("pivots+sizes")
        # ---- collate into destination-major send buffers
        send_k, send_s = B.send_k, B.send_s
        sc, rc = sh.collate(all_sizes_h, send_k.data_ptr(), send_s.data_ptr())
        lap("collate")
        sc = [int(x) for x in sc]
        rc = [int(x) for x in rc]
        total = sum(rc)
        # ---- THE exchange (RCCL all-to-all-v over xGMI on GPUs)
        recv_k, recv_s = B.recv_k, B.recv_s
        t0 = time.perf_counter()
        plan = exchange_plan(rc, sc, 8, dev)                      # one agreement for both tensors (keys are the wider)
        all_to_all_v(recv_k[:total], send_k[:sum(sc)], rc, sc, plan)
        all_to_all_v(recv_s[:total], send_s[:sum(sc)], rc, sc, plan)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        ms_exchange = 1e3 * (time.perf_counter() - t0)
        lap("exchange")
        # ---- phase 2
        SA, LCP = B.SA, B.LCP
        lap("alloc_out")
        sh.phase2(recv_k.data_ptr(), recv_s.data_ptr(), SA.data_ptr(), LCP.data_ptr())
        lap("phase2")
        info = _finish(sh, LCP, dev, rank, world)
        lap("boundary")
        info["ms_exchange"] = ms_exchange
        info["exchange"] = 1
        info["path"] = "samplesort"
        info["direct_fallback"] = inf["direct_fallback"]
        info["exchange_elems_sent"] = sum(sc) - sc[rank]
        if prof is not None:
            info["host_profile_ms"] = prof
        info["send_counts"] = sc
        info["recv_counts"] = rc
        return SA[:total], LCP[:total], info["slice_off"], info
    finally:
        if shard is None:
            sh.close()


def _finish(sh, LCP, dev, rank: int, world: int) -> dict:
    """Boundary LCP between consecutive slices: every rank needs the last SA value of the nearest non-empty slice below."""
    mine = sh.last_sa()
    last = torch.tensor([mine - (1 << 64) if mine >= (1 << 63) else mine], dtype=torch.int64, device=dev)
    lasts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(lasts, last)
    prev = 0xFFFFFFFFFFFFFFFF
    for v in reversed(lasts[:rank].tolist()):
        if v != -1:
            prev = v & 0xFFFFFFFFFFFFFFFF
            break
    sh.fix_first_lcp(prev, LCP.data_ptr())
    return sh.info()


def bench_main(args, rank: int, local_rank: int, world: int):
    """bench.py's N > 1 leg: same text on every GPU, strong scaling."""
    import caps_sa_amd
    from bench import WORKLOADS, make_text, roofline, cpu_baseline
    # CAPS_SA_DIST_REHEARSAL=1 (a one-GPU box): every rank on device 0, collectives over gloo -- RCCL refuses two ranks on one
    # device.  Exercises this function's N > 1 logic (mode calibration, max over ranks, slice sums, the rank-0 line); its timings
    # mean nothing (the ranks share the GPU).
    rehearsal = os.environ.get("CAPS_SA_DIST_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if rehearsal:
        dist.init_process_group("gloo")
    elif world == 1 and "RANK" not in os.environ:        # forced single-rank run without torchrun
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    else:
        dist.init_process_group("nccl", device_id=dev)
    L = caps_sa_amd.lib()
    n_bases, kind, desc = WORKLOADS[args.workload]
    if args.bases:
        n_bases, desc = args.bases, f"custom: {args.bases} bases of kind {kind} + remapped newline, p={args.p}"
    n = n_bases + 1
    idx_bits = 32 if n <= 0xFFFFFFFF else 64
    w = idx_bits // 8
    T = make_text(torch, n_bases, args.seed, dev, kind)          # identical on every rank (same seed)
    stream = torch.cuda.current_stream().cuda_stream

    # --shard-mode: "local" = no data-path collective (every rank classifies the whole replicated text and keeps its groups),
    # "exchange" = every rank classifies every world-th tile, ONE all-to-all of (key32, sa) over xGMI, "auto" = both are run
    # once during the warm-up and the faster one (max over the ranks) is timed.  The library reads the mode when a shard is made.
    def make_shard(mode):
        if mode == "exchange":
            os.environ["CAPS_SA_SHARD_EXCHANGE"] = "1"
        else:
            os.environ.pop("CAPS_SA_SHARD_EXCHANGE", None)
        s = L.shard(T.data_ptr(), n, args.p, idx_bits, rank, world, stream)     # workspace of the rank, allocated once
        return s, ShardBuffers(s.info(), dev, _idx_dtype(idx_bits))

    mode = getattr(args, "shard_mode", "auto")
    if os.environ.get("CAPS_SA_SHARD_EXCHANGE") == "1" and mode == "auto":
        mode = "exchange"
    calib = None
    if mode == "auto" and world > 1:
        calib, calib_errors = {}, {}
        for m in ("local", "exchange"):
            # a trial that fails (the exchange variant has never run over RCCL: its buffers, its all-to-all) must not cost the run:
            # the ranks agree on the failure (errors of this kind -- out of memory, a refused message size -- hit every rank alike)
            # and the mode is left out.  The default mode failing is an error as before.
            s_ = b_ = None

            def agree(e):
                """Did the step fail on ANY rank?  Called by every rank at the same point, before the next collective."""
                f = torch.tensor([1.0 if e is not None else 0.0], dtype=torch.float64, device=dev)
                dist.all_reduce(f, op=dist.ReduceOp.MAX)
                return float(f.item()) != 0.0

            # (1) allocations: no collective inside, so a rank that fails here (out of memory) still meets the others in agree()
            err = None
            try:
                s_, b_ = make_shard(m)
            except Exception as e:          # noqa: BLE001
                err = e
            failed = agree(err)
            took = 0.0
            if not failed:
                # (2) the builds hold collectives: a rank that raises INSIDE one has left its peers waiting in it -- no agreement is
                # possible any more, and going on to the next collective would hang until the RCCL time-out: fatal
                try:
                    build_sharded(L, T, args.p, idx_bits, stream, shard=s_, bufs=b_)    # first build: allocations, RCCL connections
                    dist.barrier()
                    torch.cuda.synchronize()
                    t_ = time.perf_counter()
                    build_sharded(L, T, args.p, idx_bits, stream, shard=s_, bufs=b_)
                    dist.barrier()
                    torch.cuda.synchronize()
                    took = time.perf_counter() - t_
                except Exception as e:      # noqa: BLE001
                    import sys
                    import traceback
                    traceback.print_exc()
                    print(f"rank {rank}: the calibration build of shard mode '{m}' failed inside a collective sequence: {e!r}", file=sys.stderr, flush=True)
                    os._exit(3)
            if failed and m == "local":
                raise SystemExit(f"rank {rank}: the default shard mode could not be set up: {err!r}" if err is not None
                                 else f"rank {rank}: the default shard mode could not be set up on another rank")
            el_ = torch.tensor([took], dtype=torch.float64, device=dev)
            dist.all_reduce(el_, op=dist.ReduceOp.MAX)
            if failed:
                calib_errors[m] = repr(err) if err is not None else "failed on another rank"
            else:
                calib[m] = 1e3 * float(el_[0].item())
            if s_ is not None:
                s_.close()
            del s_, b_
            torch.cuda.empty_cache()
        mode = min(calib, key=calib.get)
        if calib_errors:
            calib = dict(calib, errors=calib_errors)
    elif mode == "auto":
        mode = "local"
    sh, bufs = make_shard(mode)

    def step():
        return build_sharded(L, T, args.p, idx_bits, stream, shard=sh, bufs=bufs)

    prewarm_steps = 0
    tp = time.perf_counter()
    while getattr(args, "prewarm_s", 0) > 0 and prewarm_steps < 64:          # (bench.py --prewarm-s: a freshly started box is slow at first)
        go = torch.tensor([1.0 if time.perf_counter() - tp < args.prewarm_s else 0.0], device=T.device)
        dist.all_reduce(go, op=dist.ReduceOp.MIN)                           # every rank leaves the loop in the same round
        if go.item() == 0.0:
            break
        out = step()
        del out
        torch.cuda.synchronize()
        prewarm_steps += 1
    for _ in range(args.warmup):
        out = step()
        del out
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos = []
    for _ in range(args.steps):
        SA, LCP, off, info = step()
        infos.append(info)
    dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    errs = None if args.no_verify else verify_sharded(L, T, SA, LCP, off, idx_bits)
    # every rank's stage times of the last build (max and min over the ranks: imbalance shows here, not in rank 0's figures)
    last = infos[-1]
    stage_keys = ("ms_level_a", "ms_scatter", "ms_exchange", "ms_sort", "ms_level_b", "ms_tile_sort", "ms_merge_passes", "ms_phase1", "ms_pivots",
                  "ms_collate", "ms_phase2")
    mine_v = torch.tensor([float(last.get(k, 0.0) or 0.0) for k in stage_keys] + [float(last["recv_total"])], dtype=torch.float64, device=dev)
    allr = torch.empty(world * mine_v.numel(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(allr, mine_v)
    allr = allr.view(world, -1)
    ranks_ms = {k: {"max": float(allr[:, i].max()), "min": float(allr[:, i].min())} for i, k in enumerate(stage_keys) if float(allr[:, i].max()) > 0}
    share = allr[:, -1]
    if int(share.sum().item()) != n:
        raise SystemExit(f"the ranks' slices hold {int(share.sum().item())} suffixes, the text has {n}")
    # rank 0 assembles the line; whatever happens there, EVERY rank reaches the barrier and leaves the process group (an
    # exception in this block used to leave the other ranks in the barrier, and the CPU baseline -- minutes at --cpu-full sizes --
    # ran while they waited inside a collective with a time-out).  The baseline is timed after the group is gone.
    out, failure = None, None
    if rank == 0:
        try:
            info = infos[-1]
            direct = all(i["path"] == "direct" for i in infos)
            mine = info["recv_total"]
            roof = None
            if not direct:    # samplesort sequence: each phase is two streaming passes (bucket scatter + tile sort) over the rank's share
                k = len(infos)
                roof = roofline({"phase1_sort_subarrays": (sum(i["ms_phase1"] for i in infos) / 2, k, sum(i["local_elems"] for i in infos), None),
                                 "phase2_sort_partitions": (sum(i["ms_phase2"] for i in infos) / 2, k, sum(i["recv_total"] for i in infos), None)}, w, "",
                                sum(i["ms_phase1"] + i["ms_pivots"] + i["ms_collate"] + i["ms_exchange"] + i["ms_phase2"] for i in infos))
                if roof:
                    roof["scope"] = (f"rank 0 of {world}, samplesort sequence: a phase is two streaming passes (bucket scatter + tile sort) over the "
                                     "rank's share; avg_launch_ms = half the phase")
            if direct:        # rank 0's kernels over ITS share of the suffixes, same convention as at N = 1
                k = len(infos)
                roof = roofline({"level_a_scatter": (sum(i["ms_level_a"] for i in infos), k, sum(i["level_a_elems"] for i in infos), None),
                                 "level_b_scatter": (sum(i["ms_level_b"] for i in infos), k, sum(i["recv_total"] for i in infos), None),
                                 "tile_sort_kernel": (sum(i["ms_tile_sort"] for i in infos), k, sum(i["recv_total"] for i in infos), None),
                                 "merge_pass_kernel": (sum(i["ms_merge_passes"] for i in infos), k if info["ms_merge_passes"] > 0 else 0,
                                                       sum(i["recv_total"] for i in infos), None)}, w, "",
                                sum(i["ms_scatter"] + i["ms_exchange"] + i["ms_sort"] for i in infos))
                if roof:
                    roof["scope"] = f"rank 0 of {world}: its kernels over its {mine} suffixes"
            ms_x = sum(i["ms_exchange"] for i in infos) / len(infos)
            sent = info["exchange_elems_sent"] * ((info.get("key_bytes", 8) if direct else 8) + w)
            keys = ("ms_scatter", "ms_exchange", "ms_sort") if direct else ("ms_phase1", "ms_pivots", "ms_collate", "ms_exchange", "ms_phase2")
            out = {
                "metric": "suffixes/sec (SA+LCP build)", "value": n / (elapsed / args.steps), "unit": "suffixes/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": prewarm_steps, "ms_per_step": 1e3 * elapsed / args.steps,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": f"u{idx_bits}",
                "data": "synthetic",
                "config": {"workload": desc, "n": n, "subproblems": info["p"], "bits_per_char": info["bits_per_char"],
                           "construction": ("direct, no exchange (same pivots on every rank; every rank scatters the whole replicated text "
                                            "into groups and keeps the ones it owns; per-group sort)" if direct and not info.get("exchange", 1)
                                            else "direct (same pivots on every rank, one scatter of every world-th tile into groups, ONE "
                                            "all-to-all of (key, sa), per-group sort)" if direct
                                            else f"samplesort (fallback reason {info.get('direct_fallback')})"),
                           "groups": info["direct_groups"], "streams_per_group": info["direct_sub"],
                           "workspace": "preallocated",
                           "parallelism": (f"{world} GPUs, one process each: text replicated, groups of partitions (slices of the suffix "
                                           "array) sharded; no data-path collective, only the ranks' reports, the agreement and the "
                                           "boundary LCPs cross RCCL" if direct and not info.get("exchange", 1) else
                                           f"{world} GPUs, one process each: text replicated, tiles of the text and groups of partitions "
                                           "sharded, one RCCL all-to-all over xGMI")},
                "rank0_ms": {k_: info[k_] for k_ in keys},
                "ranks_ms": ranks_ms,
                "suffixes_per_rank": {"max": int(share.max().item()), "min": int(share.min().item())},
                "shard_mode": "exchange" if info.get("exchange") else "local",
                "shard_mode_requested": args.shard_mode,
                "process_group": {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(),
                                  "devices_visible_to_rank0": torch.cuda.device_count() if torch.cuda.is_available() else 0},
                "exchange": {"ms": ms_x, "bytes_sent_per_rank": sent, "GBps_per_rank": (sent / 1e9) / (ms_x * 1e-3) if ms_x > 0 and sent else None,
                             "key_bytes": info.get("key_bytes", 8) if direct else 8, "key_retries": sum(i.get("key_retry", 0) for i in infos),
                             "note": ("no data-path collective: every rank scatters the whole (replicated) text and keeps the groups it owns"
                                      if direct and not info.get("exchange", 1) else
                                      "keys and indices in two all-to-all calls; region gaps (10 %) travel too; 32-bit keys on 2-bit texts")},
                "rank0_host_profile_ms": info.get("host_profile_ms"),
                "roofline": roof,
                "verify_errors": errs,
                "cpu_baseline": None,
            }
            if calib:
                out["shard_mode_calibration_ms"] = calib
        except Exception as e:              # noqa: BLE001 -- reported below, after the other ranks have been released
            failure = e
    dist.barrier()
    sh.close()
    dist.destroy_process_group()
    if failure is not None:
        raise SystemExit(f"rank 0 could not assemble the bench line: {failure!r}")
    if rank == 0:
        if not args.no_cpu_baseline:          # rank 0 times the CPU baseline at every world size (the other ranks have left)
            out["cpu_baseline"] = cpu_baseline(T, min(args.cpu_sample, n), args.p, n)
        print(json.dumps(out))
    if errs:
        raise SystemExit(f"verify_errors = {errs}")


def verify_sharded(lib, T, SA, LCP, off, idx_bits) -> int:
    """Exact check of a sharded result WITHOUT gathering it (at BASELINE config 4 the gathered arrays would be 137 GB on every
    GPU): every rank runs the slice verifier on its own slice (values in range, none twice, adjacent order, exact LCPs), checks
    the pair (last suffix of the slice below, its own first suffix) the same way -- order and the LCP at its slice head -- and the
    error counts and slice lengths are summed over the ranks.  Slices that are sorted, joined in order and n entries long in all
    hold every suffix exactly once."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = T.numel()
    dev = T.device
    cnt = SA.numel()
    errs = lib.verify_slice_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), cnt, off == 0, idx_bits=idx_bits) if cnt else 0
    ends = torch.tensor([int(SA[-1].item()) if cnt else -1, cnt, off], dtype=torch.int64, device=dev)
    allv = torch.empty(3 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allv, ends)
    allv = allv.view(world, 3).tolist()
    below = [v for v in allv[:rank] if v[1] > 0]
    if cnt and below:
        if below[-1][2] + below[-1][1] != off:
            errs += 1                                             # the slices do not join
        pair_sa = torch.tensor([below[-1][0], int(SA[0].item())], dtype=SA.dtype, device=dev)
        pair_lcp = torch.stack([LCP[0], LCP[0]])
        errs += lib.verify_slice_device(T.data_ptr(), n, pair_sa.data_ptr(), pair_lcp.data_ptr(), 2, False, idx_bits=idx_bits)
    elif cnt and off != 0:
        errs += 1
    tot = torch.tensor([errs, cnt], dtype=torch.int64, device=dev)
    dist.all_reduce(tot)
    errs_all, cnt_all = (int(x) for x in tot.tolist())
    return errs_all + (0 if cnt_all == n else 1)
