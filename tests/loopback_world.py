"""A whole world of shards in ONE process: every rank's Shard lives on the same device and the
collectives of caps_sa_dist.build_sharded are replaced by tensor copies.  RCCL cannot put two
ranks on one GPU, and a test box has one GPU: this is how the shard KERNELS are exercised at
world sizes > 1 on real hardware (the collectives themselves: tests/test_dist_gloo.py on CPU,
world size 1 over RCCL in test_gpu_parity.py, N = 2, 4, 8 in the driver's scaling run)."""
import numpy as np
import torch


def build_world(lib, T: torch.Tensor, p: int, world: int, idx_bits: int = 32):
    """Returns (SA, LCP) of the whole text, concatenated from the ranks' slices."""
    from caps_sa_dist import ShardBuffers, _idx_dtype
    n = T.numel()
    dev, dt = T.device, _idx_dtype(idx_bits)
    stream = torch.cuda.current_stream().cuda_stream if dev.type == "cuda" else 0
    shards = [lib.shard(T.data_ptr(), n, p, idx_bits, r, world, stream) for r in range(world)]
    try:
        infos = [s.info() for s in shards]
        P = infos[0]["p"]
        bufs = [ShardBuffers(i, dev, dt) for i in infos]
        sync = (lambda: torch.cuda.synchronize()) if dev.type == "cuda" else (lambda: None)
        attempt = 0
        while infos[0]["direct_fallback"] == 0:
            # direct path: scatter -> [all_gather reports] -> plan -> [all-to-all of the blocks] -> sort
            for s, B in zip(shards, bufs):
                s.scatter(B.send_k.data_ptr(), B.send_s.data_ptr(), B.report.data_ptr())
            sync()
            reports_h = torch.stack([B.report for B in bufs]).cpu().numpy().astype(np.uint64)
            plans = [s.plan(reports_h) for s in shards]
            codes = {c for c, _, _ in plans}
            assert len(codes) == 1, f"ranks disagree on the path: {codes}"
            if codes != {0}:
                break
            kbs = {s.info()["key_bytes"] for s in shards}
            assert len(kbs) == 1, f"ranks disagree on the key width: {kbs}"
            kt = torch.int32 if kbs == {4} else torch.int64           # 32-bit keys travel in exchange mode on a 2-bit text
            exchange = {s.info()["exchange"] for s in shards}
            assert len(exchange) == 1
            quant = {s.info()["direct_quantile"] for s in shards}
            assert len(quant) == 1, f"ranks disagree on the bucket mode: {quant}"
            build_world.last_quantile = quant.pop()
            sc = [[int(x) for x in a] for _, a, _ in plans]
            rc = [[int(x) for x in b] for _, _, b in plans]
            if exchange == {0}:                                       # nothing travels: every rank sorts what it scattered
                assert all(sum(a) == 0 for a in sc) and all(sum(b) == 0 for b in rc)
                worst = max(s.sort_owned(B.send_k.data_ptr(), B.send_s.data_ptr(), B.SA.data_ptr(), B.LCP.data_ptr())
                            for s, B in zip(shards, bufs))
                assert worst == 0 and kt == torch.int64
                build_world.last_key_bytes, build_world.last_key_retry, build_world.last_exchange = 8, 0, 0
                return _stitch(shards, bufs, n, sync, "direct")
            build_world.last_exchange = 1
            for r in range(world):
                assert [sc[q][r] for q in range(world)] == rc[r], "send/receive counts disagree"
                ro = 0
                for q in range(world):
                    so, c = sum(sc[q][:r]), sc[q][r]
                    bufs[r].recv_k.view(kt)[ro:ro + c] = bufs[q].send_k.view(kt)[so:so + c]
                    bufs[r].recv_s[ro:ro + c] = bufs[q].send_s[so:so + c]
                    ro += c
            worst = max(s.sort_owned(B.recv_k.data_ptr(), B.recv_s.data_ptr(), B.SA.data_ptr(), B.LCP.data_ptr())
                        for s, B in zip(shards, bufs))                # all_reduce(max)
            if worst != 0:
                assert worst == 6 and attempt == 0, (worst, attempt)   # CAPS_SA_FB_KEY32, once
                attempt += 1
                for s in shards:
                    s.set_key_bits(64)
                continue
            build_world.last_key_bytes = 4 if kt == torch.int32 else 8
            build_world.last_key_retry = attempt
            return _stitch(shards, bufs, n, sync, "direct")
        for s, B in zip(shards, bufs):
            s.phase1(B.sk.data_ptr(), B.ss.data_ptr())
        all_k = torch.cat([B.sk[:i["m_local"]] for B, i in zip(bufs, infos)])          # all_gather(samples)
        all_s = torch.cat([B.ss[:i["m_local"]] for B, i in zip(bufs, infos)])
        assert all_k.numel() == infos[0]["m_total"]
        for s, B in zip(shards, bufs):
            s.pivots(all_k.data_ptr(), all_s.data_ptr(), B.sizes.data_ptr())
        if dev.type == "cuda":
            torch.cuda.synchronize()
        all_sizes_h = torch.stack([B.sizes for B in bufs]).cpu().numpy().astype(np.uint64).reshape(world, P)
        sc, rc = [], []
        for s, B in zip(shards, bufs):
            a, b = s.collate(all_sizes_h, B.send_k.data_ptr(), B.send_s.data_ptr())
            sc.append([int(x) for x in a])
            rc.append([int(x) for x in b])
        if dev.type == "cuda":
            torch.cuda.synchronize()
        for r in range(world):                                                            # all_to_all_v
            assert [sc[q][r] for q in range(world)] == rc[r], "send/receive counts disagree"
            ro = 0
            for q in range(world):
                so = sum(sc[q][:r])
                c = sc[q][r]
                bufs[r].recv_k[ro:ro + c] = bufs[q].send_k[so:so + c]
                bufs[r].recv_s[ro:ro + c] = bufs[q].send_s[so:so + c]
                ro += c
        for s, B in zip(shards, bufs):
            s.phase2(B.recv_k.data_ptr(), B.recv_s.data_ptr(), B.SA.data_ptr(), B.LCP.data_ptr())
        return _stitch(shards, bufs, n, sync, "samplesort")
    finally:
        for s in shards:
            s.close()


def _stitch(shards, bufs, n, sync, path):
    lasts = [s.last_sa() for s in shards]                                             # all_gather(last SA)
    for r, (s, B) in enumerate(zip(shards, bufs)):
        prev = 0xFFFFFFFFFFFFFFFF
        for q in range(r - 1, -1, -1):
            if lasts[q] != 0xFFFFFFFFFFFFFFFF:
                prev = lasts[q]
                break
        s.fix_first_lcp(prev, B.LCP.data_ptr())
    sync()
    infos = [s.info() for s in shards]
    offs = [i["slice_off"] for i in infos]
    lens = [i["recv_total"] for i in infos]
    assert offs == [sum(lens[:r]) for r in range(len(shards))] and sum(lens) == n, (offs, lens, n)
    SA = torch.cat([B.SA[:c] for B, c in zip(bufs, lens)])
    LCP = torch.cat([B.LCP[:c] for B, c in zip(bufs, lens)])
    build_world.last_path = path
    build_world.last_tie_groups = sum(int(i.get("tie_groups_deferred", 0)) for i in infos)
    return SA, LCP
