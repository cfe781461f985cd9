import os, torch, torch.distributed as dist, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for n in (20_000_000, (1 << 28) - 1, (1 << 28) + 1, 400_000_000):
    for dt in (torch.int64, torch.int32):
        send = torch.arange(n, dtype=dt, device=dev)
        recv = torch.full((n,), -7, dtype=dt, device=dev)
        t0 = time.time()
        dist.all_to_all_single(recv, send, output_split_sizes=[n], input_split_sizes=[n])
        torch.cuda.synchronize()
        bad = int((recv != send).sum().item())
        print(f"n={n} dtype={dt} bytes={n*send.element_size()} mismatches={bad} first_bad={int((recv != send).nonzero()[0].item()) if bad else -1} t={time.time()-t0:.3f}s", flush=True)
        del send, recv
dist.destroy_process_group()
