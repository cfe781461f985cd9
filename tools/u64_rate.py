"""Device-resident build with 64-bit indices on random DNA (BASELINE config 4 runs u64 on 8 GPUs; this is the
single-GPU rate of the u64 kernels at a size one GPU holds), verified on the device.  usage: u64_rate.py [n]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caps_sa_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_001
L = caps_sa_amd.lib()
g = torch.Generator(device="cuda"); g.manual_seed(42)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
T = torch.empty(n, dtype=torch.uint8, device="cuda")
for o in range(0, n, 1 << 28):
    m = min(1 << 28, n - o)
    T[o:o + m] = lut[torch.randint(0, 4, (m,), device="cuda", generator=g, dtype=torch.int64)]
SA = torch.empty(n, dtype=torch.int64, device="cuda"); LCP = torch.empty(n, dtype=torch.int64, device="cuda")
need = L.workspace_bytes(n, 8000, 64)
ws = torch.empty(need, dtype=torch.uint8, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000, idx_bits=64, workspace_ptr=ws.data_ptr(), workspace_bytes=need)
    torch.cuda.synchronize(); dt = time.time() - t0
errs = L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), idx_bits=64)
print(json.dumps({"n": n, "idx_bits": 64, "ms": 1e3 * dt, "G_suffixes_per_s": n / dt / 1e9, "verify_errors": errs, "workspace_gb": need / 1e9,
                  "tile_sort_ms": st["tile_sort_ms"], "bucket_scatter_ms": st["bucket_scatter_ms"], "ms_locate_pivots": st["ms_locate_pivots"]}))
