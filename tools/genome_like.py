"""Genome-like synthetic DNA on the GPU: an order-k Markov chain with skewed transition
probabilities (composition bias: k-mer frequencies spread over ~2 orders of magnitude) plus
planted repeats (copies of earlier segments with point mutations).  No real genome is available
offline; this stands in for the *shape* of one when tuning the bucket maps."""
import torch


def markov_dna(n, order=5, seed=7, device="cuda", skew=0.5, repeats=0.02, repeat_len=5000, mut=0.01):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    states = 4 ** order
    # Dirichlet(skew) rows via gamma sampling
    torch.manual_seed(seed)                       # the transition matrix comes from the CPU generator: same text every run
    gam = torch.distributions.Gamma(torch.full((states, 4), skew), torch.ones(states, 4)).sample().to(device)
    cdf = torch.cumsum(gam / gam.sum(1, keepdim=True), 1)
    # chains run in parallel from random states and are concatenated (stationary enough)
    lanes = 1 << 16
    steps = (n + lanes - 1) // lanes
    out = torch.empty((steps, lanes), dtype=torch.uint8, device=device)
    st = torch.randint(0, states, (lanes,), device=device, generator=g)
    for t in range(steps):
        u = torch.rand(lanes, device=device, generator=g)
        c = (u.unsqueeze(1) > cdf[st]).sum(1).clamp_(max=3)
        out[t] = c.to(torch.uint8)
        st = (st * 4 + c) % states
    seq = out.t().contiguous().view(-1)[:n].clone()     # lane-major: each lane is one contiguous chain
    # planted repeats
    n_rep = int(n * repeats / repeat_len)
    if n_rep:
        src = torch.randint(0, n - repeat_len, (n_rep,), device=device, generator=g)
        dst = torch.randint(0, n - repeat_len, (n_rep,), device=device, generator=g)
        for s, d in zip(src.tolist(), dst.tolist()):
            seg = seq[s:s + repeat_len].clone()
            m = torch.rand(repeat_len, device=device, generator=g) < mut
            seg[m] = torch.randint(0, 4, (int(m.sum()),), device=device, generator=g, dtype=torch.uint8)
            seq[d:d + repeat_len] = seg
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out_t = torch.empty(n, dtype=torch.uint8, device=device)
    for o in range(0, n, 1 << 28):                # in pieces: seq.long() of 3e9 elements would take 24 GB
        out_t[o:o + (1 << 28)] = lut[seq[o:o + (1 << 28)].long()]
    return out_t


def plant_genome_repeats(T, n_bases, seed=5):
    """GRCh38-shaped repeat content on top of a text of ACGT letters (in place).  Sizes are those of
    the 3e9-base workload, scaled down with the text (factor n_bases / 3e9, at least 1/16):
      * a 20 Mb tandem array of a 171-base monomer, every copy 1.5 % diverged (alpha satellite) -- half of it as a
        higher-order repeat: a unit of 12 monomers 20 % apart from each other, the units 0.5 % apart (period 2052:
        far beyond the run table's 16), so thousands of suffixes share their first 32 bases and differ ~200 chars on;
      * 100,000 copies of a 300-base family, each 12 % from the consensus (Alu-like), at disjoint places;
      * five 100-kb duplications at 1 % divergence and ONE exact 50-kb duplicate (segmental duplications).
    Deterministic for a given (n_bases, seed): no two planted pieces overlap."""
    dev = T.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    f = max(1.0 / 16, min(1.0, n_bases / 3e9))

    def rnd(m):
        return lut[torch.randint(0, 4, (m,), device=dev, generator=g, dtype=torch.int64)]

    def mutate(seg, rate):
        seg = seg.clone()
        m = torch.rand(seg.numel(), device=dev, generator=g) < rate
        seg[m] = rnd(int(m.sum()))
        return seg

    mono = rnd(171)
    half = int(10_000_000 * f)
    a0 = n_bases // 10                                               # [0.10 n, 0.10 n + 2 * half): the satellite
    assert a0 + 2 * half < n_bases // 4
    T[a0:a0 + half] = mutate(mono.repeat(half // 171 + 1)[:half], 0.015)
    unit = torch.cat([mutate(mono, 0.2) for _ in range(12)])
    T[a0 + half:a0 + 2 * half] = mutate(unit.repeat(half // unit.numel() + 1)[:half], 0.005)
    copies = int(100_000 * f)                                        # [0.30 n, 0.90 n): the 300-base family, one copy per stride
    stride = (6 * n_bases // 10) // copies
    assert stride >= 1000
    cons = rnd(300)
    pos = (n_bases * 3 // 10) + torch.arange(copies, device=dev, dtype=torch.int64) * stride \
        + torch.randint(0, stride - 300, (copies,), device=dev, generator=g)
    fam = mutate(cons.repeat(copies), 0.12).view(copies, 300)
    T[(pos[:, None] + torch.arange(300, device=dev)).view(-1)] = fam.view(-1)
    dl = int(100_000 * max(f, 0.5))                                  # duplications: sources in [0.02 n, 0.08 n), copies in [0.92 n, 0.99 n)
    for k in range(5):
        s0 = n_bases * 2 // 100 + k * (n_bases // 100)
        d0 = n_bases * 92 // 100 + k * (n_bases // 100)
        T[d0:d0 + dl] = mutate(T[s0:s0 + dl], 0.01)
    s0, d0 = n_bases * 8 // 100, n_bases * 98 // 100
    T[d0:d0 + 50_000] = T[s0:s0 + 50_000]
    return T


if __name__ == "__main__":
    import os, sys, time, json
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import caps_sa_amd
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_001
    L = caps_sa_amd.lib()
    T = markov_dna(n)
    if os.environ.get("NBLOCKS"):                 # N-block stand-ins (the CLI maps N to G, src/main.cpp:61-68): long single-letter runs
        g = torch.Generator(device="cuda")
        g.manual_seed(99)
        for ln, cnt in ((2_000_000, 1), (500_000, 5), (50_000, 100)):
            for a in torch.randint(0, n - ln, (cnt,), device="cuda", generator=g).tolist():
                T[a:a + ln] = ord("G")
    bits = int(os.environ.get("IDX", "32"))          # IDX=64: the u64 kernels
    dt_idx = torch.int32 if bits == 32 else torch.int64
    SA = torch.empty(n, dtype=dt_idx, device="cuda")
    LCP = torch.empty(n, dtype=dt_idx, device="cuda")
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        st = L.build_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), p=8000, idx_bits=bits)
        torch.cuda.synchronize()
        dt = time.time() - t0
    errs = L.verify_device(T.data_ptr(), n, SA.data_ptr(), LCP.data_ptr(), idx_bits=bits)
    keep = ("long_runs", "slot_splits", "slot_splits_redone", "ms_total", "ms_sort_subarrays", "ms_merge_partitions", "merge_passes_phase1", "merge_passes_phase2",
            "max_partition", "tile_sort_ms", "merge_pass_ms", "bucket_scatter_ms")
    print(json.dumps({"n": n, "idx_bits": bits, "wall_ms": 1e3 * dt, "G_suffixes_per_s": n / dt / 1e9, "verify_errors": errs,
                      "max_lcp": int(LCP.max().item()), "mean_lcp": float(LCP.double().mean().item()),
                      **{k: st[k] for k in keep}}))
