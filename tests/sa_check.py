"""Independent SA/LCP construction for the tests: prefix doubling (numpy) + Kasai.
O(n log^2 n) whatever the text, so it also checks the deep-LCP inputs (a^n, long runs,
tandem arrays) on which the reference algorithm -- and hence the oracle -- is quadratic.
Order: bytes as signed char, the shorter suffix first (src/Suffix_Array.cpp:75-77)."""
import numpy as np


def sa_lcp(T, idx_bits: int = 32):
    T = np.ascontiguousarray(np.asarray(T, dtype=np.uint8))
    n = T.size
    dt = np.uint32 if idx_bits == 32 else np.uint64
    if n == 0:
        return np.zeros(0, dt), np.zeros(0, dt)
    t = T.view(np.int8).astype(np.int64)
    rank = t - t.min() + 1                       # >= 1; 0 = past the end
    k = 1
    while True:
        r2 = np.zeros(n, np.int64)
        if k < n:
            r2[:n - k] = rank[k:]
        key = rank * (n + 2) + r2
        sa = np.argsort(key, kind="stable")
        ks = key[sa]
        rs = np.empty(n, np.int64)
        rs[0] = 1
        if n > 1:
            rs[1:] = 1 + np.cumsum(ks[1:] != ks[:-1])
        rank = np.empty(n, np.int64)
        rank[sa] = rs
        if rs[-1] == n or k >= n:
            break
        k *= 2
    # Kasai
    tl = T.tobytes()
    sal = sa.tolist()
    rk = rank.tolist()
    lcp = [0] * n
    h = 0
    for i in range(n):
        r = rk[i] - 1
        if r > 0:
            j = sal[r - 1]
            while i + h < n and j + h < n and tl[i + h] == tl[j + h]:
                h += 1
            lcp[r] = h
            if h:
                h -= 1
        else:
            h = 0
    return sa.astype(dt), np.array(lcp, dtype=dt)


def check_bounded(T, SA, LCP, ctx: int) -> dict:
    """Independent check of a BOUNDED-CONTEXT result (0 < ctx < n; reference src/Suffix_Array.cpp:57-95 with max_context): shares no
    code with csrc/bounded.h or the oracle.  What the reference's merges guarantee whatever their history:
      * SA is a permutation of 0 .. n-1;
      * neighbours are in order under the comparison the merge makes -- the first ctx chars, then the char behind them
        (cpp:76-77 reads T[x + n] with n = ctx), a suffix that ends first sorting first -- i.e. non-decreasing in their
        first ctx + 1 chars; ties (ctx + 1 equal chars) may stand in any order: the merge history decides, this check cannot;
      * LCP[0] = 0 and min(lcp, ctx) <= LCP[i] <= lcp for the true lcp of the neighbours (the boundary LCPs of
        compute_partition_boundary_lcp, cpp:431-447, are not cut at ctx).
    Returns counts of violations (all zero = passed).  numpy, O(n * ctx)."""
    T = np.ascontiguousarray(np.asarray(T, dtype=np.uint8))
    n = T.size
    SA = np.asarray(SA).astype(np.int64)
    LCP = np.asarray(LCP).astype(np.int64)
    res = {"not_a_permutation": 0, "order": 0, "lcp_low": 0, "lcp_high": 0, "lcp0": int(n > 0 and LCP[0] != 0)}
    if n == 0:
        return res
    seen = np.zeros(n, dtype=bool)
    ok = (SA >= 0) & (SA < n)
    seen[SA[ok]] = True
    res["not_a_permutation"] = int(n - seen.sum()) + int((~ok).sum())
    if res["not_a_permutation"] or n == 1:
        return res
    t = np.concatenate([T.view(np.int8).astype(np.int16), np.full(ctx + 2, -1000, dtype=np.int16)])   # past the end: smaller than any char
    a, b = SA[:-1], SA[1:]
    l = np.zeros(n - 1, dtype=np.int64)
    live = np.ones(n - 1, dtype=bool)
    for k in range(ctx + 1):                       # lcp of the neighbours, capped at ctx + 1
        same = live & (t[a + k] == t[b + k]) & (a + k < n) & (b + k < n)
        l += same
        live = same
    tied = l >= ctx + 1
    ca, cb = t[a + l], t[b + l]                    # first differing char (or the end marker of the one that ended)
    res["order"] = int((~tied & ~(ca < cb)).sum())
    got = LCP[1:]
    res["lcp_low"] = int((got < np.minimum(l, ctx)).sum())
    # the upper bound needs the full lcp only where the reported value exceeds the capped one
    over = np.nonzero(got > np.minimum(l, ctx))[0]
    for i in over.tolist():
        x, y, full = int(a[i]), int(b[i]), 0
        while x + full < n and y + full < n and T[x + full] == T[y + full]:
            full += 1
        if got[i] > full:
            res["lcp_high"] += 1
    return res
