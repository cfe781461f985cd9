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
