// caps-sa_amd/csrc/bounded.h
//
// Bounded-context construction: Suffix_Array(T, n, p, max_context) with 0 < max_context < n.
//
// With a bounded context the reference compares two suffixes on their first max_context chars, then on the ONE char behind
// them, and calls everything that still agrees a tie, which its merge resolves by taking the element of run "Y"
// (src/Suffix_Array.cpp:71-80) -- and X and Y change roles whenever a Y element is taken (:85-92).  The result is not "the"
// suffix array of anything: the order of tied suffixes depends on the shape of every merge tree (merge_sort's halving,
// :121-127; sort_partition's halving over the p runs of a partition, :417-427), on which samples become pivots (the truncated
// gap of sample_pivots, :191-193) and on where upper_bound's truncated comparisons put them (:252-297).  So this path does not
// use the keys / buckets / tile sorts of the unbounded construction at all: it runs the reference's own sequence --
//     identity -> p subarray merge sorts -> samples -> sample merge sort -> pivots -> upper_bound matrix -> partition sizes,
//     ruler, collate (run-head LCPs reset) -> per-partition merge trees -> partition-boundary LCPs (unbounded: :440)
// -- with the reference's tree shapes, ONE THREAD PER MERGE NODE executing the sequential LCP-merge (the `m` / `l_x`
// bookkeeping of :57-95 decides which comparisons are made at all, and with ties their outcome), level by level from the
// leaves up.  All nodes of a level are independent; the top levels have few, long merges, so this is a compatibility mode
// (seconds where the unbounded build takes milliseconds), not a fast path.  Parity: UNPINNED by any reference-held vector
// (SURVEY 8c); the tests compare with tests' own CPU restatement of the same reference functions.
#pragma once
#include "kernels.h"

namespace caps {

// code of char `pos` of the packed text (pos < n)
template <int BITS> HD uint32_t code_at(const uint32_t* __restrict__ P, uint64_t pos)
{
    constexpr uint32_t CPW = TextTraits<BITS>::CPW;
    return (P[pos / CPW] >> (32 - BITS - (uint32_t)(pos % CPW) * BITS)) & ((1u << BITS) - 1u);
}

// lcp of T[a ..) and T[b ..), at most `limit` chars (the reference's LCP<8>(x, y, min_len), include/Suffix_Array.hpp:195-241);
// a + limit <= n and b + limit <= n
template <int BITS> HD uint64_t lcp_upto(const uint32_t* __restrict__ P, uint64_t a, uint64_t b, uint64_t limit)
{
    constexpr uint32_t KCH = TextTraits<BITS>::KCH;
    for (uint64_t l = 0; l < limit; l += KCH) {
        const uint64_t wa = window64<BITS>(P, a + l), wb = window64<BITS>(P, b + l);
        if (wa != wb) {
            const uint64_t d = l + (uint32_t)caps_clz64(wa ^ wb) / BITS;
            return d < limit ? d : limit;
        }
    }
    return limit;
}

// The reference's merge (src/Suffix_Array.cpp:48-109), one thread: runs X[0, lx) and Y[0, ly) with their LCP arrays -> Z, LZ.
// `own` is the run whose next LCP entry is relative to the last output (the reference's "X" after its swaps).
template <typename idx_t, int BITS>
DEV_INLINE void bounded_merge(const uint32_t* __restrict__ P, uint64_t n, uint64_t ctx, const idx_t* X, uint64_t lx, const idx_t* Y, uint64_t ly,
                              const idx_t* LX, const idx_t* LY, idx_t* Z, idx_t* LZ)
{
    const idx_t* run[2] = {X, Y};
    const idx_t* rl[2] = {LX, LY};
    const uint64_t len[2] = {lx, ly};
    uint64_t at[2] = {0, 0};
    uint32_t own = 0;
    uint64_t m = 0, k = 0;
    while (at[own] < len[own] && at[own ^ 1u] < len[own ^ 1u]) {
        const uint32_t oth = own ^ 1u;
        const uint64_t xi = (uint64_t)run[own][at[own]], yj = (uint64_t)run[oth][at[oth]];
        const uint64_t l_x = (uint64_t)rl[own][at[own]];
        bool take_own;
        if (l_x > m) {                                   // :61-64
            take_own = true;
            LZ[k] = (idx_t)l_x;
        } else if (l_x < m) {                            // :65-68
            take_own = false;
            LZ[k] = (idx_t)m;
            m = l_x;
        } else {                                         // :69-80
            const uint64_t hi = xi > yj ? xi : yj;
            const uint64_t max_n = n - hi;
            const uint64_t context = ctx < max_n ? ctx : max_n;
            const uint64_t nn = m + lcp_upto<BITS>(P, xi + m, yj + m, context - m);
            uint64_t win;
            if (nn == max_n) win = hi;                   // the shorter suffix is a prefix of the other: it goes first
            else win = code_at<BITS>(P, xi + nn) < code_at<BITS>(P, yj + nn) ? xi : yj;      // (a tie goes to Y)
            take_own = win == xi;
            LZ[k] = (idx_t)(take_own ? l_x : m);
            m = nn;
        }
        if (take_own) { Z[k] = (idx_t)xi; ++at[own]; }
        else { Z[k] = (idx_t)yj; ++at[oth]; own = oth; }
        ++k;
    }
    for (uint32_t r = 0; r < 2; ++r)                     // :98-104 (at most one run has a tail)
        for (uint64_t q = at[r]; q < len[r]; ++q) { Z[k + (q - at[r])] = run[r][q]; LZ[k + (q - at[r])] = rl[r][q]; }
    if (k < lx + ly) LZ[k] = (idx_t)m;                   // :107-108
}

// Node k of depth d of the halving tree over `cnt` items (merge_sort over elements, sort_partition over runs): the reference
// halves h = cnt / 2 at every node.  false: no such node (an ancestor was a leaf).
HD bool halving_node(uint64_t cnt, uint32_t d, uint64_t k, uint64_t* lo, uint64_t* len)
{
    uint64_t a = 0, c = cnt;
    for (uint32_t lev = 0; lev < d; ++lev) {
        if (c < 2) return false;
        const uint64_t h = c / 2;
        if ((k >> (d - 1 - lev)) & 1u) { a += h; c -= h; } else c = h;
    }
    *lo = a;
    *len = c;
    return true;
}

// One level of G independent merge sorts (merge_sort, :112-129): segment g holds the elements [seg_lo(g), + seg_len(g)) --
// g * s with the last one taking the remainder (the subarrays), or one segment (the samples).  Depth d: buf[d & 1] receives
// the merge of the two children, which sit in buf[(d + 1) & 1] (leaves: both buffers hold the input, both LCP buffers 0).
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) bounded_sort_level_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, uint64_t ctx, uint64_t total,
                                                       uint32_t G, uint64_t s, uint32_t d, idx_t* v0, idx_t* l0, idx_t* v1, idx_t* l1)
{
    PAR(tid) {
        const uint64_t per = 1ull << d;
        for (uint64_t t = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; t < (uint64_t)G * per; t += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
            const uint64_t g = t / per, k = t % per;
            const uint64_t seg0 = g * s, seglen = g + 1 < G ? s : total - seg0;
            uint64_t lo, len;
            if (!halving_node(seglen, d, k, &lo, &len) || len < 2) continue;
            const uint64_t h = len / 2, a = seg0 + lo;
            idx_t *zv = (d & 1u) ? v1 : v0, *zl = (d & 1u) ? l1 : l0;
            const idx_t *xv = (d & 1u) ? v0 : v1, *xl = (d & 1u) ? l0 : l1;
            bounded_merge<idx_t, BITS>(P, n, ctx, xv + a, h, xv + a + h, len - h, xl + a, xl + a + h, zv + a, zl + a);
        }
    }
}

// One level of the p per-partition merge trees (sort_partition, :412-428): partition j = elements [scan[j], scan[j + 1]), its
// p runs delimited by ruler[j * (p + 1) + 0 .. p]; the tree halves the RUN count.
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) bounded_partition_level_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, uint64_t ctx, uint32_t p,
                                                            const uint64_t* __restrict__ scan, const idx_t* __restrict__ ruler, uint32_t d,
                                                            idx_t* v0, idx_t* l0, idx_t* v1, idx_t* l1)
{
    PAR(tid) {
        const uint64_t per = 1ull << d;
        for (uint64_t t = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; t < (uint64_t)p * per; t += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
            const uint64_t j = t / per, k = t % per;
            uint64_t lo, cnt;
            if (!halving_node(p, d, k, &lo, &cnt) || cnt < 2) continue;
            const idx_t* S = ruler + j * ((uint64_t)p + 1) + lo;
            const uint64_t h = cnt / 2, a = scan[j] + (uint64_t)S[0];
            const uint64_t nl = (uint64_t)S[h] - (uint64_t)S[0], nr = (uint64_t)S[cnt] - (uint64_t)S[h];
            idx_t *zv = (d & 1u) ? v1 : v0, *zl = (d & 1u) ? l1 : l0;
            const idx_t *xv = (d & 1u) ? v0 : v1, *xl = (d & 1u) ? l0 : l1;
            bounded_merge<idx_t, BITS>(P, n, ctx, xv + a, nl, xv + a + nl, nr, xl + a, xl + a + nl, zv + a, zl + a);
        }
    }
}

// a[i] = b[i] = i (permute, :148-158) or a[i] = b[i] = src[i]; la[i] = lb[i] = 0 (the leaves' LCPs, :117-118)
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) bounded_init_kernel(KCTX uint64_t cnt, const idx_t* __restrict__ src, idx_t* a, idx_t* b, idx_t* la, idx_t* lb)
{
    PAR(tid) {
        for (uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; i < cnt; i += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
            const idx_t v = src ? src[i] : (idx_t)i;
            a[i] = v;
            b[i] = v;
            la[i] = 0;
            lb[i] = 0;
        }
    }
}

// out[g * m + i] = X_g[(i + 1) * gap - 1], gap = len_g / m (sample_pivots, :187-194; the top of every run stays unsampled)
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) bounded_sample_kernel(KCTX const idx_t* __restrict__ X, uint64_t total, uint32_t G, uint64_t s, uint64_t m,
                                                   idx_t* __restrict__ out)
{
    PAR(tid) {
        for (uint64_t t = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; t < (uint64_t)G * m; t += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
            const uint64_t g = t / m, i = t % m;
            const uint64_t seg0 = g * s, seglen = g + 1 < G ? s : total - seg0;
            out[t] = X[seg0 + (i + 1) * (seglen / m) - 1];
        }
    }
}

// Pm[i * (p + 1) + j + 1] = upper_bound(subarray i, pivot j) (locate_pivots :225-249, upper_bound :252-297: truncated at
// 65,536 chars and at the context; "equal as far as compared" moves right)
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) bounded_locate_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, uint64_t ctx, uint32_t p, uint64_t s,
                                                   const idx_t* __restrict__ X, const idx_t* __restrict__ pivot, idx_t* __restrict__ Pm)
{
    PAR(tid) {
        const uint64_t per = (uint64_t)p + 1;
        for (uint64_t t = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; t < (uint64_t)p * per; t += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
            const uint64_t i = t / per, jj = t % per;
            const uint64_t seg0 = i * s, len = i + 1 < p ? s : n - seg0;
            if (jj == 0) { Pm[t] = 0; continue; }
            if (jj == p) { Pm[t] = (idx_t)len; continue; }
            const uint64_t piv = (uint64_t)pivot[jj - 1], P_len = n - piv;
            const idx_t* Xi = X + seg0;
            int64_t l = -1, r = (int64_t)len;
            uint64_t soln = len, lcp_l = 0, lcp_r = 0;
            const uint64_t cutoff = 65536;
            while (r - l > 1) {
                const uint64_t c = (uint64_t)((l + r) / 2);
                const uint64_t suf = (uint64_t)Xi[c], suf_len = n - suf;
                uint64_t lc = lcp_l < lcp_r ? lcp_l : lcp_r;
                if (lc > cutoff) lc = cutoff;
                uint64_t cap = suf_len < P_len ? suf_len : P_len;
                if (ctx < cap) cap = ctx;
                if (cutoff < cap) cap = cutoff;
                lc += lcp_upto<BITS>(P, suf + lc, piv + lc, cap - lc);
                if (lc == cap) {
                    if (lc == P_len) {
                        if (P_len == suf_len) { soln = c + 1; break; }
                        r = (int64_t)c; lcp_r = lc; soln = c;
                    } else { l = (int64_t)c; lcp_l = lc; }
                } else if (code_at<BITS>(P, suf + lc) < code_at<BITS>(P, piv + lc)) { l = (int64_t)c; lcp_l = lc; }
                else { r = (int64_t)c; lcp_r = lc; soln = c; }
            }
            Pm[t] = (idx_t)soln;
        }
    }
}

// sizes[j] = sum_i (Pm[i][j + 1] - Pm[i][j]);  ruler[j][i] = sum_{i' < i} of the same (:305-316, :340-360); one thread per j
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) bounded_ruler_kernel(KCTX uint32_t p, const idx_t* __restrict__ Pm, idx_t* __restrict__ ruler,
                                                  uint64_t* __restrict__ sizes)
{
    PAR(tid) {
        const uint64_t j = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (j < p) {
            uint64_t cur = 0;
            for (uint64_t i = 0; i < p; ++i) {
                ruler[j * ((uint64_t)p + 1) + i] = (idx_t)cur;
                cur += (uint64_t)Pm[i * ((uint64_t)p + 1) + j + 1] - (uint64_t)Pm[i * ((uint64_t)p + 1) + j];
            }
            ruler[j * ((uint64_t)p + 1) + p] = (idx_t)cur;
            sizes[j] = cur;
        }
    }
}

// collate (:335-364): element x of partition j comes from run i = the run of the ruler that holds x; the LCP of every run
// head is reset to 0.  (The copy into the other buffer pair -- dup, :376-384 -- follows as two device copies.)
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) bounded_collate_kernel(KCTX uint64_t n, uint32_t p, uint64_t s, const uint64_t* __restrict__ scan,
                                                    const idx_t* __restrict__ ruler, const idx_t* __restrict__ Pm,
                                                    const idx_t* __restrict__ X, const idx_t* __restrict__ LX,
                                                    idx_t* __restrict__ a, idx_t* __restrict__ la)
{
    PAR(tid) {
        for (uint64_t x = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; x < n; x += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
            uint32_t jl = 0, jh = p;                                   // partition: largest j with scan[j] <= x
            while (jh - jl > 1) { const uint32_t mid = (jl + jh) / 2; if (scan[mid] <= x) jl = mid; else jh = mid; }
            const uint64_t j = jl, off = x - scan[j];
            const idx_t* S = ruler + j * ((uint64_t)p + 1);
            uint32_t il = 0, ih = p;                                   // run: largest i with S[i] <= off (skipping empty runs)
            while (ih - il > 1) { const uint32_t mid = (il + ih) / 2; if ((uint64_t)S[mid] <= off) il = mid; else ih = mid; }
            const uint64_t i = il, src = i * s + (uint64_t)Pm[i * ((uint64_t)p + 1) + j] + (off - (uint64_t)S[i]);
            const idx_t v = X[src], l = off == (uint64_t)S[i] ? (idx_t)0 : LX[src];
            a[x] = v;
            la[x] = l;
        }
    }
}

// LCP at the head of every partition but the first, UNBOUNDED (compute_partition_boundary_lcp, :431-447; empty partitions at
// the end make the reference read past its arrays: skipped here)
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) bounded_boundary_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, uint32_t p,
                                                     const uint64_t* __restrict__ scan, const idx_t* __restrict__ SA, idx_t* __restrict__ LCP)
{
    PAR(tid) {
        const uint64_t j = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (j >= 1 && j < p) {
            const uint64_t at = scan[j];
            if (at != 0 && at < n) LCP[at] = (idx_t)deep_lcp<BITS, true>(P, n, (uint64_t)SA[at - 1], (uint64_t)SA[at], 0);
        }
    }
}

}  // namespace caps
