// caps-sa_amd/csrc/kernels.h
//
// The HIP kernels of the suffix-array / LCP-array construction path, written for
// gfx950 (CDNA4, wave64, 160 KiB LDS per CU) in the phase style of kernel_lang.h.
//
// Reference rows -> kernels:
//   a2 LCP<8> comparator                      -> text.h (window64 / pair_lcp / suffix_less)
//   a4 merge_sort, a5 sort_subarrays          -> tile_sort_kernel (+ merge passes below)
//   a3 merge (LCP-merge of two runs)          -> merge_partition_kernel + merge_pass_kernel
//   a6 sample_pivots / select_pivots          -> sample_kernel, pick_pivots_kernel (+ the sorts)
//   a7 upper_bound, a8 locate_pivots          -> locate_kernel
//   a9 partition_sub_subarrays                -> partition_sizes_kernel, scan_sizes_kernel,
//                                                collate_kernel
//   a10 merge_sub_subarrays / sort_partition  -> tile_sort_kernel + merge passes per partition
//   a11 compute_partition_boundary_lcp        -> finalize_kernel (segment-head LCPs while gathering SA/LCP)
//
// Data layout in HBM: struct-of-arrays per element -- key (u64, first 64/BITS chars of
// the suffix), sa (idx_t, text position), and -- written only by the sort step that
// completes a segment -- lcp (idx_t, lcp with the predecessor in the sorted segment; 0 for
// the segment head, the reference's convention at src/Suffix_Array.cpp:117-118,354-356).  A "segment" is an independent sort problem
// (a subarray in phase 1, the sample set, a partition in phase 2); segments are
// described by seg_start[G+1] and are cut into tiles of TILE_E elements aligned to the
// segment start.
#pragma once
#include "kernel_lang.h"
#include "text.h"

namespace caps {

// Tile geometry (compile-time; -DCAPS_TILE_E=... etc. build the variants compared in
// profiles/).  Default: 4096-element tiles, 1024 threads, two workgroups per CU
// (LDS 48 KiB at 32-bit indices, 64 KiB at 64-bit; 8 waves per SIMD => 64 VGPRs).
#ifndef CAPS_TILE_E
#define CAPS_TILE_E 4096
#endif
#ifndef CAPS_TILE_NT
#define CAPS_TILE_NT 1024
#endif
#ifndef CAPS_TILE_WAVES
#define CAPS_TILE_WAVES 8          /* min waves per SIMD = workgroups per CU * CAPS_TILE_NT / 256 */
#endif
#ifndef CAPS_LOCK_K
#define CAPS_LOCK_K 4              /* binary searches one thread advances in lockstep */
#endif
constexpr uint32_t TILE_E = CAPS_TILE_E;       // elements per tile (one workgroup)
constexpr uint32_t TILE_NT = CAPS_TILE_NT;     // threads per tile workgroup
constexpr uint32_t TILE_EPT = TILE_E / TILE_NT;
constexpr uint32_t LOCK_K = CAPS_LOCK_K < TILE_EPT ? CAPS_LOCK_K : TILE_EPT;
#define TILE_WAVES_PER_SIMD CAPS_TILE_WAVES
#ifdef CAPS_NO_WAVES_BOUND   /* experiment: let the compiler pick the register budget */
#undef LAUNCH_BOUNDS2
#define LAUNCH_BOUNDS2(n, w) LAUNCH_BOUNDS(n)
#endif
static_assert(TILE_E % TILE_NT == 0 && TILE_EPT % LOCK_K == 0, "tile geometry");
#ifndef CAPS_TILE_BINS_DIV
#define CAPS_TILE_BINS_DIV 2
#endif
constexpr uint32_t TILE_BINS_ = TILE_E / CAPS_TILE_BINS_DIV;   // bins of the in-LDS bucket sort of one tile

// Segment/tile descriptor shared by the tile-granular kernels.
// One record per tile, written once by tile_map_kernel: a tile kernel starts with ONE load of
// its record instead of the dependent chain tile -> segment -> segment bounds (a workgroup is
// a few microseconds of work, so every dependent global load at its start shows).
struct alignas(16) TileInfo {
    uint64_t s0, s1;     // segment range
    uint32_t tl;         // tile index inside the segment
    uint32_t g;          // segment
    uint32_t pad_[2];
};

struct SegDesc {
    const uint64_t* seg_start;   // [G+1] element offsets of the segments
    const uint32_t* tile_off;    // [G+1] exclusive scan of ceil(len/TILE_E); tile_off[G] = #tiles
    const TileInfo* tile_rec;    // [#tiles] record of each tile
    uint32_t G;
};

// End of segment g: seg_start[g + 1] for contiguous segments; seg_end[g] when the segments sit in
// fixed-capacity regions with gaps between them (the groups of the direct path, pipeline.h).
DEV_INLINE uint64_t seg_end_of(const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ seg_end, uint64_t g)
{
    return seg_end ? seg_end[g] : seg_start[g + 1];
}

DEV_INLINE TileInfo tile_info(const SegDesc& sd, uint32_t b) { return sd.tile_rec[b]; }

// Geometry of the run pair a tile belongs to during a merge pass with run length R
// (R is a multiple of TILE_E).  single_la != ~0: the segment is ONE pair whose first run
// has length single_la (used by the stand-alone merge entry point).
struct PairInfo {
    uint64_t a0;         // start of run A (= start of the merged output)
    uint64_t la, lb;     // run lengths
    uint64_t d0;         // first output diagonal of this tile inside the pair
};

DEV_INLINE PairInfo pair_info(const TileInfo& t, uint64_t R, uint64_t single_la)
{
    PairInfo p;
    if (single_la != ~0ull) {
        p.a0 = t.s0;
        p.la = single_la;
        p.lb = (t.s1 - t.s0) - single_la;
        p.d0 = (uint64_t)t.tl * TILE_E;
        return p;
    }
    const uint64_t tpp = (2 * R) / TILE_E;             // tiles per pair
    const uint64_t q = t.tl / tpp;
    p.a0 = t.s0 + q * 2 * R;
    const uint64_t left = t.s1 - p.a0;
    p.la = left < R ? left : R;
    p.lb = left - p.la < R ? left - p.la : R;
    p.d0 = ((uint64_t)t.tl - q * tpp) * TILE_E;
    return p;
}

// ----------------------------------------------------------------------------------
// Input preparation
// ----------------------------------------------------------------------------------

// Which byte values occur in T: 256-bit presence set (8 x u32), OR-reduced.
// 16 bytes per lane per load; a byte marks its flag in a 256-entry LDS table with a plain
// store (idempotent, so colliding lanes need no atomic); the table is folded into the global
// bit set once per workgroup.  T must be readable in 16-byte units up to n rounded up (the
// callers' buffers are; the tail is handled bytewise).
GLOBAL_FN LAUNCH_BOUNDS(256) alphabet_kernel(KCTX const uint8_t* __restrict__ T, uint64_t n, uint32_t* __restrict__ present)
{
    SHARED_ARRAY(uint32_t, flags, 256);
    PAR(tid) { flags[tid & 255] = 0; }
    SYNC();
    PAR(tid) {
        const uint64_t stride = (uint64_t)K_GRID_DIM * K_BLOCK_DIM * 16;
        const bool aligned = (reinterpret_cast<uintptr_t>(T) & 15) == 0;
        for (uint64_t i = ((uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid) * 16; i < n; i += stride) {
            if (aligned && i + 16 <= n) {
                const uint32_t* q = reinterpret_cast<const uint32_t*>(T + i);
                const uint32_t w[4] = {q[0], q[1], q[2], q[3]};
                UNROLL
                for (int k = 0; k < 4; ++k) {
                    flags[w[k] & 255] = 1;
                    flags[(w[k] >> 8) & 255] = 1;
                    flags[(w[k] >> 16) & 255] = 1;
                    flags[w[k] >> 24] = 1;
                }
            } else {
                const uint64_t e = i + 16 < n ? i + 16 : n;
                for (uint64_t j = i; j < e; ++j) flags[T[j]] = 1;
            }
        }
    }
    SYNC();
    PAR(tid) {
        if (tid < 256 && flags[tid]) ATOMIC_OR_U32(&present[tid >> 5], 1u << (tid & 31));
    }
}

// Pack raw bytes into BITS-wide codes, big-endian inside each 32-bit word (text.h).
// lut[256] maps byte -> code (staged in LDS).  One thread per 16 input bytes (one 16-byte
// load -> one word at BITS = 2, four words at BITS = 8); words past the text are 0.
// bad != null: the LUT comes from a SAMPLE of the text (prepare_text); a byte it does not know (code 0xFF) raises bad[0] and
// the caller redoes the preparation with the alphabet of the whole text.
template <int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) pack_kernel(KCTX const uint8_t* __restrict__ T, uint64_t n, const uint8_t* __restrict__ lut,
                                         uint32_t* __restrict__ P, uint64_t n_words, uint32_t* __restrict__ bad)
{
    constexpr uint32_t CPW = TextTraits<BITS>::CPW;
    constexpr uint32_t WPT = 16 / CPW;                    // output words per 16 input bytes
    SHARED_ARRAY(uint8_t, slut, 256);
    PAR(tid) { slut[tid & 255] = lut[tid & 255]; }
    SYNC();
    PAR(tid) {
        const uint64_t stride = (uint64_t)K_GRID_DIM * K_BLOCK_DIM;
        const uint64_t n_units = (n_words + WPT - 1) / WPT;
        const bool aligned = (reinterpret_cast<uintptr_t>(T) & 15) == 0;
        for (uint64_t u = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; u < n_units; u += stride) {
            const uint64_t base = u * 16;
            uint8_t c[16];
            if (aligned && base + 16 <= n) {
                const uint32_t* q = reinterpret_cast<const uint32_t*>(T + base);
                UNROLL
                for (int k = 0; k < 4; ++k) {
                    const uint32_t w = q[k];
                    c[4 * k] = (uint8_t)w; c[4 * k + 1] = (uint8_t)(w >> 8); c[4 * k + 2] = (uint8_t)(w >> 16); c[4 * k + 3] = (uint8_t)(w >> 24);
                }
            } else {
                UNROLL
                for (int k = 0; k < 16; ++k) c[k] = base + k < n ? T[base + k] : 0;
            }
            uint32_t unknown = 0;
            UNROLL
            for (uint32_t wo = 0; wo < WPT; ++wo) {
                uint32_t v = 0;
                UNROLL
                for (uint32_t k = 0; k < CPW; ++k) {
                    const uint64_t i = base + wo * CPW + k;
                    const uint32_t code = i < n ? slut[c[wo * CPW + k]] : 0u;
                    unknown |= code;
                    v |= (code & ((1u << BITS) - 1u)) << (32 - BITS * (k + 1));
                }
                if (u * WPT + wo < n_words) P[u * WPT + wo] = v;
            }
            if (BITS == 2 && bad && (unknown & 0x80u)) bad[0] = 1;       // benign race: every writer stores 1
        }
    }
}

// Run table (text.h), step 1: one entry per aligned block of KCH chars.  A block with smallest
// period d whose d-periodic stretch ends inside the next block (or at the end of the text) gets
// its final value; one whose stretch covers the whole next block gets RUN_LINKED (= "the value
// of the block after me": that block then has the same smallest period) and raises *flag.
// Blocks that are aperiodic or reach past the text get 0.
template <int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) run_blocks_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, uint64_t entries,
                                               uint64_t* __restrict__ R, uint64_t* __restrict__ flag)
{
    constexpr uint32_t KCH = TextTraits<BITS>::KCH;
    PAR(tid) {
        const uint64_t full = n / KCH;                            // blocks that lie inside the text
        const uint64_t stride = (uint64_t)K_GRID_DIM * K_BLOCK_DIM;
        bool linked = false;
        for (uint64_t b = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; b < entries; b += stride) {
            uint64_t r = 0;
            if (b < full) {
                const uint64_t w0 = ((uint64_t)P[2 * b] << 32) | P[2 * b + 1];
                const uint32_t d = block_period<BITS>(w0);
                if (d) {
                    const uint64_t w1 = ((uint64_t)P[2 * b + 2] << 32) | P[2 * b + 3];
                    const uint32_t sh = d * BITS;
                    const uint64_t x = w1 ^ ((w0 << (64 - sh)) | (w1 >> sh));   // T[q] ^ T[q - d] over the next block
                    const uint64_t cand = x ? KCH * (b + 1) + (uint32_t)caps_clz64(x) / BITS : KCH * (b + 2);
                    if (cand >= n) r = ((uint64_t)d << 56) | n;
                    else if (!x) { r = RUN_LINKED; linked = true; }
                    else r = ((uint64_t)d << 56) | cand;
                }
            }
            R[b] = r;
        }
        if (linked) *flag = 1;                                     // benign race: every writer stores 1
    }
}

// Run table, steps 2-4: R[b] = R[first block >= b that is not RUN_LINKED], a suffix scan with the
// operator "mine unless linked".  Chunks of RUN_CHUNK blocks: (2) first unlinked value of every
// chunk, (3) one workgroup carries them from the right, (4) every chunk resolves its entries.
// All three return at once when no block is linked (any text without a periodic stretch of
// 2 * KCH chars: step 1 has already written final values).
GLOBAL_FN LAUNCH_BOUNDS(256) run_chunk_heads_kernel(KCTX const uint64_t* __restrict__ R, uint64_t entries,
                                                    const uint64_t* __restrict__ flag, uint64_t* __restrict__ head)
{
    SHARED_ARRAY(uint32_t, first, 1);
    if (!*flag) return;
    PAR(tid) { if (tid == 0) first[0] = RUN_CHUNK; }
    SYNC();
    const uint64_t base = (uint64_t)K_BLOCK_IDX * RUN_CHUNK;
    PAR(tid) {
        for (uint32_t k = 0; k < RUN_PER; ++k) {
            const uint64_t b = base + (uint64_t)tid * RUN_PER + k;
            if (b < entries && R[b] != RUN_LINKED) {
                ATOMIC_MIN_U32(&first[0], tid * RUN_PER + k);
                break;
            }
        }
    }
    SYNC();
    PAR(tid) {
        if (tid == 0) head[K_BLOCK_IDX] = first[0] < RUN_CHUNK ? R[base + first[0]] : RUN_LINKED;
    }
}

// carry[c] = first unlinked value right of chunk c (exclusive); one workgroup, right to left.
GLOBAL_FN LAUNCH_BOUNDS(1024) run_carry_kernel(KCTX const uint64_t* __restrict__ head, uint32_t chunks,
                                               const uint64_t* __restrict__ flag, uint64_t* __restrict__ carry)
{
    SHARED_ARRAY(uint64_t, v, 2 * 1024);
    SHARED_ARRAY(uint64_t, from_right, 1);
    if (!*flag) return;
    PAR(tid) { if (tid == 0) from_right[0] = 0; }              // the last entries of the table are 0, never linked
    SYNC();
    for (uint64_t hi = chunks; hi > 0; hi -= (hi < 1024 ? hi : 1024)) {
        const uint32_t cnt = (uint32_t)(hi < 1024 ? hi : 1024);
        const uint64_t lo = hi - cnt;                              // this round: chunks [lo, hi)
        PAR(tid) { v[tid] = tid < cnt ? head[lo + tid] : RUN_LINKED; }
        SYNC();
        uint32_t src = 0;
        for (uint32_t d = 1; d < 1024; d <<= 1) {                  // inclusive suffix scan, ping-pong halves
            PAR(tid) {
                const uint64_t mine = v[src * 1024 + tid];
                v[(src ^ 1) * 1024 + tid] = (mine == RUN_LINKED && tid + d < 1024) ? v[src * 1024 + tid + d] : mine;
            }
            SYNC();
            src ^= 1;
        }
        PAR(tid) {
            if (tid < cnt) {
                const uint64_t right = tid + 1 < cnt ? v[src * 1024 + tid + 1] : RUN_LINKED;   // inclusive value of the chunk to my right
                carry[lo + tid] = right != RUN_LINKED ? right : from_right[0];
            }
        }
        SYNC();
        PAR(tid) {
            if (tid == 0 && v[src * 1024] != RUN_LINKED) from_right[0] = v[src * 1024];
        }
        SYNC();
    }
}

// Also: longest[0] = max over blocks of (end of its stretch - start of the block).
GLOBAL_FN LAUNCH_BOUNDS(256) run_resolve_kernel(KCTX uint64_t* __restrict__ R, uint64_t entries, const uint64_t* __restrict__ flag,
                                                const uint64_t* __restrict__ carry, uint32_t kch, uint64_t* __restrict__ longest)
{
    SHARED_ARRAY(uint64_t, v, 2 * RUN_NT);
    SHARED_ARRAY(uint64_t, lmax, 1);
    if (!*flag) return;
    PAR(tid) { if (tid == 0) lmax[0] = 0; }
    const uint64_t base = (uint64_t)K_BLOCK_IDX * RUN_CHUNK;
    TL_DECL(uint64_t, r, RUN_PER);
    PAR(tid) {
        uint64_t head = RUN_LINKED;                                // first unlinked value among my blocks
        for (uint32_t k = 0; k < RUN_PER; ++k) {
            const uint64_t b = base + (uint64_t)tid * RUN_PER + k;
            TL(r, tid, k) = b < entries ? R[b] : 0;
        }
        for (int k = RUN_PER - 1; k >= 0; --k) if (TL(r, tid, k) != RUN_LINKED) head = TL(r, tid, k);
        v[tid] = head;
    }
    SYNC();
    uint32_t src = 0;
    for (uint32_t d = 1; d < RUN_NT; d <<= 1) {
        PAR(tid) {
            const uint64_t mine = v[src * RUN_NT + tid];
            v[(src ^ 1) * RUN_NT + tid] = (mine == RUN_LINKED && tid + d < RUN_NT) ? v[src * RUN_NT + tid + d] : mine;
        }
        SYNC();
        src ^= 1;
    }
    PAR(tid) {
        uint64_t right = tid + 1 < RUN_NT ? v[src * RUN_NT + tid + 1] : RUN_LINKED;
        if (right == RUN_LINKED) right = carry[K_BLOCK_IDX];
        uint64_t ext = 0;
        for (int k = RUN_PER - 1; k >= 0; --k) {
            const uint64_t b = base + (uint64_t)tid * RUN_PER + k;
            if (TL(r, tid, k) == RUN_LINKED) {
                if (b < entries) R[b] = right;
                const uint64_t e = (right & RUN_POS_MASK) - b * kch;
                ext = e > ext ? e : ext;
            } else right = TL(r, tid, k);
        }
        if (ext >= RUN_LONG) ATOMIC_MAX_LDS_U64(&lmax[0], ext);
    }
    SYNC();
    PAR(tid) { if (tid == 0 && lmax[0]) ATOMIC_MAX_U64(longest, lmax[0]); }
}

// key/sa for an arbitrary list of suffix positions (sample sort, stand-alone entry points).
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) make_keys_kernel(KCTX const uint32_t* __restrict__ P, const idx_t* __restrict__ sa, uint64_t cnt,
                                              uint64_t* __restrict__ key)
{
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < cnt) key[i] = window64<BITS>(P, (uint64_t)sa[i]);
    }
}

// ----------------------------------------------------------------------------------
// Segment descriptors
// ----------------------------------------------------------------------------------

// seg_start[g] = min(g * s, n_total) for g < G, seg_start[G] = n_total
// (subarrays of the reference: src/Suffix_Array.cpp:171-173, the last one takes n % p).
GLOBAL_FN LAUNCH_BOUNDS(256) uniform_segments_kernel(KCTX uint64_t* __restrict__ seg_start, uint32_t G, uint64_t s, uint64_t n_total)
{
    PAR(tid) {
        const uint64_t g = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (g < G) seg_start[g] = g * s;
        else if (g == G) seg_start[g] = n_total;
    }
}

// seg_start[0 .. 1] = {s0, s1}: ONE segment anywhere in the arrays (a letter-run bucket, pipeline.h sort_run_buckets)
GLOBAL_FN LAUNCH_BOUNDS(64) segment_range_kernel(KCTX uint64_t* __restrict__ seg_start, uint64_t s0, uint64_t s1)
{
    PAR(tid) { if (tid == 0 && K_BLOCK_IDX == 0) { seg_start[0] = s0; seg_start[1] = s1; } }
}

// Exclusive scan of sizes[G] into seg_start[G+1] (single workgroup).
// Partition offsets of the reference's serial scan (src/Suffix_Array.cpp:319-330).
// gfx950: per-wave inclusive scan with wave64 shuffles, wave totals combined through LDS.
GLOBAL_FN LAUNCH_BOUNDS(1024) scan_sizes_kernel(KCTX const uint64_t* __restrict__ sizes, uint32_t G, uint64_t* __restrict__ seg_start)
{
#ifdef CAPS_EMUL
    uint64_t run = 0;
    for (uint32_t g = 0; g < G; ++g) { seg_start[g] = run; run += sizes[g]; }
    seg_start[G] = run;
#else
    __shared__ uint64_t wave_tot[16];
    __shared__ uint64_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < G; base += 1024) {
        const uint32_t g = base + tid;
        const uint64_t v = g < G ? sizes[g] : 0;
        uint64_t x = v;                                    // inclusive scan inside the wave
        UNROLL
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t y = __shfl_up(x, d, 64);
            if ((int)lane >= d) x += y;
        }
        if (lane == 63) wave_tot[wv] = x;
        __syncthreads();
        uint64_t off = carry_s;
        for (uint32_t w = 0; w < wv; ++w) off += wave_tot[w];
        if (g < G) seg_start[g] = off + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = off + x;
        __syncthreads();
    }
    if (tid == 0) seg_start[G] = carry_s;
#endif
}

// tile_off[G+1] = exclusive scan of ceil(len/TILE_E); out2[0] = #tiles, out2[1] = max segment
// length (out2 must be zeroed before the launch).  Single workgroup; LDS Hillis-Steele scan
// per chunk of 1024 segments.
// skip (optional): segments with skip[g] != 0 get no tiles and do not count for the maximum (they are sorted elsewhere:
// the letter-run buckets of run_bucket_mark_kernel)
GLOBAL_FN LAUNCH_BOUNDS(1024) seg_prepare_kernel(KCTX const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ seg_end,
                                                 uint32_t G, uint32_t* __restrict__ tile_off, uint64_t* __restrict__ out2,
                                                 const uint8_t* __restrict__ skip)
{
    SHARED_ARRAY(uint32_t, buf, 2048);
    uint32_t carry = 0;
    for (uint32_t base = 0; base < G; base += 1024) {
        PAR(tid) {
            const uint32_t g = base + tid;
            uint64_t len = 0;
            if (g < G && !(skip && skip[g])) len = seg_end_of(seg_start, seg_end, g) - seg_start[g];
            buf[tid] = (uint32_t)((len + TILE_E - 1) / TILE_E);
            if (len) ATOMIC_MAX_U64(&out2[1], len);
        }
        SYNC();
        uint32_t src = 0;
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            PAR(tid) {
                buf[(src ^ 1) * 1024 + tid] = buf[src * 1024 + tid] + (tid >= d ? buf[src * 1024 + tid - d] : 0u);
            }
            SYNC();
            src ^= 1;
        }
        PAR(tid) {
            const uint32_t g = base + tid;
            if (g < G) {
                const uint64_t len = skip && skip[g] ? 0 : seg_end_of(seg_start, seg_end, g) - seg_start[g];
                tile_off[g] = carry + buf[src * 1024 + tid] - (uint32_t)((len + TILE_E - 1) / TILE_E);
            }
        }
        SYNC();
        carry += buf[src * 1024 + 1023];     // block-uniform
        SYNC();
    }
    PAR(tid) {
        if (tid == 0) { tile_off[G] = carry; out2[0] = carry; }
    }
}

// ---- multi-workgroup exclusive scan (segment counts in the millions: the buckets) ------
// block_scan_kernel: workgroup b scans its chunk of SCAN_CHUNK inputs (exclusive, local) and
// records the chunk total; the totals are scanned by scan_sizes_kernel; add_offsets_kernel
// adds them back and writes the grand total to out[n].
constexpr uint32_t SCAN_CHUNK = 4096;

template <typename OutT>
GLOBAL_FN LAUNCH_BOUNDS(1024) block_scan_kernel(KCTX const uint64_t* __restrict__ in, uint64_t n, OutT* __restrict__ out,
                                                uint64_t* __restrict__ sums)
{
    const uint64_t base = (uint64_t)K_BLOCK_IDX * SCAN_CHUNK;
#ifdef CAPS_EMUL
    uint64_t run = 0;
    for (uint32_t i = 0; i < SCAN_CHUNK && base + i < n; ++i) { const uint64_t v = in[base + i]; out[base + i] = (OutT)run; run += v; }
    sums[K_BLOCK_IDX] = run;
#else
    __shared__ uint64_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint64_t v[4], sum = 0;
    UNROLL
    for (uint32_t i = 0; i < 4; ++i) {
        const uint64_t idx = base + (uint64_t)tid * 4 + i;
        v[i] = idx < n ? in[idx] : 0;
        sum += v[i];
    }
    uint64_t x = sum;
    UNROLL
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = __shfl_up(x, d, 64);
        if ((int)lane >= d) x += y;
    }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    uint64_t run = x - sum;
    for (uint32_t w = 0; w < wv; ++w) run += wave_tot[w];
    UNROLL
    for (uint32_t i = 0; i < 4; ++i) {
        const uint64_t idx = base + (uint64_t)tid * 4 + i;
        if (idx < n) out[idx] = (OutT)run;
        run += v[i];
    }
    if (tid == 1023) sums[blockIdx.x] = run;
#endif
}

template <typename OutT>
GLOBAL_FN LAUNCH_BOUNDS(1024) add_offsets_kernel(KCTX OutT* __restrict__ out, uint64_t n, const uint64_t* __restrict__ offs,
                                                 uint32_t n_chunks)
{
    const uint64_t base = (uint64_t)K_BLOCK_IDX * SCAN_CHUNK;
    const uint64_t off = offs[K_BLOCK_IDX];
    PAR(tid) {
        for (uint32_t i = tid; i < SCAN_CHUNK; i += K_BLOCK_DIM)
            if (base + i < n) out[base + i] = (OutT)((uint64_t)out[base + i] + off);
        if (K_BLOCK_IDX == 0 && tid == 0) out[n] = (OutT)offs[n_chunks];
    }
}

// cnt[g] = ceil(len_g / TILE_E); out2[1] = max len (out2 zeroed before the launch).
GLOBAL_FN LAUNCH_BOUNDS(256) tile_count_kernel(KCTX const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ seg_end,
                                               uint32_t G, uint64_t* __restrict__ cnt, uint64_t* __restrict__ out2,
                                               const uint8_t* __restrict__ skip)
{
    SHARED_ARRAY(uint64_t, mx, 1);
    PAR(tid) { if (tid == 0) mx[0] = 0; }
    SYNC();
    PAR(tid) {
        const uint64_t g = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (g < G) {
            const uint64_t len = skip && skip[g] ? 0 : seg_end_of(seg_start, seg_end, g) - seg_start[g];
            cnt[g] = (len + TILE_E - 1) / TILE_E;
            if (len) ATOMIC_MAX_LDS_U64(&mx[0], len);
        }
    }
    SYNC();
    PAR(tid) { if (tid == 0 && mx[0]) ATOMIC_MAX_U64(&out2[1], mx[0]); }
}

// out2[0] = tile_off[G] (number of tiles), for the host read-back.
GLOBAL_FN LAUNCH_BOUNDS(64) tile_total_kernel(KCTX const uint32_t* __restrict__ tile_off, uint32_t G, uint64_t* __restrict__ out2)
{
    PAR(tid) { if (tid == 0 && K_BLOCK_IDX == 0) out2[0] = tile_off[G]; }
}

// tile_rec[b] for the segment g with tile_off[g] <= b < tile_off[g+1].
GLOBAL_FN LAUNCH_BOUNDS(256) tile_map_kernel(KCTX const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ seg_end,
                                             const uint32_t* __restrict__ tile_off, uint32_t G, TileInfo* __restrict__ tile_rec)
{
    PAR(tid) {
        const uint64_t b = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (b < tile_off[G]) {
            uint32_t lo = 0, hi = G;               // largest g in [0, G) with tile_off[g] <= b
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) / 2;
                if (tile_off[mid] <= b) lo = mid; else hi = mid;
            }
            TileInfo t;
            t.s0 = seg_start[lo];
            t.s1 = seg_end_of(seg_start, seg_end, lo);
            t.tl = (uint32_t)b - tile_off[lo];
            t.g = lo;
            t.pad_[0] = t.pad_[1] = 0;
            tile_rec[b] = t;
        }
    }
}


// K independent lower-bound searches over LDS-resident sorted (key, sa) runs:
// on return lo[k] = lo[k] + #{elements of [lo[k], hi[k]) that are < (key[k], sa[k])}.
//
// The hot part is branch-free and runs a FIXED number of steps (log2(top), top = a power of
// two > the longest range, block-uniform): step s probes element lo + s - 1 and advances by
// s when it is smaller.  The K probes of a step are independent, so their LDS reads are in
// flight together.  Only keys are compared there (equal key = "not smaller"); a suffix whose
// key also occurs in the range finishes with the exact comparator over the remaining
// candidates -- never on random DNA, routinely on repeats.
template <typename idx_t, int BITS, int K, bool RUNS = true>
DEV_INLINE void multi_lower_bound(const uint32_t* __restrict__ P, uint64_t n, const uint64_t* skey, const idx_t* ssa,
                                  const uint64_t (&key)[K], const idx_t (&sa)[K], uint32_t (&lo)[K], const uint32_t (&hi)[K],
                                  uint32_t top)
{
    for (uint32_t s = top >> 1; s >= 1; s >>= 1) {
        UNROLL
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = lo[k] + s - 1;
            const uint64_t mk = skey[idx < TILE_E ? idx : TILE_E - 1];
            if (idx < hi[k] && mk < key[k]) lo[k] += s;
        }
    }
    UNROLL
    for (int k = 0; k < K; ++k) {
        if (lo[k] < hi[k] && skey[lo[k]] == key[k]) {              // key tie: exact order among [lo, hi)
            uint32_t a = lo[k], b = hi[k];
            while (a < b) {
                const uint32_t mid = (a + b) >> 1;
                if (suffix_less<BITS, RUNS>(P, n, skey[mid], (uint64_t)ssa[mid], key[k], (uint64_t)sa[k])) a = mid + 1;
                else b = mid;
            }
            lo[k] = a;
        }
    }
}

// Deferred ties (pipeline.h SortOpts::defer_ties; msd_* kernels below).  A sort that completes segments of THE suffix array may leave
// the order inside a group of EQUAL keys open: the comparison sort of a tile orders such a group by position (descending: as if
// the text ended behind the key), the merge passes merge by key alone (stably: run A before run B), and the pass that emits the
// LCPs writes TIE_SENTINEL (all ones: no LCP has that value) for every element whose key equals its predecessor's.  The groups
// are then ordered by re-keying them KCH chars deeper, level by level (msd_refine, pipeline.h) -- one window per suffix and level
// instead of a walk through the text per comparison (thousands of suffixes of a tandem array share their first 32 bases and
// agree with their neighbours for ~850 chars: 50 of the 142 ms of the GRCh38-shaped workload were such comparisons).
// K bounds over LDS-resident sorted keys, by key only: lower bound (#{< key}) or upper bound (#{<= key}) per search.
template <int K>
DEV_INLINE void multi_key_bound(const uint64_t* skey, const uint64_t (&key)[K], const bool (&upper)[K], uint32_t (&lo)[K],
                                const uint32_t (&hi)[K], uint32_t top)
{
    for (uint32_t s = top >> 1; s >= 1; s >>= 1) {
        UNROLL
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = lo[k] + s - 1;
            const uint64_t mk = skey[idx < TILE_E ? idx : TILE_E - 1];
            if (idx < hi[k] && (upper[k] ? mk <= key[k] : mk < key[k])) lo[k] += s;
        }
    }
}

// smallest power of two strictly greater than x (x < 2^31)
HD uint32_t pow2_above(uint32_t x) { return 1u << (32 - (x ? __builtin_clz(x) : 32)); }

struct BucketParams {        // per parent segment
    uint64_t kmin;           // smallest key of the segment's range
    uint64_t range;          // largest key - smallest key
    uint32_t shift;          // normalisation: ((key - kmin) << shift) >> 32 spans [0, R32], R32 in [2^31, 2^32)
    uint32_t B;
    uint32_t post;           // bucket = mulhi32(d32, m) >> post, clamped to B - 1
    uint32_t m;
};

// Monotone linear map of [kmin, kmin + range] onto B buckets, in 32-bit arithmetic (the maps run once
// per suffix in four kernels, and a 64 x 64 -> 128 multiply is ~10 quarter-rate VALU instructions):
// d32 = the top 32 bits of the difference, range normalised to [2^31, 2^32); bucket = d32 * B / (R32 + 1).
// The factor B * 2^32 / (R32 + 1) lies in [B, 2B]; it is kept as a 31..32-bit fixed-point number
// (scaled by 2^post), so bucket boundaries are exact to ~2^-31 of the range -- a multiplier with only
// log2(B) significant bits shifts the boundaries by up to 1/B of the range, and the key ranges that
// bucket_ranges_kernel hands to the tile sort no longer match the buckets.
HD BucketParams make_bucket_params(uint64_t kmin, uint64_t kmax, uint32_t B)
{
    BucketParams q;
    q.kmin = kmin;
    q.B = B;
    q.range = kmax > kmin ? kmax - kmin : 0;
    q.shift = q.range ? (uint32_t)caps_clz64(q.range) : 0u;
    const uint64_t r32 = (q.range << q.shift) >> 32;                    // in [2^31, 2^32) unless range == 0
    uint32_t bits = 1;
    while ((2ull * B) >> bits) ++bits;                                  // 2B < 2^bits
    q.post = bits < 32 ? 32 - bits : 0;
    // floor(B * 2^(32 + post) / (r32 + 1)) < 2^32; the numerator is < 2^63 (B < 2^(bits - 1))
    const uint64_t num = (uint64_t)B << (32 + q.post);
    q.m = q.range ? (uint32_t)(num / (r32 + 1)) : 0u;
    return q;
}

DEV_INLINE uint32_t bucket_of(const BucketParams& bp, uint64_t key)
{
    if (bp.B <= 1) return 0;
    if (key <= bp.kmin) return 0;
    const uint64_t d = key - bp.kmin;
    if (d >= bp.range) return bp.B - 1;                               // at or above the top of the nominal range
    const uint32_t d32 = (uint32_t)((d << bp.shift) >> 32);
    const uint32_t b = caps_umul32hi(d32, bp.m) >> bp.post;
    return b < bp.B ? b : bp.B - 1;
}

// Exclusive scan of NBINS counters in LDS, in place; h[NBINS] = total.
// gfx950: every thread owns NBINS / TILE_NT consecutive counters; wave64 shuffle scan of the
// per-thread sums, wave totals combined through LDS (two barriers).
template <uint32_t NBINS>
DEV_INLINE void block_exclusive_scan(KCTX uint32_t* h)
{
#ifdef CAPS_EMUL
    (void)kctx_;
    // the same barrier structure as the product code below (the race detector of tests/emul counts on it): every counter is read,
    // barrier, every sum is written, barrier
    std::vector<uint32_t> c(NBINS);
    for (uint32_t i = 0; i < NBINS; ++i) c[i] = h[i];
    SYNC_LDS();
    uint32_t run = 0;
    for (uint32_t i = 0; i < NBINS; ++i) { h[i] = run; run += c[i]; }
    h[NBINS] = run;
    SYNC_LDS();
#else
    constexpr uint32_t BPT = NBINS / TILE_NT;
    static_assert(BPT >= 1 && BPT * TILE_NT == NBINS, "counters per thread");
    __shared__ uint32_t wave_tot[TILE_NT / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t v[BPT], sum = 0;
    UNROLL
    for (uint32_t i = 0; i < BPT; ++i) { v[i] = h[tid * BPT + i]; sum += v[i]; }
    // inclusive scan over the wave with DPP moves (rows of 16 lanes: shifts by 1, 2, 4, 8; then the last lane of row 0 / 2 into
    // row 1 / 3, and lane 31 into rows 2 and 3): six move + add pairs.  (__shfl_up costs an index computation and a
    // ds_bpermute per step: the four scans of a tile were ~10 % of tile_sort_eq_kernel's instructions.)
    uint32_t x = sum;
#define CAPS_DPP_ADD(ctrl, rows) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xF, false)
    CAPS_DPP_ADD(0x111, 0xF);        // row_shr:1
    CAPS_DPP_ADD(0x112, 0xF);        // row_shr:2
    CAPS_DPP_ADD(0x114, 0xF);        // row_shr:4
    CAPS_DPP_ADD(0x118, 0xF);        // row_shr:8
    CAPS_DPP_ADD(0x142, 0xA);        // row_bcast:15 -> rows 1 and 3
    CAPS_DPP_ADD(0x143, 0xC);        // row_bcast:31 -> rows 2 and 3
#undef CAPS_DPP_ADD
    if (lane == 63) wave_tot[wv] = x;
    SYNC_LDS();
    uint32_t run = x - sum;
    if (TILE_NT / 64 <= 16) {
        // the waves' totals: lanes 0 .. 15 hold one each, prefix by DPP within the row, this wave's through readlane
        const uint32_t wt = lane < TILE_NT / 64 ? wave_tot[lane] : 0u;
        uint32_t ws = wt;
#define CAPS_DPP_ADD(ctrl) ws += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ws, ctrl, 0xF, 0xF, false)
        CAPS_DPP_ADD(0x111);
        CAPS_DPP_ADD(0x112);
        CAPS_DPP_ADD(0x114);
        CAPS_DPP_ADD(0x118);
#undef CAPS_DPP_ADD
        const int wvs = __builtin_amdgcn_readfirstlane((int)wv);
        run += (uint32_t)__builtin_amdgcn_readlane((int)ws, wvs) - (uint32_t)__builtin_amdgcn_readlane((int)wt, wvs);
    } else {
        for (uint32_t w = 0; w < wv; ++w) run += wave_tot[w];
    }
    UNROLL
    for (uint32_t i = 0; i < BPT; ++i) { h[tid * BPT + i] = run; run += v[i]; }
    if (tid == TILE_NT - 1) h[NBINS] = run;
    SYNC_LDS();
#endif
}
DEV_INLINE void block_exclusive_scan_bins(KCTX uint32_t* hist) { block_exclusive_scan<TILE_BINS_>(KCTX_PASS hist); }

// Final destination of a sort whose result is THE suffix / LCP array (phase 2): SA and LCP slots
// of the caller plus, per sorted segment, its first and last (key, sa) -- the boundary records
// from which head_lcp_kernel computes the LCP at every segment head (a11).
template <typename idx_t> struct FinalOut {
    idx_t* sa = nullptr;
    idx_t* lcp = nullptr;
    uint64_t* first_key = nullptr;
    uint64_t* last_key = nullptr;
    idx_t* first_sa = nullptr;
    idx_t* last_sa = nullptr;
};

// Bins of the in-LDS bucket sort of one tile, and the bin occupancy above which the tile
// falls back to the merge levels.
#ifdef CAPS_EMUL
extern "C" void caps_emul_count_tile(bool fast, bool known_range);   // test statistics (tests/emul/emul_lib.cpp)
extern "C" void caps_emul_count_tile2(bool two_level_ok);
extern "C" void caps_emul_count_tile3(bool equalised_ok);
#endif
constexpr uint32_t TILE_BINS = TILE_BINS_;
// Largest bin the exact in-bin ranking (quadratic in the bin) accepts: in tile_sort_kernel, whose linear map either
// works (uniform keys: bins of 2-3) or is hopeless, and in tile_sort_eq_kernel, the last stop before the comparison sort.
#ifndef CAPS_TILE_BIN_LIMIT
#define CAPS_TILE_BIN_LIMIT 48    /* genome-like 3e9: 16 / 24 / 48 / 128 -> 300 / 301 / 299 / 320 ms (the eq kernel sorts crowded tiles cheaper than a quadratic ranking of big bins) */
#endif
#ifndef CAPS_EQ_BIN_LIMIT
#define CAPS_EQ_BIN_LIMIT 128
#endif
constexpr uint32_t TILE_BIN_LIMIT = CAPS_TILE_BIN_LIMIT;
constexpr uint32_t EQ_BIN_LIMIT = CAPS_EQ_BIN_LIMIT;

// ----------------------------------------------------------------------------------
// a4/a5: tile sort -- one workgroup sorts up to TILE_E suffixes in LDS and emits the
// sorted run with its LCP array (reference: merge_sort, src/Suffix_Array.cpp:112-129,
// called per subarray at :171-173).
//
// FROM_TEXT: the tile is TILE_E consecutive text positions (phase 1), element i of the
// arrays being text position text_base + i (text_base != 0 on a shard that owns a slice of
// the subarrays); keys are cut from the packed text.  Otherwise (key, sa) pairs are read
// from in_key/in_sa.
// In-LDS algorithm: bottom-up merge sort where every element finds its output slot
// by a binary search in the sibling run (rank merge): no divergent serial merge,
// ragged runs need no padding.  LCPs are produced once, at the end, from adjacent
// keys (text only on equal keys).
// LDS: TILE_E x (8 + sizeof(idx_t)).
// ----------------------------------------------------------------------------------
// LCP of two neighbours of a sorted tile from their keys, 64-bit or 32-bit (text.h); cs: the 32-bit keys' shift
// min(l, chars the shorter of the suffixes a, b has), in the width of the indices: with 32-bit indices n < 2^32, and the
// 64-bit max / subtract / min of the generic form are a tenth of tile_sort_kernel's instructions per suffix
template <typename idx_t>
DEV_INLINE uint64_t lcp_capped(uint32_t l, uint64_t n, idx_t a, idx_t b)
{
    if (sizeof(idx_t) == 4) {
        const uint32_t room = (uint32_t)n - (uint32_t)(a > b ? a : b);
        return l < room ? l : room;
    }
    const uint64_t room = n - (uint64_t)(a > b ? a : b);
    return l < room ? l : room;
}
template <int BITS, bool RUNS, typename idx_t>
DEV_INLINE uint64_t tile_pair_lcp(const uint32_t* __restrict__ P, uint64_t n, uint64_t ka, idx_t a, uint64_t kb, idx_t b, uint32_t)
{
    const uint64_t x = ka ^ kb;
    if (x) return lcp_capped<idx_t>((uint32_t)caps_clz64(x) / BITS, n, a, b);
    return pair_lcp<BITS, RUNS>(P, n, ka, (uint64_t)a, kb, (uint64_t)b);
}
template <int BITS, bool RUNS, typename idx_t>
DEV_INLINE uint64_t tile_pair_lcp(const uint32_t* __restrict__ P, uint64_t n, uint32_t ka, idx_t a, uint32_t kb, idx_t b, uint32_t cs)
{
    return pair_lcp32<BITS, RUNS>(P, n, ka, (uint64_t)a, kb, (uint64_t)b, cs);
}

// ... or TIE_SENTINEL for a pair of equal keys when ties are deferred (bflag != null: "Deferred ties" above)
template <typename idx_t> HD idx_t tie_sentinel() { return (idx_t)~(idx_t)0; }
template <int BITS, bool RUNS, typename idx_t>
DEV_INLINE uint64_t deferring_pair_lcp(const uint32_t* __restrict__ P, uint64_t n, uint64_t* __restrict__ bflag, uint32_t g, uint64_t ka, idx_t a,
                                       uint64_t kb, idx_t b)
{
    if (bflag && ka == kb) { bflag[g] = 1; return (uint64_t)tie_sentinel<idx_t>(); }       // (benign race: every writer stores 1)
    return tile_pair_lcp<BITS, RUNS>(P, n, ka, a, kb, b, 0u);
}

// Measurement only (make variant VARIANT_DEFS=-DCAPS_PHASE_CLOCK; never in the product build): thread 0 of every workgroup
// adds the cycles between two marks to caps_phase_clock[slot] -- which barrier-separated phase of a tile sort takes the time
// (tools/phase_clock.py reads the counters through caps_sa_hip_phase_clock()).  The results of the kernels are unchanged.
#if defined(CAPS_PHASE_CLOCK) && !defined(CAPS_EMUL)
__device__ unsigned long long caps_phase_clock[32];
#define PHASE_T0() unsigned long long pc_t_ = clock64()
#define PHASE_MARK(i) do { const unsigned long long t_ = clock64(); if (threadIdx.x == 0) atomicAdd(&caps_phase_clock[i], t_ - pc_t_); pc_t_ = clock64(); } while (0)
#else
#define PHASE_T0() ((void)0)
#define PHASE_MARK(i) ((void)0)
#endif

// Debugging only (make variant VARIANT_DEFS=-DCAPS_EQ_CHECK): every index tile_sort_eq_kernel derives from data it was handed (queue
// entries, tile records, slots, bins, suffix positions) is range-checked; a failed check is counted in caps_eq_check[id] (read
// out through caps_sa_hip_eq_check()) and the access is skipped or clamped instead of made -- so a wild index shows up as a
// number rather than as a memory access fault.  Never in the product build.
#if defined(CAPS_EQ_CHECK) && !defined(CAPS_EMUL)
__device__ unsigned int caps_eq_check[64];
#define EQ_OK(ok, id) ((ok) ? true : (atomicAdd(&caps_eq_check[id], 1u), false))
#else
#define EQ_OK(ok, id) (true)
#endif
#define EQ_CHK(ok, id) ((void)EQ_OK((ok), (id)))

// Where a tile of a sort over slots reads its keys: the slot array, or -- a bucket that outgrew its slot (speculative split by
// knots: spill_gather_kernel / spill_place_kernel have put it together there) -- its place in the output array, sorted in place.
// (32-bit keys live in slots only: such a sort gives up when a slot overflows.)
DEV_INLINE const uint64_t* tile_src(const uint64_t* in, const uint64_t* out, bool from_out) { return from_out ? out : in; }
DEV_INLINE const uint32_t* tile_src(const uint32_t* in, const uint64_t*, bool) { return in; }

// Shared pieces of the two tile sort kernels (macros: they use the kernels' TL registers).
#define TILE_SORT_PROLOGUE                                                                                      \
    const uint32_t b = K_BLOCK_IDX;                                                                             \
    if (b >= sd.tile_off[sd.G]) return;                                                                         \
    const uint32_t g = sd.tile_rec[b].g;                                                                          \
    const TileInfo t = tile_info(sd, b);                                                                        \
    const uint64_t start = t.s0 + (uint64_t)t.tl * TILE_E;                                                      \
    const uint32_t cnt = (uint32_t)(t.s1 - start < TILE_E ? t.s1 - start : TILE_E);                             \
    /* LCPs are only needed from the sort that completes a segment: lcp_mode 0 = never,                      */ \
    /* 1 = when the whole segment is this tile (otherwise its final merge pass emits them).                  */ \
    const bool with_lcp = lcp_mode != 0 && (t.s1 - t.s0) <= TILE_E;                                             \
    /* fin.sa != null: a segment completed here goes straight to the caller's SA / LCP arrays                */ \
    /* (its head LCP is filled in by head_lcp_kernel from the boundary records).                             */ \
    const bool direct = with_lcp && fin.sa != nullptr;                                                          \
    /* slot_cap != 0: the input of segment (bucket) g sits in its fixed-capacity slot, see bucket_scatter_kernel.  */ \
    /* (A bucket that outgrew its slot -- speculative split by knots -- was put together at its place in the OUTPUT */ \
    /* arrays: TILE_OUTGROWN.  The kernels at the head of the queue chain never see such a tile: the tile table    */ \
    /* they are launched with shows it EMPTY (hot_tiles_kernel), they pass it on as unfinished, and the builds      */ \
    /* behind them -- launched with the true table -- read it there, tile_src below.  No code for it here: a test   */ \
    /* in tile_sort_eq_kernel's plain build cost ten more SGPR spills and 5 % of its time.)                         */ \
    const uint64_t in0 = slot_cap ? (uint64_t)g * slot_cap : start;
#define TILE_OUTGROWN (slot_cap != 0 && t.s1 - t.s0 > (uint64_t)slot_cap)

#define TILE_SRC_KEY in_key
#define TILE_SRC_SA in_sa
#define TILE_SORT_LOAD                                                                                          \
    PAR(tid) {                                                                                                  \
        if (tid == 0) { kmm[0] = ~0ull; kmm[1] = 0; flag[0] = 0; }                                              \
        for (uint32_t i = tid; i <= TILE_BINS; i += K_BLOCK_DIM) hist[i] = 0;                                   \
        UNROLL                                                                                                  \
        for (uint32_t k = 0; k < TILE_EPT; ++k) {                                                               \
            const uint32_t e = tid + k * TILE_NT;                                                               \
            if (e < cnt) {                                                                                      \
                uint64_t key;                                                                                   \
                idx_t sa;                                                                                       \
                if (FROM_TEXT) {                                                                                \
                    key = window64<BITS>(P, text_base + start + e);                                             \
                    sa = (idx_t)(text_base + start + e);                                                        \
                } else if (TILE_KEYS_FROM_TEXT) {   /* 32-bit keys in the slots: the 64-bit key is cut from the text */ \
                    sa = STREAM_LOAD(&TILE_SRC_SA[in0 + e]);                                                    \
                    key = window64<BITS>(P, (uint64_t)sa);                                                      \
                } else {                                                                                        \
                    key = STREAM_LOAD(&TILE_SRC_KEY[in0 + e]);                                                  \
                    sa = STREAM_LOAD(&TILE_SRC_SA[in0 + e]);                                                    \
                }                                                                                               \
                TL(rk, tid, k) = (decltype(TL(rk, tid, k) + 0))key;                                             \
                TL(rs, tid, k) = sa;                                                                            \
            }                                                                                                   \
        }                                                                                                       \
    }

// Bin map of the tile: prepared in advance when the tile belongs to a key-range bucket (seg_map, from
// bucket_ranges_kernel: its 64-bit division would otherwise be repeated by every thread of every
// tile), otherwise from the min / max over the tile.
/* TILE_SYNC: the barrier of the tile kernels' shared pieces.  Everything the threads of a tile hand each other goes through  */
/* LDS, so tile_sort_eq_kernel (which keeps spilled registers in scratch and emits straight before its next tile) uses the  */
/* LDS-scope barrier: __syncthreads() would also wait for every scratch store and every result store of the wave.           */
#define TILE_SYNC() SYNC()
#define TILE_SORT_RANGE                                                                                         \
    const bool known_range = seg_map != nullptr;                                                                \
    if (!known_range) {                                                                                         \
        TILE_SYNC(); /* kmm initialised */                                                                      \
        PAR(tid) {                                                                                              \
            uint64_t mn = ~0ull, mx = 0;                                                                        \
            UNROLL                                                                                              \
            for (uint32_t k = 0; k < TILE_EPT; ++k) {                                                           \
                const uint32_t e = tid + k * TILE_NT;                                                           \
                if (e < cnt) {                                                                                  \
                    const uint64_t key = (uint64_t)TL(rk, tid, k);                                              \
                    mn = key < mn ? key : mn;                                                                   \
                    mx = key > mx ? key : mx;                                                                   \
                }                                                                                               \
            }                                                                                                   \
            BLOCK_MINMAX_U64(&kmm[0], &kmm[1], mn, mx);                                                         \
        }                                                                                                       \
    }                                                                                                           \
    TILE_SYNC();                                                                                                \
    const BucketParams tb = known_range ? seg_map[g] : make_bucket_params(kmm[0], kmm[1], TILE_BINS); /* block-uniform */

// registers -> final slots TL(rd) in LDS
#define TILE_SORT_PLACE_FINAL                                                                                   \
    PAR(tid) {                                                                                                  \
        UNROLL                                                                                                  \
        for (uint32_t k = 0; k < TILE_EPT; ++k) {                                                               \
            const uint32_t e = tid + k * TILE_NT;                                                               \
            if (e < cnt) {                                                                                      \
                const uint32_t d = TL(rd, tid, k);                                                              \
                skey[d] = TL(rk, tid, k);                                                                       \
                ssa[d] = TL(rs, tid, k);                                                                        \
            }                                                                                                   \
        }                                                                                                       \
    }                                                                                                           \
    TILE_SYNC();

// sorted tile in LDS -> HBM (+ LCPs from adjacent keys, + boundary records)
#define TILE_SORT_EMIT TILE_SORT_EMIT_(0u)
/* OV: an lcp already known for slot e (0 = none: from the keys, the text on equal keys) */
#define TILE_EMIT_LCP_(ka, a, kb, b) tile_pair_lcp<BITS, TILE_RUNS>(P, n, ka, a, kb, b, TILE_KEY_SHIFT)
#define TILE_SORT_EMIT_(OV)                                                                                     \
    PAR(tid) {                                                                                                  \
        UNROLL                                                                                                  \
        for (uint32_t k = 0; k < TILE_EPT; ++k) {                                                               \
            const uint32_t e = tid + k * TILE_NT;                                                               \
            if (e < cnt) {                                                                                      \
                const uint64_t key = skey[e];                                                                   \
                const idx_t sa = ssa[e];                                                                        \
                uint64_t l = 0;                                                                                 \
                if (with_lcp && e) {                                                                            \
                    const uint32_t ov_ = (OV);                                                                  \
                    l = ov_ ? ov_ : TILE_EMIT_LCP_(skey[e - 1], ssa[e - 1], skey[e], sa);                          \
                }                                                                                               \
                if (direct) {                                                                                   \
                    STREAM_STORE(&fin.sa[start + e], sa);                                                       \
                    STREAM_STORE(&fin.lcp[start + e], (idx_t)l);                                                \
                    if (e == 0) { fin.first_key[g] = key; fin.first_sa[g] = sa; }                               \
                    if (e == cnt - 1) { fin.last_key[g] = key; fin.last_sa[g] = sa; }                           \
                } else {                                                                                        \
                    out_key[start + e] = key;                                                                   \
                    out_sa[start + e] = sa;                                                                     \
                    if (with_lcp) out_lcp[start + e] = (idx_t)l;                                                \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }

// ---- tile_sort_kernel: interpolation bucket sort in LDS ------------------------------
// bin = monotone linear map of the key onto TILE_BINS bins over the tile's key range; a
// counting sort by bin (LDS histogram + scan) places every element next to the few others of
// its bin, and a short exact scan of its own bin fixes the order.  ~100 VALU instructions per
// suffix instead of ~1000 for a comparison sort.  Completes the tile when no bin holds more than
// TILE_BIN_LIMIT elements (always on keys that are roughly uniform in their range: random DNA,
// buckets of the bucketing stage); otherwise it writes nothing and sets redo[tile], and
// tile_sort_general_kernel sorts the tile.
// KT = uint32_t: the slots hold 32-bit keys (text.h key32_of; kshift[bucket] = the shift of the bucket's group).  Only
// completed segments written straight to SA / LCP (`direct`) exist then; a tile this kernel cannot finish is re-sorted by the
// kernels below from 64-bit keys cut from the text.
template <typename idx_t, int BITS, bool FROM_TEXT, typename KT = uint64_t>
GLOBAL_FN LAUNCH_BOUNDS2(TILE_NT, TILE_WAVES_PER_SIMD) tile_sort_kernel(KCTX SegDesc sd, const uint32_t* __restrict__ P, uint64_t n,
                                                  uint64_t text_base, uint32_t lcp_mode, uint32_t slot_cap, const KT* in_key,
                                                  const idx_t* in_sa, uint64_t* out_key, idx_t* out_sa, idx_t* out_lcp,
                                                  FinalOut<idx_t> fin, const BucketParams* __restrict__ seg_map,
                                                  uint32_t* __restrict__ redo, const uint8_t* __restrict__ kshift)
{
    constexpr bool TILE_RUNS = false;        // ties here are shallow (suffix_less_tie_bounded), so are the LCPs
    constexpr bool K32 = sizeof(KT) == 4;
    constexpr bool TILE_KEYS_FROM_TEXT = false;
    static_assert(!K32 || !FROM_TEXT, "32-bit keys come from the slots of level B");
    TILE_SORT_PROLOGUE
    const uint32_t TILE_KEY_SHIFT = K32 ? kshift[g] : 0u;                                  // block-uniform
    const uint32_t tie_from = K32 ? (TILE_KEY_SHIFT + 32u) / BITS : TextTraits<BITS>::KCH;   // chars two equal keys stand for
    SHARED_ARRAY(KT, skey, TILE_E);
    SHARED_ARRAY(idx_t, ssa, TILE_E);
    SHARED_ARRAY(uint32_t, hist, TILE_BINS + 1);
    SHARED_ARRAY(uint64_t, kmm, 2);          // min / max key of the tile
    SHARED_ARRAY(uint32_t, flag, 1);         // a bin overflowed
    // SLOT_ORDER: from the rank phase on a thread works on the elements in slots tid + k * TILE_NT instead of the ones it
    // loaded (their bins: sbin).  The lanes of a wave then hold neighbouring slots -- mostly one bin or two: the keys they scan
    // are the same LDS words (one read serves all), the loops have the same length, and the final placement moves every
    // element a few slots, neighbours to neighbours, where the loaded elements sat in 64 bins all over the tile (half of
    // this kernel's LDS cycles were bank conflicts).  64-bit indices: no room in LDS (two workgroups per CU) for sbin.
    constexpr bool SLOT_ORDER = sizeof(idx_t) == 4;
    SHARED_ARRAY(uint16_t, sbin, SLOT_ORDER ? TILE_E : 1);
    // ... and no second placement (as in tile_sort_eq_kernel): the element that knows its final position d notes perm[d] = its slot,
    // the emit phase reads keys and indices through perm -- one barrier, two LDS writes and a read per element less
    SHARED_ARRAY(uint16_t, perm, SLOT_ORDER ? TILE_E : 1);
    TL_DECL(KT, rk, TILE_EPT);
    TL_DECL(idx_t, rs, TILE_EPT);
    TL_DECL(uint32_t, rd, TILE_EPT);
    TL_DECL(uint32_t, rb, TILE_EPT);

    PHASE_T0();
    TILE_SORT_LOAD
    TILE_SORT_RANGE
    PHASE_MARK(0);                                             // load (+ range)
    bool fast = cnt > 1 && tb.range > 0;
    if (fast) {
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    const uint32_t bin = bucket_of(tb, TL(rk, tid, k));
                    const uint32_t r = FETCH_ADD_U32(&hist[bin], 1u);
                    if (r >= TILE_BIN_LIMIT) flag[0] = 1;
                    TL(rb, tid, k) = bin;
                    TL(rd, tid, k) = r;
                }
            }
        }
        SYNC();
        fast = flag[0] == 0;
        PHASE_MARK(1);                                         // histogram
    }
    if (cnt == 1) {                              // nothing to sort
        PAR(tid) { if (tid == 0) { skey[0] = TL(rk, tid, 0); ssa[0] = TL(rs, tid, 0); } }
        SYNC();
        fast = true;
    } else if (fast) {
        block_exclusive_scan_bins(KCTX_PASS hist);         // hist[b] = first slot of bin b, hist[TILE_BINS] = cnt
        PHASE_MARK(2);                                         // scan
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    const uint32_t slot = hist[TL(rb, tid, k)] + TL(rd, tid, k);
                    skey[slot] = TL(rk, tid, k);
                    ssa[slot] = TL(rs, tid, k);
                    if (SLOT_ORDER) sbin[slot] = (uint16_t)TL(rb, tid, k);
                    else TL(rd, tid, k) = slot;
                }
            }
        }
        SYNC();
        PHASE_MARK(3);                                         // place by bin
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    const uint32_t slot = SLOT_ORDER ? e : TL(rd, tid, k), bin = SLOT_ORDER ? (uint32_t)sbin[slot] : TL(rb, tid, k);
                    const uint32_t bs = hist[bin], be = hist[bin + 1];
                    const uint64_t key = SLOT_ORDER ? (uint64_t)skey[slot] : (uint64_t)TL(rk, tid, k);
                    uint32_t less = 0;                        // members of my bin that sort before me
                    for (uint32_t j = bs; j < be; ++j) {
                        const uint64_t kj = skey[j];
                        less += kj < key ? 1u : 0u;
                        if (kj == key && j != slot) {                                                               // rare: text
                            const uint64_t sa = SLOT_ORDER ? (uint64_t)ssa[slot] : (uint64_t)TL(rs, tid, k);
                            const uint32_t c = suffix_less_tie_bounded<BITS>(P, n, (uint64_t)ssa[j], sa, tie_from);
                            if (c == 2u) flag[0] = 1;                                                               // a deep tie: not here
                            less += c & 1u;
                        }
                    }
                    // (bs + less < be <= cnt whatever the ties did; RACY_STORE: two members of a tie too deep for this kernel get one
                    // rank and one cell -- the tile has raised its flag by then and nothing reads perm)
                    if (SLOT_ORDER) RACY_STORE_U16(&perm[bs + less], (uint16_t)slot);
                    else TL(rd, tid, k) = bs + less;
                }
            }
        }
        SYNC();
        fast = flag[0] == 0;
        PHASE_MARK(4);                                         // rank inside the bin
        if (fast && !SLOT_ORDER) {
            TILE_SORT_PLACE_FINAL
            PHASE_MARK(5);                                     // place final
        }
    }
#ifdef CAPS_EMUL
    caps_emul_count_tile(fast, known_range);
#endif
    // unfinished: queue the tile for tile_sort_general_kernel (redo[0] = queue length)
    if (!fast) {
        PAR(tid) { if (tid == 0) redo[1 + FETCH_ADD_U32(&redo[0], 1u)] = b; }
        return;
    }
    if (SLOT_ORDER && cnt > 1) {
        // sorted order -> HBM through perm (+ LCPs from neighbouring keys, + boundary records)
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    const uint32_t sl = perm[e];
                    const KT key = skey[sl];
                    const idx_t sa = ssa[sl];
                    uint64_t l = 0;
                    if (with_lcp && e) { const uint32_t sp = perm[e - 1]; l = TILE_EMIT_LCP_(skey[sp], ssa[sp], key, sa); }
                    if (direct) {
                        STREAM_STORE(&fin.sa[start + e], sa);
                        STREAM_STORE(&fin.lcp[start + e], (idx_t)l);
                        if (e == 0) { fin.first_key[g] = key; fin.first_sa[g] = sa; }
                        if (e == cnt - 1) { fin.last_key[g] = key; fin.last_sa[g] = sa; }
                    } else {
                        out_key[start + e] = key;
                        out_sa[start + e] = sa;
                        if (with_lcp) out_lcp[start + e] = (idx_t)l;
                    }
                }
            }
        }
    } else {
        TILE_SORT_EMIT
    }
    PHASE_MARK(6);                                             // emit (+ LCPs); the stores themselves drain after the mark
}

// The tile table for the kernels at the head of the queue chain of a sort over slots with outgrown buckets (TILE_SORT_PROLOGUE):
// a copy in which the tiles of every segment longer than slot_cap are empty.
GLOBAL_FN LAUNCH_BOUNDS(256) hot_tiles_kernel(KCTX SegDesc sd, uint32_t slot_cap, TileInfo* __restrict__ hot)
{
    PAR(tid) {
        const uint32_t nt = sd.tile_off[sd.G];
        const uint32_t i = K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < nt) {
            TileInfo t = sd.tile_rec[i];
            if (t.s1 - t.s0 > (uint64_t)slot_cap) { t.s1 = t.s0; t.tl = 0; }
            hot[i] = t;
        }
    }
}

// Every tile straight into tile_sort_kernel's queue of unfinished tiles (skewed keys, pipeline.h SortOpts::skewed_keys:
// the linear bin map would crowd nearly every tile -- 97 % of a genome-like text's -- and its attempt is then pure cost).
GLOBAL_FN LAUNCH_BOUNDS(256) queue_all_tiles_kernel(KCTX SegDesc sd, uint32_t* __restrict__ redo)
{
    PAR(tid) {
        const uint32_t nt = sd.tile_off[sd.G];
        const uint32_t i = K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < nt) redo[1 + i] = i;
        if (i == 0) redo[0] = nt;
    }
}

// ---- tile_sort_eq_kernel: second chance for the tiles tile_sort_kernel could not finish --------
// Keys that are far from uniform over the tile's range (skewed k-mer composition: every real genome)
// crowd a few bins of the linear map.  Here the map is equalised with the tile's own data, iteratively:
// every element holds a 24-bit POSITION (bin.fraction, first from the linear map); the histogram of the
// positions' bins is a piecewise-linear CDF, and the new position is the estimated rank it gives:
// (count before my bin + fraction x count of my bin) * TILE_BINS / cnt.  After EQ_ROUNDS rounds the bin of
// the position is the counting sort's bin -- monotone in the key, so the exact in-bin ranking of tile_sort_kernel
// finish the job unchanged.  Takes the tiles from tile_sort_kernel's queue (redo[0] = length), with a fixed grid or one
// workgroup per entry (PERSIST below); a tile that still has a bin above TILE_BIN_LIMIT (clusters of suffixes that share more
// chars than the coarse bins resolve), or a deep tie, goes on to tile_sort_general_kernel (redo2).
// A separate kernel rather than a branch of tile_sort_kernel: that one runs at the register budget.
#ifndef CAPS_EQ_ROUNDS
#define CAPS_EQ_ROUNDS 2          /* measured on skewed-Markov tiles (simulation): largest bin, median 243 (linear) -> 50 / 26 / 13 after 1 / 2 / 3 rounds */
#endif
constexpr uint32_t EQ_ROUNDS = CAPS_EQ_ROUNDS;
#ifndef CAPS_TIE_G1
#define CAPS_TIE_G1 8
#endif
#ifndef CAPS_TIE_LIST_MAX
#define CAPS_TIE_LIST_MAX 2       /* g3 / g3r / g2r at 1, 2, 3, 6, 12: 97.9 98.2 98.4 99.9 100.1 / 150.7 150.5 151.7 153.0 - / 17.8 17.9 17.5 18.5 18.4 ms (beyond: phase B) */
#endif
constexpr uint32_t TIE_G1 = CAPS_TIE_G1;          // lanes (= windows) per listed tie in the first round,
constexpr uint32_t TIE_G2 = 64;                   //   per tie that is still equal in the second
constexpr uint32_t TIE_LIST_CAP = TILE_E / 8;     // list entries per tile (pairs from the bottom, elements' own entries from the top)
constexpr uint32_t TIE_LIST_MAX = CAPS_TIE_LIST_MAX;   // an element lists up to this many equal keys (more: it scans them itself)
constexpr uint32_t TIE_DEEP_CAP = 64;             // entries that may go to the second round
constexpr uint32_t TIE_VDEEP_CAP = 32;            //   ... and on to the third (any depth: long exact duplicates)
constexpr uint32_t TIE_BIG_CAP = 1024;            // elements with more than TIE_LIST_MAX equal keys a tile may hold
constexpr uint32_t TIE_BIG_LANES = 64;            //   lanes that share the equal keys of one of them (fewer when there are many)
constexpr uint32_t EQ_FRAC_BITS = 13;             // position inside a bin; a position is bin * 2^13 + fraction < 2^24 at 2048 bins
static_assert((uint64_t)TILE_BINS_ << EQ_FRAC_BITS <= (1u << 24), "positions fit 24 bits");

// Position of the key on the tile's linear map: (bin of bucket_of(tb, .)) << EQ_FRAC_BITS plus that many more bits
DEV_INLINE uint32_t eq_pos(const BucketParams& bp, uint64_t key)
{
    const uint32_t top = (bp.B << EQ_FRAC_BITS) - 1u;
    if (key <= bp.kmin) return 0;
    const uint64_t d = key - bp.kmin;
    if (d >= bp.range) return top;
    const uint32_t d32 = (uint32_t)((d << bp.shift) >> 32);
    const uint64_t prod = (uint64_t)d32 * bp.m;
    const uint32_t sh = 32u + bp.post;                                   // bucket = prod >> sh
    const uint32_t x = (uint32_t)(sh >= EQ_FRAC_BITS ? prod >> (sh - EQ_FRAC_BITS) : prod << (EQ_FRAC_BITS - sh));
    return x < top ? x : top;
}

// the lcp noted for slot e in the table of the few lcps beyond 15 bits (0: not there -- the caller derives it from the text)
DEV_INLINE uint32_t eq_big_lcp(const uint16_t* vslot, const uint32_t* vlcp, uint32_t cnt, uint32_t e)
{
    uint32_t l = 0;
    for (uint32_t q = 0; q < cnt && q < TIE_VDEEP_CAP; ++q)
        if (vslot[q] == e) l = vlcp[q];
    return l;
}

// lcp of two neighbours of the sorted tile that no tie has settled.  TEXT = false (the plain build): their keys differ --
// every pair of equal keys has its lcp from the tie phases, and those are never 0 -- so the keys tell; the comparator through
// the text, unrolled four times in the emit phase, cost that phase registers (scratch) for a path no tile takes.
template <int BITS, bool TEXT, typename idx_t>
DEV_INLINE uint64_t eq_emit_lcp(const uint32_t* __restrict__ P, uint64_t n, uint64_t ka, idx_t a, uint64_t kb, idx_t b, uint32_t* lost, bool test_drop)
{
    if (TEXT) return pair_lcp<BITS, false>(P, n, ka, (uint64_t)a, kb, (uint64_t)b);
    const uint64_t x = ka ^ kb;
    if (x == 0) {
        // The invariant the plain build rests on: equal keys always come with their lcp from the tie phases.  Should a note ever be
        // lost, the capped key length would be a WRONG lcp: the tile is handed to the comparison sort instead (*lost, read by the
        // kernel behind its last barrier), which sorts it again from its input.  The emulation aborts (unless the test switch that
        // drops notes on purpose is on): there the invariant is a test.
#ifdef CAPS_EMUL
        if (!test_drop) { std::fprintf(stderr, "eq_emit_lcp: neighbours %llu, %llu with equal keys and no lcp from the tie phases\n", (unsigned long long)a, (unsigned long long)b); std::abort(); }
#endif
        *lost = 1;                                                   // (benign race: every writer stores 1)
    }
    return lcp_capped<idx_t>(x ? (uint32_t)caps_clz64(x) / BITS : TextTraits<BITS>::KCH, n, a, b);
}

#ifndef CAPS_EQ_WAVES
#define CAPS_EQ_WAVES TILE_WAVES_PER_SIMD
#endif
#undef TILE_EMIT_LCP_
#define TILE_EMIT_LCP_(ka, a, kb, b) eq_emit_lcp<BITS, VDEEP>(P, n, ka, a, kb, b, &flag[1], TEST_DROP)
#undef TILE_SRC_KEY
#undef TILE_SRC_SA
#define TILE_SRC_KEY src_key
#define TILE_SRC_SA src_sa
#ifndef CAPS_EQ_FULL_SYNC          /* measurement: -DCAPS_EQ_FULL_SYNC keeps __syncthreads() in this kernel */
#undef TILE_SYNC
#define TILE_SYNC() SYNC_LDS()
#endif
// this kernel and tile_sort_general_kernel: every PAR region takes its thread index afresh (kernel_lang.h PAR_FRESH; measured
// at 3e9 genome-like: 57.3 -> 54.1 ms here, no register left in scratch)
#if !defined(CAPS_EMUL) && !defined(CAPS_PAR_PLAIN_ONLY)
#pragma push_macro("PAR")
#undef PAR
#define PAR(tid) PAR_FRESH(tid)
#define CAPS_PAR_SWITCHED
#endif
// VDEEP: the build with the third tie stage (any depth).  A build of its own, behind the plain one in the queue chain: the
// plain kernel runs at the register budget, and the stage's code in it cost every tile of a genome-like text 14 % (76 -> 87 ms
// at 3e9) for the sake of the few that hold an exact long duplicate.
// PERSIST = false: one queue entry per workgroup (the grid covers the queue: the caller knows its length, or a bound).  The
// loop over the queue lets the compiler hoist per-thread addresses and constants out of it, and what it hoists it must hold
// across all ~20 phases of a tile: at the 64-register budget that put 41 registers in scratch memory, reloaded 58 times per
// tile (and 13 GB of scratch stores reached HBM at 3e9); without the loop: 11, reloaded 5 times, 19 % fewer instructions.
template <typename idx_t, int BITS, bool FROM_TEXT, bool VDEEP = false, bool PERSIST = true>
GLOBAL_FN LAUNCH_BOUNDS2(TILE_NT, CAPS_EQ_WAVES) tile_sort_eq_kernel(KCTX SegDesc sd, const uint32_t* __restrict__ P, uint64_t n,
                                                  uint64_t text_base, uint32_t lcp_mode, uint32_t slot_cap, const uint64_t* in_key,
                                                  const idx_t* in_sa, uint64_t* out_key, idx_t* out_sa, idx_t* out_lcp,
                                                  FinalOut<idx_t> fin, const BucketParams* __restrict__ seg_map,
                                                  const uint32_t* __restrict__ redo, uint32_t* __restrict__ redo2, uint32_t keys_from_text,
                                                  uint32_t* __restrict__ redo_deep)
{
    // redo_deep (the plain build): the queue of the tiles whose ONLY obstacle was a tie deeper than the first two rounds see -- the
    // VDEEP build's; everything else this kernel cannot finish goes to redo2, the comparison sort's
    // keys_from_text != 0: the slots hold 32-bit keys (tile_sort_kernel<..., uint32_t> ran first); this kernel works on the
    // 64-bit keys, cut from the text at the elements' positions (seg_map is then null: the tile's own key range)
    constexpr bool TILE_RUNS = false;
    const bool TILE_KEYS_FROM_TEXT = (keys_from_text & 1u) != 0;
    // bit 1 of keys_from_text (tests only, CAPS_SA_TEST_DROP_NOTE): the lcp notes of tied pairs are dropped on purpose -- the emit
    // phase must then notice equal keys without a note and hand the tile to the comparison sort (eq_emit_lcp)
    const bool TEST_DROP = (keys_from_text & 2u) != 0;
    const uint32_t TILE_KEY_SHIFT = 0;
    (void)TILE_RUNS;
    (void)TILE_KEY_SHIFT;
    SHARED_ARRAY(uint64_t, skey, TILE_E);
    SHARED_ARRAY(idx_t, ssa, TILE_E);
    SHARED_ARRAY(uint32_t, hist, TILE_BINS + 1);
    // 32-bit indices: 80 KB of LDS with these two (two workgroups per CU); 64-bit indices have no room for sbin, and a shorter list
    constexpr bool SLOT_ORDER = sizeof(idx_t) == 4;
    constexpr uint32_t BIG_CAP = sizeof(idx_t) == 4 ? TIE_BIG_CAP : TIE_BIG_CAP / 4;
    SHARED_ARRAY(uint16_t, sbin, SLOT_ORDER ? TILE_E : 1);   // the bin of the element in slot e (after the placement by bin)
    // PERM (where sbin exists): no second placement.  Once an element knows its final position d it only notes perm[d] = its slot (in
    // sbin's memory, dead by then) and leaves its lcp note in tinfo[slot]; the emit phase reads keys and indices THROUGH perm.  The
    // pick-up of (key, index) into registers, the barrier, their second trip through LDS and the registers they held are gone.
    constexpr bool PERM = SLOT_ORDER;
    uint16_t* perm = sbin;
    SHARED_ARRAY(uint64_t, kmm, 2);
    SHARED_ARRAY(uint32_t, flag, 2);         // [0] the tile is the comparison sort's; [1] it only needs the third tie stage; during the
                                             //   emit phase [1]: neighbours with equal keys and no lcp note (eq_emit_lcp)
    TL_DECL(uint64_t, rk, TILE_EPT);
    TL_DECL(idx_t, rs, TILE_EPT);
    TL_DECL(uint32_t, rd, TILE_EPT);
    TL_DECL(uint32_t, rb, TILE_EPT);
    TL_DECL(uint32_t, rt, TILE_EPT);         // rank phase: rank inside the bin | first own entry << 8 | ties << 28
    TL_DECL(uint32_t, rl, TILE_EPT);         // lcp with the predecessor where a tie settled it
    TL_DECL(uint64_t, twa, 1);               // tie rounds: the lane's pair of windows,
    TL_DECL(uint64_t, twb, 1);
    TL_DECL(uint32_t, tpi, 1);               //   its entry (index | own-entry flag << 31 | virtual index << 16; ~0: none)
    SHARED_ARRAY(uint32_t, plist, TIE_LIST_CAP);      // ties: first differing window << 24 | slot << 12 | the other's slot;
                                                      //   an own entry once settled: 0x80 | the other sorts first, << 24 | lcp
    SHARED_ARRAY(uint16_t, deep, TIE_DEEP_CAP);       // entries (virtual index) still equal after the first round
    SHARED_ARRAY(uint16_t, vdeep, TIE_VDEEP_CAP);     //   ... and after the second: settled one by one, the whole workgroup on each
    SHARED_ARRAY(uint32_t, pcnt, 8);                  // pairs listed, own entries listed, entries on `deep`, on `vdeep`, [4] first hit, [5] big lcps,
                                                      //   [6] elements on `big`
    SHARED_ARRAY(uint32_t, big, BIG_CAP);             // elements with more than TIE_LIST_MAX equal keys: slot | first of bin << 12 | bin size - 1 << 24
    SHARED_ARRAY(uint32_t, bmore, BIG_CAP);           //   how many of its equal keys sort before it,
    SHARED_ARRAY(uint32_t, bbest, BIG_CAP);           //   the largest lcp with one of those
    SHARED_ARRAY(uint32_t, vlcp, TIE_VDEEP_CAP);      // third stage: the lcp of pair q (too large for the 15 bits of its members' notes);
                                                      //   after the final placement: the lcps >= 0x7FFF, with ...
    SHARED_ARRAY(uint16_t, vslot, TIE_VDEEP_CAP);     //   ... the slots they belong to
    SHARED_ARRAY(uint32_t, vlcp2, PERM ? TIE_VDEEP_CAP : 1);   // PERM: the lcps >= 0x7FFF live here (vlcp is still being read when they are noted)
    static_assert((TILE_BINS + 1) * sizeof(uint32_t) >= TILE_E * sizeof(uint16_t) && TILE_E <= (1u << 12) && EQ_BIN_LIMIT <= 128 &&
                  TILE_NT % TIE_G1 == 0 && TILE_NT % TIE_G2 == 0 && TIE_G1 + TIE_G2 < 0x80u && TIE_LIST_CAP <= (1u << 12) &&
                  TIE_BIG_CAP <= (1u << 12) && (TILE_NT & (TILE_NT - 1u)) == 0 &&
                  TextTraits<BITS>::KCH * (1u + TIE_G1 + TIE_G2) < 0x7FFFu,
                  "tinfo fits hist; slots and entries fit 12 bits; ranks inside a bin fit 8 bits; lcps of the first two rounds fit 15 bits");
    // per slot: "the other member of my pair sorts before me" << 15 | their lcp (0x7FFF: see vlcp); after the final placement: the
    // lcp with the predecessor where a tie settled it (0x7FFF: see vslot / vlcp)
    uint16_t* tinfo = reinterpret_cast<uint16_t*>(hist);
    // redo == nullptr (PERSIST = false only): every tile is this kernel's and the grid is the tile count -- no queue to read (one
    // dependent round trip to memory less at the head of every tile)
    const uint32_t n_redo = redo ? redo[0] : K_GRID_DIM;
    PAR_TID_DECL;
    // (fetching the queue entry and the record of the NEXT tile while this one is sorted was tried: no gain, and the record's
    // registers, held across the whole tile, went to scratch)
    uint32_t qi = K_BLOCK_IDX;
    if (qi >= n_redo) return;
    do {
    PAR_TID_RESET;
    const uint32_t b = redo ? redo[1 + qi] : qi;
    if (!EQ_OK(b < sd.tile_off[sd.G], 0)) return;
    const TileInfo t = tile_info(sd, b);
    const uint32_t g = t.g;
    const uint64_t start = t.s0 + (uint64_t)t.tl * TILE_E;
    if (!EQ_OK(g < sd.G && t.s0 <= start && start < t.s1, 1)) return;
    const uint32_t cnt = (uint32_t)(t.s1 - start < TILE_E ? t.s1 - start : TILE_E);
    const bool with_lcp = lcp_mode != 0 && (t.s1 - t.s0) <= TILE_E;
    const bool direct = with_lcp && fin.sa != nullptr;
    // a bucket that outgrew its slot (TILE_SORT_PROLOGUE): the plain build sees its tiles empty and passes them on (`fast` below is
    // false for them, no flag is up: redo_deep); the VDEEP build reads them where they were put together -- at their place in the
    // output arrays -- and sorts them in place
    const bool in_slot = !VDEEP || !TILE_OUTGROWN;
    const uint64_t in0 = slot_cap != 0 && in_slot ? (uint64_t)g * slot_cap : start;
    const uint64_t* src_key = tile_src(in_key, out_key, !in_slot);
    const idx_t* src_sa = in_slot ? in_sa : out_sa;
    PHASE_T0();
    TILE_SORT_LOAD
#if defined(CAPS_EQ_CHECK) && !defined(CAPS_EMUL)
    PAR(tid) { for (uint32_t k = 0; k < TILE_EPT; ++k) if (tid + k * TILE_NT < cnt) EQ_CHK((uint64_t)TL(rs, tid, k) < n || n == 0, 2); }
#endif
    TILE_SORT_RANGE
    PHASE_MARK(8);                                             // load
    bool fast = cnt > EQ_BIN_LIMIT && tb.range > 0 && tb.B == TILE_BINS;
    if (fast) {
        PAR_FRESH_SET(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) TL(rb, tid, k) = eq_pos(tb, TL(rk, tid, k));
            }
        }
        // position -> estimated rank, EQ_ROUNDS times: histogram of the positions' bins, its prefix sums are a
        // piecewise-linear CDF, the new position is the CDF value scaled back to [0, TILE_BINS)
        // estimated rank -> position: E * TILE_BINS / cnt.  In single precision: only monotony in E counts (equal or larger E
        // never gives a smaller position: conversion, product and truncation are all monotone), and E < 2^27, so the 64-bit
        // product this used to be cost three times the instructions for bits the bins never see.
        const float Kf = (float)TILE_BINS / (float)cnt;
        for (uint32_t round = 0; round < EQ_ROUNDS; ++round) {
            if (round) {
                PAR_SAME_G(1a, tid) { for (uint32_t i = tid; i <= TILE_BINS; i += K_BLOCK_DIM) hist[i] = 0; }
                TILE_SYNC();
            }
            PAR_SAME_G(1b, tid) {
                UNROLL
                for (uint32_t k = 0; k < TILE_EPT; ++k) {
                    const uint32_t e = tid + k * TILE_NT;
                    if (e < cnt && EQ_OK((TL(rb, tid, k) >> EQ_FRAC_BITS) < TILE_BINS, 3)) FETCH_ADD_U32(&hist[TL(rb, tid, k) >> EQ_FRAC_BITS], 1u);
                }
            }
            TILE_SYNC();
            block_exclusive_scan_bins(KCTX_PASS hist);
            PAR_SAME_G(1c, tid) {
                UNROLL
                for (uint32_t k = 0; k < TILE_EPT; ++k) {
                    const uint32_t e = tid + k * TILE_NT;
                    if (e < cnt) {
                        const uint32_t x = TL(rb, tid, k);
                        const uint32_t bn = x >> EQ_FRAC_BITS, fr = x & ((1u << EQ_FRAC_BITS) - 1u);
                        EQ_CHK(bn < TILE_BINS, 4);
                        const uint32_t before = hist[bn], here = hist[bn + 1] - before;
                        const uint32_t E = (before << EQ_FRAC_BITS) + here * fr;                          // < cnt * 2^13 <= 2^25 (+ 2^25)
                        const uint32_t y = (uint32_t)((float)E * Kf);                                     // ~ E * TILE_BINS / cnt
                        TL(rb, tid, k) = y < (TILE_BINS << EQ_FRAC_BITS) ? y : (TILE_BINS << EQ_FRAC_BITS) - 1u;
                    }
                }
            }
            TILE_SYNC();
        }
        PHASE_MARK(9);                                         // equalisation rounds
        PAR_SAME_G(1d, tid) {
            for (uint32_t i = tid; i <= TILE_BINS; i += K_BLOCK_DIM) hist[i] = 0;
            if (tid < 8) pcnt[tid] = 0;
            if (tid == 0) flag[1] = 0;
            for (uint32_t i = tid; i < BIG_CAP; i += K_BLOCK_DIM) { bmore[i] = 0; bbest[i] = 0; }
        }
        TILE_SYNC();
        PAR_SAME_G(1e, tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    const uint32_t bin = TL(rb, tid, k) >> EQ_FRAC_BITS;
                    EQ_CHK(bin < TILE_BINS, 5);
                    const uint32_t r = FETCH_ADD_U32(&hist[bin], 1u);
                    if (r >= EQ_BIN_LIMIT) flag[0] = 1;
                    TL(rb, tid, k) = bin;
                    TL(rd, tid, k) = r;
                }
            }
        }
        TILE_SYNC();
        fast = flag[0] == 0;
        PHASE_MARK(10);                                        // final histogram
    }
    if (fast) {                                                // from here on: as in tile_sort_kernel
        block_exclusive_scan_bins(KCTX_PASS hist);
        PHASE_MARK(11);                                        // scan
        PAR_SAME_G(1f, tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    const uint32_t slot = hist[TL(rb, tid, k)] + TL(rd, tid, k);
                    if (!EQ_OK(slot < cnt, 6)) continue;
                    skey[slot] = TL(rk, tid, k);
                    ssa[slot] = TL(rs, tid, k);
                    if (SLOT_ORDER) sbin[slot] = (uint16_t)TL(rb, tid, k);
                    else TL(rd, tid, k) = slot;
                }
            }
        }
        TILE_SYNC();
        PHASE_MARK(12);                                        // place by bin
        // ---- exact rank inside the bin.  Ties (equal keys: 2 % of a genome-like text's neighbours, repeats with a
        // mutation every ~100 chars) need the text, several windows deep, every window a dependent HBM read of ~1 us.  Inside
        // the ranking loop such a read stalls the whole wave once per tie and window, one after the other: measured with the
        // phase clock (tools/phase_clock.py), 53 % of this kernel's time, and 24 % more in the emit phase, which read the same
        // text again for the lcps.  So:
        //  R1  rank by keys only and note the ties.  A PAIR of equal keys (91 % of the tied elements of a genome-like text) is
        //      listed once, by its member with the higher slot; an element with 2 .. TIE_LIST_MAX equal keys lists (itself, the
        //      other) for each of them -- its own entries, from the top of the same list;
        //  T   the workgroup settles the listed entries together, several lanes per entry, one window each, all loads of a
        //      round in flight at once: TIE_G1 windows deep for ONE memory latency.  The lane with the first difference (an
        //      LDS atomic min on the entry) derives order and lcp: for a pair it leaves them for both members in tinfo, for
        //      an own entry in the entry itself.  What is still equal (one tie in eight on such a text) goes on a short list
        //      and gets TIE_G2 lanes per entry in a second round; still equal after that (TIE_G1 + TIE_G2 windows), or more
        //      ties than the lists hold: the tile is the comparison sort's;
        //  B   an element with more than TIE_LIST_MAX equal keys: its own lanes, one comparison each (below);
        //  R3  the elements pick their outcomes up.
        // The lcp of an element with its predecessor in the final order is the largest lcp with a smaller member of its tie
        // group; it travels with the element to its final slot (slcp) and the emit phase uses it instead of the text.
        // tinfo / slcp: u16 per slot, in the memory of hist (free once R1 has read the bin bounds).
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    // from here on a thread works on the elements in SLOTS tid + k * TILE_NT (the slot's bin: sbin), not on the
                    // ones it loaded: the lanes of a wave then hold neighbouring slots -- mostly one bin, so they scan the same
                    // keys (one LDS read serves them all) for the same number of steps, where the lanes of a wave used to sit
                    // in 64 bins of 64 sizes.  The elements live in skey / ssa only until the final placement.
                    // (64-bit indices: no room in LDS for sbin; every thread stays with the elements it loaded)
                    const uint32_t slot = SLOT_ORDER ? e : TL(rd, tid, k), bin = SLOT_ORDER ? (uint32_t)sbin[slot] : TL(rb, tid, k);
                    TL(rd, tid, k) = slot;
                    EQ_CHK(bin < TILE_BINS && slot < cnt, 7);
                    const uint32_t bs = hist[bin], be = hist[bin + 1];
                    EQ_CHK(bs <= slot && slot < be && be <= cnt && be - bs <= EQ_BIN_LIMIT, 8);
                    const uint64_t key = skey[slot];
                    uint32_t less = 0, ties = 0, tj = 0;
                    for (uint32_t j = bs; j < be; ++j) {
                        const uint64_t kj = skey[j];
                        less += kj < key ? 1u : 0u;
                        if (kj == key && j != slot) { ++ties; tj = j; }
                    }
                    uint32_t own = 0;                                      // my first entry, counted from the top of the list
                    if (ties == 1u && slot > tj) {
                        const uint32_t pi = FETCH_ADD_U32(&pcnt[0], 1u);
                        // (RACY_STORE: pairs from the bottom and own entries from the top run into each other only when np + nm >
                        // TIE_LIST_CAP, and then the tile fails below before anything reads the list)
                        if (pi < TIE_LIST_CAP) RACY_STORE_U32(&plist[pi], 0xFF000000u | (slot << 12) | tj);
                    } else if (ties >= 2u && ties <= TIE_LIST_MAX) {
                        own = FETCH_ADD_U32(&pcnt[1], ties);
                        if (own + ties <= TIE_LIST_CAP) {
                            uint32_t q = own;
                            for (uint32_t j = bs; j < be; ++j)
                                if (skey[j] == key && j != slot) RACY_STORE_U32(&plist[TIE_LIST_CAP - 1u - q++], 0xFF000000u | (slot << 12) | j);
                        } else {
                            own = 0;                                       // (the tile fails below: the list is full)
                        }
                    } else if (ties > TIE_LIST_MAX) {
                        own = FETCH_ADD_U32(&pcnt[6], 1u);                 // here: my place on `big`
                        if (own < BIG_CAP) big[own] = slot | (bs << 12) | ((be - bs - 1u) << 24);
                        else { own = 0; flag[0] = 1; }                     // (more such elements than the list holds: not here)
                    }
                    TL(rb, tid, k) = bs | (be << 16);                      // the bin id is not needed any more
                    TL(rt, tid, k) = less | (own << 8) | (ties > 15u ? 15u << 28 : ties << 28);   // less <= 127, own < 2^12
                }
            }
        }
        TILE_SYNC();                                                // hist is free: tinfo / slcp from here on
        PHASE_MARK(16);                                        // R1: rank by keys
        // B: the elements with more than TIE_LIST_MAX equal keys (0.1 % of a genome-like text's suffixes, in clusters: every
        // tenth tile holds some).  Each gets up to TIE_BIG_LANES lanes that share its bin: every lane compares it with its share
        // of the equal keys through the text, the comparisons of an element in flight at once -- in the pick-up phase, where every such element
        // went through its equal keys one after the other (c - 1 dependent trips to the text for each member of a cluster of
        // c), the whole workgroup waited for the longest chain, and the comparator's code, unrolled there four times, put
        // that phase on scratch memory.
        {
            const uint32_t nb = pcnt[6] < BIG_CAP ? pcnt[6] : BIG_CAP;                     // block-uniform
            uint32_t lg = 0;                                                               // lanes per element: 2^lg <= TIE_BIG_LANES,
            while ((2u << lg) <= TIE_BIG_LANES && (uint64_t)nb * (2u << lg) <= TILE_NT) ++lg;   //   as many as one round has room for
            for (uint32_t base = 0; base < nb; base += TILE_NT >> lg) {
                PAR(tid) {
                    const uint32_t i = base + (tid >> lg);
                    if (i < nb) {
                        const uint32_t ent = big[i], slot = ent & 0xFFFu, bs = (ent >> 12) & 0xFFFu, be = bs + (ent >> 24) + 1u;
                        EQ_CHK(slot < cnt && be <= cnt, 9);
                        const uint64_t key = skey[slot];
                        const uint64_t sa = (uint64_t)ssa[slot];
                        uint32_t more = 0, best = 0;
                        for (uint32_t j = bs + (tid & ((1u << lg) - 1u)); j < be; j += 1u << lg) {
                            if (skey[j] == key && j != slot) {
                                uint32_t l;
                                const uint32_t c = tie_order_lcp_bounded<BITS>(P, n, (uint64_t)ssa[j], sa, l);   // 1: j sorts before me
                                if (c == 2u) flag[0] = 1;
                                if (c == 1u) { ++more; best = l > best ? l : best; }
                            }
                        }
                        if (more) { FETCH_ADD_U32(&bmore[i], more); ATOMIC_MAX_U32(&bbest[i], best); }
                    }
                }
            }
        }
        PHASE_MARK(18);                                        // B: elements with many equal keys
        {
            const uint32_t np = pcnt[0], nm = pcnt[1];                                     // block-uniform
            constexpr uint32_t KCH_ = TextTraits<BITS>::KCH;
            if (np + nm > TIE_LIST_CAP) {
                PAR(tid) { if (tid == 0) flag[0] = 1; }                                    // more ties than the list holds: not here
            } else if (np + nm) {
                for (uint32_t round = 0; round < 2; ++round) {
                    // round 0: every entry, TIE_G1 lanes each; round 1: the entries on the short list, TIE_G2 lanes each
                    const uint32_t G = round ? TIE_G2 : TIE_G1, W0 = round ? TIE_G1 : 0u;
                    const uint32_t n_ent = round ? (pcnt[2] < TIE_DEEP_CAP ? pcnt[2] : TIE_DEEP_CAP) : np + nm;
                    if (round && pcnt[2] > TIE_DEEP_CAP) { PAR(tid) { if (tid == 0) flag[0] = 1; } }
                    for (uint32_t base = 0; base < n_ent; base += TILE_NT / G) {
                        PAR_FRESH_SET(tid) {
                            const uint32_t vi = base + tid / G, W = W0 + tid % G;
                            TL(twa, tid, 0) = 0;
                            TL(twb, tid, 0) = 0;
                            TL(tpi, tid, 0) = ~0u;
                            if (vi < n_ent) {
                                const uint32_t vj = round ? deep[vi] : vi;
                                const uint32_t pi = vj < np ? vj : TIE_LIST_CAP - 1u - (vj - np);
                                const uint32_t ent = RACY_LOAD_U32(&plist[pi]);     // (the lanes of this entry may have touched it: below)
                                TL(tpi, tid, 0) = pi | (vj < np ? 0u : 0x80000000u) | (vj << 16);
                                // open: untouched (0xFF) or touched by another lane of THIS round (its window number) -- not
                                // "untouched" alone: the lanes of an entry would then depend on each other's timing (they do run
                                // in lockstep, one wave holds them all, but nothing here should need that; the reversed-order
                                // emulation, tests/test_emul_pipeline.py, found it)
                                const uint32_t tb = ent >> 24;
                                if (tb == 0xFFu || (tb >= W0 && tb < W0 + G)) {
                                    EQ_CHK((ent & 0xFFFu) < cnt && ((ent >> 12) & 0xFFFu) < cnt, 10);
                                    const uint64_t a = (uint64_t)ssa[ent & 0xFFFu], b2 = (uint64_t)ssa[(ent >> 12) & 0xFFFu];
                                    EQ_CHK(a < n && b2 < n, 11);
                                    const uint64_t maxlen = a < n && b2 < n ? n - (a > b2 ? a : b2) : 0;   // corrupt index: settle at once
                                    const uint64_t l = (uint64_t)KCH_ * (1u + W);
                                    bool hit = l >= maxlen;               // the shorter suffix ends before this window
                                    if (!hit) {
                                        const uint64_t wa = window64<BITS>(P, a + l), wb = window64<BITS>(P, b2 + l);
                                        TL(twa, tid, 0) = wa;
                                        TL(twb, tid, 0) = wb;
                                        hit = wa != wb;
                                    }
                                    if (hit) ATOMIC_MIN_U32(&plist[pi], (W << 24) | (ent & 0xFFFFFFu));
                                }
                            }
                        }
                        TILE_SYNC();
                        PAR_SAME_G(2, tid) {
                            const uint32_t W = W0 + tid % G, tp = TL(tpi, tid, 0);
                            if (tp != ~0u) {
                                const uint32_t pi = tp & 0xFFFFu, vj = (tp >> 16) & 0x7FFFu;
                                // (racing with the winner's store below by design: a settled own entry reads 0x80 / 0x81 in its top byte,
                                // which is neither a window number nor 0xFF, and the lane that reads it does nothing)
                                const uint32_t ent = RACY_LOAD_U32(&plist[pi]), lo = ent & 0xFFFu, hi = (ent >> 12) & 0xFFFu;
                                if ((ent >> 24) == W) {                   // mine is the first window that differs (or ends)
                                    const uint64_t a = (uint64_t)ssa[lo], b2 = (uint64_t)ssa[hi];
                                    const uint64_t maxlen = a < n && b2 < n ? n - (a > b2 ? a : b2) : 0;
                                    const uint64_t l = (uint64_t)KCH_ * (1u + W);
                                    const uint64_t wa = TL(twa, tid, 0), wb = TL(twb, tid, 0);
                                    uint64_t d = maxlen;
                                    bool lo_first = a > b2;               // one is a prefix of the other: the shorter first
                                    if (l < maxlen) {
                                        d = l + (uint32_t)caps_clz64(wa ^ wb) / BITS;
                                        d = d < maxlen ? d : maxlen;
                                        lo_first = wa < wb;
                                    }
                                    if (tp & 0x80000000u) {               // an element's own entry: "the other sorts before you" + lcp
                                        RACY_STORE_U32(&plist[pi], (lo_first ? 0x81000000u : 0x80000000u) | (uint32_t)d);
                                    } else {                              // a pair: the same for both members
                                        tinfo[hi] = (uint16_t)(lo_first ? 0x8000u | (uint32_t)d : 0u);
                                        tinfo[lo] = (uint16_t)(lo_first ? 0u : 0x8000u | (uint32_t)d);
                                    }
                                } else if ((ent >> 24) == 0xFFu && tid % G == 0) {
                                    if (round) {                          // deeper than TIE_G1 + TIE_G2 windows: the third stage
                                        const uint32_t q = VDEEP ? FETCH_ADD_U32(&pcnt[3], 1u) : TIE_VDEEP_CAP;
                                        if (q < TIE_VDEEP_CAP) vdeep[q] = (uint16_t)vj;
                                        else flag[VDEEP ? 0 : 1] = 1;     // (the plain build: the tile goes on to the VDEEP one)
                                    } else {
                                        const uint32_t q = FETCH_ADD_U32(&pcnt[2], 1u);
                                        if (q < TIE_DEEP_CAP) deep[q] = (uint16_t)vj;
                                    }
                                }
                            }
                        }
                        TILE_SYNC();
                    }
                    if (round || pcnt[2] == 0) break;
                }
                // ---- third stage: what is still equal after TIE_G1 + TIE_G2 windows (2,300 bases: an exact duplicate of a gene, a
                // segmental duplication; ONE exact 50-kb duplicate in a 256 Mi text left a pair like that in a third of all tiles,
                // each of which then went to the comparison sort: 43 of 58 ms).  One entry at a time, every thread of the workgroup
                // a window of its own, TILE_NT windows per round trip to the text, until the first difference (or the end of the
                // shorter suffix) is in sight: any depth.
                const uint32_t nv = !VDEEP ? 0u : pcnt[3] < TIE_VDEEP_CAP ? pcnt[3] : TIE_VDEEP_CAP;   // block-uniform (more: flag is set)
                for (uint32_t q = 0; q < nv && flag[0] == 0; ++q) {
                    const uint32_t vj = vdeep[q];
                    const uint32_t pi = vj < np ? vj : TIE_LIST_CAP - 1u - (vj - np);
                    const uint32_t ent = plist[pi], lo = ent & 0xFFFu, hi = (ent >> 12) & 0xFFFu;
                    const uint64_t a = (uint64_t)ssa[lo], b2 = (uint64_t)ssa[hi];
                    const uint64_t maxlen = a < n && b2 < n ? n - (a > b2 ? a : b2) : 0;
                    bool settled = false;
                    for (uint64_t w0 = TIE_G1 + TIE_G2; !settled; w0 += TILE_NT) {
                        PAR(tid) {
                            if (tid == 0) pcnt[4] = ~0u;
                        }
                        TILE_SYNC();
                        PAR(tid) {
                            const uint64_t l = (uint64_t)KCH_ * (1u + w0 + tid);
                            TL(twa, tid, 0) = 0;
                            TL(twb, tid, 0) = 0;
                            bool hit = l >= maxlen;
                            if (!hit) {
                                const uint64_t wa = window64<BITS>(P, a + l), wb = window64<BITS>(P, b2 + l);
                                TL(twa, tid, 0) = wa;
                                TL(twb, tid, 0) = wb;
                                hit = wa != wb;
                            }
                            if (hit) ATOMIC_MIN_U32(&pcnt[4], tid);
                        }
                        TILE_SYNC();
                        const uint32_t first = pcnt[4];                                        // block-uniform
                        if (first != ~0u) {
                            PAR(tid) {
                                if (tid == first) {
                                    const uint64_t l = (uint64_t)KCH_ * (1u + w0 + tid);
                                    const uint64_t wa = TL(twa, tid, 0), wb = TL(twb, tid, 0);
                                    uint64_t d = maxlen;
                                    bool lo_first = a > b2;
                                    if (l < maxlen) {
                                        d = l + (uint32_t)caps_clz64(wa ^ wb) / BITS;
                                        d = d < maxlen ? d : maxlen;
                                        lo_first = wa < wb;
                                    }
                                    if (d >= (vj < np ? (1ull << 32) : (1ull << 24))) flag[0] = 1;       // (an lcp the notes cannot hold)
                                    else if (vj >= np) plist[pi] = (lo_first ? 0x81000000u : 0x80000000u) | (uint32_t)d;
                                    else {                                // a pair: both members point at vlcp[q]
                                        vlcp[q] = (uint32_t)d;
                                        tinfo[hi] = (uint16_t)(lo_first ? 0xFFFFu : 0u);
                                        tinfo[lo] = (uint16_t)(lo_first ? 0u : 0xFFFFu);
                                    }
                                }
                            }
                            settled = true;
                        }
                        TILE_SYNC();
                    }
                }
            }
        }
        TILE_SYNC();
        PHASE_MARK(17);                                        // T: the listed ties
        fast = flag[0] == 0 && flag[1] == 0;                   // (the tie phases were the last to raise them)
        if (fast) {
        PAR_FRESH_SET(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (!PERM) TL(rl, tid, k) = 0;
                if (e < cnt) {
                    const uint32_t info = TL(rt, tid, k);
                    const uint32_t ties = info >> 28, slot = TL(rd, tid, k), own = (info >> 8) & 0xFFFu;
                    const uint32_t bs = TL(rb, tid, k) & 0xFFFFu;
                    uint32_t more = 0, best = 0;
                    if (ties == 1u) {
                        const uint32_t v = tinfo[slot];
                        if (v & 0x8000u) {
                            more = 1;
                            best = v & 0x7FFFu;
                            if (VDEEP && best == 0x7FFFu) {                // settled in the third stage: its pair's lcp is in vlcp
                                const uint32_t nq = pcnt[3] < TIE_VDEEP_CAP ? pcnt[3] : TIE_VDEEP_CAP;
                                for (uint32_t q = 0; q < nq; ++q) {
                                    const uint32_t vj = vdeep[q];
                                    if (vj < pcnt[0]) { const uint32_t en = plist[vj]; if ((en & 0xFFFu) == slot || ((en >> 12) & 0xFFFu) == slot) best = vlcp[q]; }
                                }
                            }
                            if (TEST_DROP) best = 0;                       // (tests: the note is lost)
                        }
                    } else if (ties >= 2u && ties <= TIE_LIST_MAX) {
                        for (uint32_t q = 0; q < ties; ++q) {
                            const uint32_t r = plist[TIE_LIST_CAP - 1u - (own + q)];
                            if (r & 0x01000000u) { ++more; best = (r & 0xFFFFFFu) > best ? (r & 0xFFFFFFu) : best; }
                        }
                    } else if (ties > TIE_LIST_MAX) {
                        more = bmore[own];                                 // (phase B)
                        best = bbest[own];
                    }
                    const uint32_t d = bs + (((info & 0xFFu) + more) & 0xFFu);
                    EQ_CHK(slot < cnt && d < cnt, 12);
                    if (PERM) {
                        // the lcp with the predecessor, where a tie settled it, stays with the SLOT (the few beyond 15 bits: vslot / vlcp2)
                        uint32_t lc = best;
                        if (VDEEP && lc >= 0x7FFFu) {
                            const uint32_t q = FETCH_ADD_U32(&pcnt[5], 1u);
                            if (q < TIE_VDEEP_CAP) { vslot[q] = (uint16_t)slot; vlcp2[q] = lc; }
                            lc = q < TIE_VDEEP_CAP ? 0x7FFFu : 0u;         // (no room: the emit phase derives it from the text)
                        }
                        tinfo[slot] = (uint16_t)lc;
                        if (d < TILE_E) perm[d] = (uint16_t)slot;          // (always, in a tile that has not failed: d < cnt)
                    } else {
                        TL(rl, tid, k) = best;
                        TL(rk, tid, k) = skey[slot];                       // picked up for the final placement (behind the barrier)
                        TL(rs, tid, k) = ssa[slot];
                        TL(rd, tid, k) = d;
                    }
                }
            }
        }
        TILE_SYNC();
        PHASE_MARK(13);                                        // rank inside the bin
        if (!PERM) {
            PAR_SAME_G(4, tid) {
                UNROLL
                for (uint32_t k = 0; k < TILE_EPT; ++k) {
                    const uint32_t e = tid + k * TILE_NT;
                    if (e < cnt) {
                        const uint32_t d = TL(rd, tid, k);
                        if (!EQ_OK(d < cnt, 13)) continue;
                        skey[d] = TL(rk, tid, k);
                        ssa[d] = TL(rs, tid, k);
                        // = slcp: the lcp with the predecessor, where a tie settled it (the few beyond 15 bits: in vslot / vlcp)
                        uint32_t lc = TL(rl, tid, k);
                        if (VDEEP && lc >= 0x7FFFu) {
                            const uint32_t q = FETCH_ADD_U32(&pcnt[5], 1u);
                            if (q < TIE_VDEEP_CAP) { vslot[q] = (uint16_t)d; vlcp[q] = lc; }
                            lc = q < TIE_VDEEP_CAP ? 0x7FFFu : 0u;         // (no room: the emit phase derives it from the text)
                        }
                        tinfo[d] = (uint16_t)lc;
                    }
                }
            }
            TILE_SYNC();
            PHASE_MARK(14);                                    // place final
        }
        }
    }
#ifdef CAPS_EMUL
    caps_emul_count_tile3(fast);
#endif
    if (!fast) {
        uint32_t* q_ = !VDEEP && redo_deep && flag[0] == 0 ? redo_deep : redo2;        // block-uniform
        PAR(tid) { if (tid == 0) q_[1 + FETCH_ADD_U32(&q_[0], 1u)] = b; }
    } else if (PERM) {
        // sorted order -> HBM through perm (+ LCPs: the tie phases' notes, else from the neighbours' keys, + boundary records)
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    const uint32_t sl = perm[e];
                    const uint64_t key = skey[sl];
                    const idx_t sa = ssa[sl];
                    uint64_t l = 0;
                    if (with_lcp && e) {
                        uint32_t ov = tinfo[sl];
                        if (VDEEP && ov == 0x7FFFu) ov = eq_big_lcp(vslot, vlcp2, pcnt[5], sl);
                        if (ov) l = ov;
                        else { const uint32_t sp = perm[e - 1]; l = TILE_EMIT_LCP_(skey[sp], ssa[sp], key, sa); }
                    }
                    if (direct) {
                        STREAM_STORE(&fin.sa[start + e], sa);
                        STREAM_STORE(&fin.lcp[start + e], (idx_t)l);
                        if (e == 0) { fin.first_key[g] = key; fin.first_sa[g] = sa; }
                        if (e == cnt - 1) { fin.last_key[g] = key; fin.last_sa[g] = sa; }
                    } else {
                        out_key[start + e] = key;
                        out_sa[start + e] = sa;
                        if (with_lcp) out_lcp[start + e] = (idx_t)l;
                    }
                }
            }
        }
    } else {
        TILE_SORT_EMIT_((!VDEEP || tinfo[e] != 0x7FFFu ? (uint32_t)tinfo[e] : eq_big_lcp(vslot, vlcp, pcnt[5], e)))
    }
    TILE_SYNC();                                                    // the staging arrays are free for the next tile
    if (!VDEEP && fast && flag[1]) {                                // block-uniform: eq_emit_lcp met equal keys without a note
        PAR(tid) { if (tid == 0) redo2[1 + FETCH_ADD_U32(&redo2[0], 1u)] = b; }
    }
    PHASE_MARK(15);                                            // emit (+ LCPs) and the barrier behind it
    qi += K_GRID_DIM;
    } while (PERSIST && qi < n_redo);
}

#undef TILE_SYNC
#define TILE_SYNC() SYNC()
#undef TILE_EMIT_LCP_
#define TILE_EMIT_LCP_(ka, a, kb, b) tile_pair_lcp<BITS, TILE_RUNS>(P, n, ka, a, kb, b, TILE_KEY_SHIFT)

// ---- tile_sort_general_kernel: the tiles tile_sort_kernel could not finish ------------------
// (keys far from uniform inside the tile, or equal keys: repeats), taken from its queue
// (redo[0] = length, redo[1..] = tile ids) by a fixed grid of workgroups, or one per entry (PERSIST).  Comparison based,
// hence independent of the key distribution:
//  1. samplesort in LDS: every 4th element is a sample; the (<= 1023) samples are sorted by
//     rank-merge levels; every element finds its bin among the sorted samples by a branch-free
//     fixed-depth binary search (exact comparator on key ties), a counting sort groups the bins,
//     and each element ranks itself exactly inside its bin (~4 candidates).  ~25 comparison
//     steps per suffix instead of ~78.  Gives up when a bin holds more than TILE_SAMPLE_LIMIT
//     suffixes;
//  2. bottom-up rank-merge levels over the whole tile (reference: merge_sort,
//     src/Suffix_Array.cpp:112-129): every element finds its slot by the same binary search in
//     the sibling run.
// (A second, finer level of interpolation bins for the overflowing bins was tried first: on
// Markov-skewed DNA it rescued < 20 % of the tiles and cost more than it saved.)
constexpr uint32_t TILE_SAMPLE_LIMIT = 64;

// PERSIST = false: one queue entry per workgroup, as in tile_sort_eq_kernel (the caller knows the queue's length)
template <typename idx_t, int BITS, bool FROM_TEXT, bool RUNS, bool PERSIST = true>
GLOBAL_FN LAUNCH_BOUNDS2(TILE_NT, TILE_WAVES_PER_SIMD) tile_sort_general_kernel(KCTX SegDesc sd, const uint32_t* __restrict__ P, uint64_t n,
                                                          uint64_t text_base, uint32_t lcp_mode, uint32_t slot_cap, const uint64_t* in_key,
                                                          const idx_t* in_sa, uint64_t* out_key, idx_t* out_sa, idx_t* out_lcp,
                                                          FinalOut<idx_t> fin, const uint32_t* __restrict__ redo, uint32_t keys_from_text,
                                                          uint64_t* __restrict__ bflag)
{
    // bflag != null: ties are deferred ("Deferred ties" above) -- equal keys are ordered by position, descending (the comparators
    // run as if the text had length 0: text.h), their LCPs are TIE_SENTINEL, and bflag[segment] says that the segment holds some
    const uint64_t n_cmp = bflag ? 0 : n;
    const uint32_t TILE_KEY_SHIFT = 0;
    (void)TILE_KEY_SHIFT;
    SHARED_ARRAY(uint64_t, skey, TILE_E);
    SHARED_ARRAY(idx_t, ssa, TILE_E);
    SHARED_ARRAY(uint64_t, smk, 2 * TILE_NT);       // samples (2x: the fixed-depth search may probe past the end)
    SHARED_ARRAY(idx_t, sms, 2 * TILE_NT);
    SHARED_ARRAY(uint32_t, hist, TILE_NT + 1);      // bin counters: #bins = #samples + 1 <= TILE_NT
    SHARED_ARRAY(uint32_t, flag, 1);
    TL_DECL(uint64_t, rk, TILE_EPT);
    TL_DECL(idx_t, rs, TILE_EPT);
    TL_DECL(uint32_t, rd, TILE_EPT);
    TL_DECL(uint32_t, rb, TILE_EPT);
    TL_DECL(uint64_t, sk1, 1);                      // the thread's sample while the samples are sorted
    TL_DECL(idx_t, ss1, 1);
    TL_DECL(uint32_t, sd1, 1);
    constexpr bool TILE_RUNS = RUNS;
    const uint32_t n_redo = redo[0];
    uint32_t qi = K_BLOCK_IDX;
    if (qi >= n_redo) return;
    do {
    const uint32_t b = redo[1 + qi];
    const uint32_t g = sd.tile_rec[b].g;
    const TileInfo t = tile_info(sd, b);
    const uint64_t start = t.s0 + (uint64_t)t.tl * TILE_E;
    const uint32_t cnt = (uint32_t)(t.s1 - start < TILE_E ? t.s1 - start : TILE_E);
    const bool with_lcp = lcp_mode != 0 && (t.s1 - t.s0) <= TILE_E;
    const bool direct = with_lcp && fin.sa != nullptr;
    // a bucket that outgrew its slot was put together at its place in the output arrays (TILE_SORT_PROLOGUE): sorted in place
    const bool in_slot = !TILE_OUTGROWN;
    const uint64_t in0 = slot_cap != 0 && in_slot ? (uint64_t)g * slot_cap : start;
    const uint64_t* src_key = tile_src(in_key, out_key, !in_slot);
    const idx_t* src_sa = in_slot ? in_sa : out_sa;
    const uint32_t S = cnt / 4 < TILE_NT - 1 ? cnt / 4 : TILE_NT - 1;      // samples
    PAR(tid) {
        if (tid == 0) flag[0] = 0;
        for (uint32_t i = tid; i <= TILE_NT; i += K_BLOCK_DIM) hist[i] = 0;
        UNROLL
        for (uint32_t k = 0; k < TILE_EPT; ++k) {
            const uint32_t e = tid + k * TILE_NT;
            if (e < cnt) {
                uint64_t key;
                idx_t sa;
                if (FROM_TEXT) {
                    key = window64<BITS>(P, text_base + start + e);
                    sa = (idx_t)(text_base + start + e);
                } else if (keys_from_text) {                      // 32-bit keys in the slots: the 64-bit key is cut from the text
                    sa = src_sa[in0 + e];
                    key = window64<BITS>(P, (uint64_t)sa);
                } else {
                    key = src_key[in0 + e];
                    sa = src_sa[in0 + e];
                }
                TL(rk, tid, k) = key;
                TL(rs, tid, k) = sa;
                skey[e] = key;
                ssa[e] = sa;
            }
        }
    }
    SYNC();
    bool done = false;
    if (S >= 16) {
        // ---- 1. samplesort.  Samples: one per thread, sorted by rank-merge levels.
        const uint32_t stride = cnt / S;
        PAR(tid) {
            if (tid < S) { smk[tid] = skey[tid * stride]; sms[tid] = ssa[tid * stride]; }
        }
        SYNC();
        for (uint32_t R = 1; R < S; R <<= 1) {
            PAR(tid) {
                uint64_t key1[1] = {0};
                idx_t sa1[1] = {0};
                uint32_t lo1[1] = {0}, hi1[1] = {0};
                uint32_t dbase = 0;
                if (tid < S) {
                    key1[0] = smk[tid];
                    sa1[0] = sms[tid];
                    const uint32_t run = tid / R;
                    uint32_t sib = (run ^ 1u) * R;
                    if (sib > S) sib = S;
                    lo1[0] = sib;
                    hi1[0] = sib + R < S ? sib + R : S;
                    dbase = (run & ~1u) * R + (tid - run * R) - sib;
                }
                multi_lower_bound<idx_t, BITS, 1, RUNS>(P, n_cmp, smk, sms, key1, sa1, lo1, hi1, 2 * R);
                TL(sk1, tid, 0) = key1[0];
                TL(ss1, tid, 0) = sa1[0];
                TL(sd1, tid, 0) = dbase + lo1[0];
            }
            SYNC();
            PAR(tid) {
                if (tid < S) { smk[TL(sd1, tid, 0)] = TL(sk1, tid, 0); sms[TL(sd1, tid, 0)] = TL(ss1, tid, 0); }
            }
            SYNC();
        }
        // bin(x) = #samples < x  (bin j = (sample j-1, sample j]); counting sort by bin
        const uint32_t top = pow2_above(S);
        PAR(tid) {
            UNROLL
            for (uint32_t gg = 0; gg < TILE_EPT; gg += LOCK_K) {
                uint64_t key[LOCK_K];
                idx_t sa[LOCK_K];
                uint32_t lo[LOCK_K], hi[LOCK_K];
                UNROLL
                for (uint32_t k = 0; k < LOCK_K; ++k) {
                    const uint32_t e = tid + (gg + k) * TILE_NT;
                    key[k] = TL(rk, tid, gg + k);
                    sa[k] = TL(rs, tid, gg + k);
                    lo[k] = 0;
                    hi[k] = e < cnt ? S : 0u;
                }
                multi_lower_bound<idx_t, BITS, LOCK_K, RUNS>(P, n_cmp, smk, sms, key, sa, lo, hi, top);
                UNROLL
                for (uint32_t k = 0; k < LOCK_K; ++k) {
                    const uint32_t e = tid + (gg + k) * TILE_NT;
                    if (e < cnt) {
                        const uint32_t r = FETCH_ADD_U32(&hist[lo[k]], 1u);
                        if (r >= TILE_SAMPLE_LIMIT) flag[0] = 1;
                        TL(rb, tid, gg + k) = lo[k];
                        TL(rd, tid, gg + k) = r;
                    }
                }
            }
        }
        SYNC();
        if (flag[0] == 0) {
            block_exclusive_scan<TILE_NT>(KCTX_PASS hist);        // hist[j] = first slot of bin j, hist[TILE_NT] = cnt
            PAR(tid) {
                UNROLL
                for (uint32_t k = 0; k < TILE_EPT; ++k) {
                    const uint32_t e = tid + k * TILE_NT;
                    if (e < cnt) {
                        const uint32_t slot = hist[TL(rb, tid, k)] + TL(rd, tid, k);
                        skey[slot] = TL(rk, tid, k);
                        ssa[slot] = TL(rs, tid, k);
                        TL(rd, tid, k) = slot;
                    }
                }
            }
            SYNC();
            PAR(tid) {
                UNROLL
                for (uint32_t k = 0; k < TILE_EPT; ++k) {
                    const uint32_t e = tid + k * TILE_NT;
                    if (e < cnt) {
                        const uint32_t bin = TL(rb, tid, k), slot = TL(rd, tid, k);
                        const uint32_t bs = hist[bin], be = hist[bin + 1];
                        const uint64_t key = TL(rk, tid, k);
                        const uint64_t sa = (uint64_t)TL(rs, tid, k);
                        uint32_t less = 0;                        // members of my bin that sort before me
                        for (uint32_t j = bs; j < be; ++j)
                            if (j != slot && suffix_less<BITS, RUNS>(P, n_cmp, skey[j], (uint64_t)ssa[j], key, sa)) ++less;
                        TL(rd, tid, k) = bs + less;
                    }
                }
            }
            SYNC();
            TILE_SORT_PLACE_FINAL
            done = true;
        }
    }
#ifdef CAPS_EMUL
    caps_emul_count_tile2(done);
#endif
    // ---- 2. bottom-up rank-merge levels
    if (!done && S >= 16) {
        PAR(tid) {                                  // the counting sort may have been abandoned half way: restore
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) { skey[e] = TL(rk, tid, k); ssa[e] = TL(rs, tid, k); }
            }
        }
        SYNC();
    }
    for (uint32_t R = 1; !done && R < cnt; R <<= 1) {
        PAR(tid) {
            UNROLL
            for (uint32_t gg = 0; gg < TILE_EPT; gg += LOCK_K) {          // LOCK_K searches in lockstep
                uint64_t key[LOCK_K];
                idx_t sa[LOCK_K];
                uint32_t lo[LOCK_K], hi[LOCK_K], dbase[LOCK_K];
                UNROLL
                for (uint32_t k = 0; k < LOCK_K; ++k) {
                    const uint32_t e = tid + (gg + k) * TILE_NT;
                    key[k] = 0;
                    sa[k] = 0;
                    lo[k] = hi[k] = dbase[k] = 0;
                    if (e < cnt) {
                        key[k] = skey[e];
                        sa[k] = ssa[e];
                        const uint32_t run = e / R;
                        uint32_t sib_start = (run ^ 1u) * R;
                        if (sib_start > cnt) sib_start = cnt;
                        lo[k] = sib_start;                               // #sibling elements < (key, sa)
                        hi[k] = sib_start + R < cnt ? sib_start + R : cnt;
                        dbase[k] = (run & ~1u) * R + (e - run * R) - sib_start;
                    }
                }
                multi_lower_bound<idx_t, BITS, LOCK_K, RUNS>(P, n_cmp, skey, ssa, key, sa, lo, hi, 2 * R);
                UNROLL
                for (uint32_t k = 0; k < LOCK_K; ++k) {
                    TL(rk, tid, gg + k) = key[k];
                    TL(rs, tid, gg + k) = sa[k];
                    TL(rd, tid, gg + k) = dbase[k] + lo[k];
                }
            }
        }
        SYNC();
        TILE_SORT_PLACE_FINAL
    }
#undef TILE_EMIT_LCP_
#define TILE_EMIT_LCP_(ka, a, kb, b) deferring_pair_lcp<BITS, TILE_RUNS, idx_t>(P, n, bflag, g, ka, a, kb, b)
    TILE_SORT_EMIT
#undef TILE_EMIT_LCP_
#define TILE_EMIT_LCP_(ka, a, kb, b) tile_pair_lcp<BITS, TILE_RUNS>(P, n, ka, a, kb, b, TILE_KEY_SHIFT)
    SYNC();                                        // before the next queued tile re-uses the LDS
    qi += K_GRID_DIM;
    } while (PERSIST && qi < n_redo);
}
#ifdef CAPS_PAR_SWITCHED
#pragma pop_macro("PAR")
#undef CAPS_PAR_SWITCHED
#endif

// ----------------------------------------------------------------------------------
// a3: LCP-merge of run pairs (reference: merge, src/Suffix_Array.cpp:48-109).
//
// A pass merges, inside every segment, runs (2q, 2q+1) of length R into runs of 2R.
// The output of a pair is cut into tiles of TILE_E outputs.
//
//  merge_partition_kernel  one thread per tile: "merge path" splits of the tile's first
//      and last output diagonal (binary search with the full suffix comparator) -> a
//      48-byte TileDesc {source pieces, destination, flags}.  Segments that are already
//      a single run (len <= R) are marked inactive when skip_finished is set: they stay
//      in the buffer their last pass wrote (parity of passes_for(len)), so skewed
//      partitions cost passes only for their own tiles.
//  merge_pass_kernel       persistent workgroups walk the tiles.  Per tile: the two input
//      pieces (key, sa) are staged in LDS, every element finds its output slot by a
//      binary search in the other piece (rank merge: no divergent serial merge), the
//      merged tile is written back coalesced.  The global loads of the NEXT tile are
//      issued into registers before the rank phase, so HBM latency overlaps the LDS work.
//
// LCPs: keys make the reference's LCP bookkeeping (`m`, `l_x`, cpp:59-79) unnecessary for
// ordering, so intermediate passes move no LCPs at all.  The pass that completes a segment
// (flag FINAL) emits them: LCP_z[k] = lcp(z[k-1], z[k]) from the adjacent keys of the merged
// tile (text only when the keys are equal); the predecessor of the tile's first output is
// the larger of the two elements preceding the pieces; a segment head keeps 0 (cpp:117).
// ----------------------------------------------------------------------------------
struct TileDesc {
    uint64_t srcA, srcB, dst;   // element offsets of the two input pieces and of the output
    uint32_t na, nb;            // piece lengths (na + nb <= TILE_E; 0/0 = inactive tile)
    uint32_t flags;             // TD_*
    uint32_t pad;               // the tile's segment
};
constexpr uint32_t TD_HALO_A = 1, TD_HALO_B = 2, TD_FINAL = 4;

// Number of merge passes a segment of `len` elements takes (runs E, 2E, 4E, ... < len).
HD uint32_t passes_for(uint64_t len)
{
    const uint64_t tiles = (len + TILE_E - 1) / TILE_E;
    return tiles <= 1 ? 0u : (uint32_t)(64 - caps_clz64(tiles - 1));
}

// stable: by key alone, an element of A before an element of B with the same key (deferred ties)
template <typename idx_t, int BITS, bool RUNS = true>
DEV_INLINE uint64_t merge_path_split(const uint32_t* __restrict__ P, uint64_t n, const uint64_t* __restrict__ key,
                                     const idx_t* __restrict__ sa, uint64_t A, uint64_t la, uint64_t B, uint64_t lb, uint64_t d,
                                     bool stable = false)
{
    uint64_t lo = d > lb ? d - lb : 0;
    uint64_t hi = d < la ? d : la;
    while (lo < hi) {                                      // #elements of A among the first d outputs
        const uint64_t mid = (lo + hi) >> 1;
        const uint64_t ia = A + mid, ib = B + (d - 1 - mid);
        const bool a_first = stable ? key[ia] <= key[ib] : suffix_less<BITS, RUNS>(P, n, key[ia], (uint64_t)sa[ia], key[ib], (uint64_t)sa[ib]);
        if (a_first) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

template <typename idx_t, int BITS, bool RUNS>
GLOBAL_FN LAUNCH_BOUNDS(256) merge_partition_kernel(KCTX SegDesc sd, const uint32_t* __restrict__ P, uint64_t n,
                                                    uint64_t R, uint64_t single_la, uint32_t skip_finished, uint32_t need_lcp,
                                                    const uint64_t* __restrict__ key, const idx_t* __restrict__ sa,
                                                    TileDesc* __restrict__ desc, uint64_t* __restrict__ moved, uint32_t defer,
                                                    uint32_t* __restrict__ n_active)
{
    // n_active (zeroed by the caller): counts the descriptors written -- desc[0 .. *n_active) is what merge_pass_kernel walks
    // moved (optional): += elements this pass really merges, one global atomic per workgroup
    // defer: the runs are merged by key alone, stably ("Deferred ties")
    SHARED_ARRAY(uint64_t, acc, 1);
    PAR(tid) { if (tid == 0) acc[0] = 0; }
    SYNC();
    PAR(tid) {
        const uint64_t b = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (b < sd.tile_off[sd.G]) {
            const TileInfo t = tile_info(sd, (uint32_t)b);
            const uint64_t seglen = t.s1 - t.s0;
            TileDesc d;
            d.srcA = d.srcB = d.dst = 0;
            d.na = d.nb = d.flags = d.pad = 0;
            if (single_la != ~0ull || !skip_finished || seglen > R) {
                const PairInfo pr = pair_info(t, R, single_la);
                const uint64_t tot = pr.la + pr.lb;
                const uint64_t d1 = pr.d0 + TILE_E < tot ? pr.d0 + TILE_E : tot;
                const uint64_t A = pr.a0, B = pr.a0 + pr.la;
                const uint64_t i0 = merge_path_split<idx_t, BITS, RUNS>(P, n, key, sa, A, pr.la, B, pr.lb, pr.d0, defer != 0);
                const uint64_t i1 = d1 < tot ? merge_path_split<idx_t, BITS, RUNS>(P, n, key, sa, A, pr.la, B, pr.lb, d1, defer != 0) : pr.la;
                const uint64_t j0 = pr.d0 - i0, j1 = d1 - i1;
                d.srcA = A + i0;
                d.srcB = B + j0;
                d.dst = pr.a0 + pr.d0;
                d.na = (uint32_t)(i1 - i0);
                d.nb = (uint32_t)(j1 - j0);
                d.flags = (i0 > 0 ? TD_HALO_A : 0u) | (j0 > 0 ? TD_HALO_B : 0u);
                if (need_lcp && (single_la != ~0ull || 2 * R >= seglen)) d.flags |= TD_FINAL;
                d.pad = t.g;                              // the segment (merge_pass_kernel flags it when it defers ties)
            }
            // only the tiles that merge something are listed (in any order: they are independent) -- a pass over a text whose
            // buckets nearly all fit a tile used to walk a million empty descriptors (3 of the 3.7 ms of a pass at 3e9)
            if (d.na + d.nb) desc[FETCH_ADD_U32(n_active, 1u)] = d;
            if (moved && (d.na + d.nb)) ATOMIC_ADD_LDS_U64(&acc[0], (uint64_t)(d.na + d.nb));
        }
    }
    SYNC();
    PAR(tid) { if (tid == 0 && moved && acc[0]) ATOMIC_ADD_U64(moved, acc[0]); }
}

template <typename idx_t, int BITS, bool RUNS>
GLOBAL_FN LAUNCH_BOUNDS2(TILE_NT, TILE_WAVES_PER_SIMD) merge_pass_kernel(KCTX const TileDesc* __restrict__ desc, const uint32_t* __restrict__ n_active,
                                                   const uint32_t* __restrict__ P, uint64_t n,
                                                   const uint64_t* __restrict__ in_key, const idx_t* __restrict__ in_sa,
                                                   uint64_t* __restrict__ out_key, idx_t* __restrict__ out_sa,
                                                   idx_t* __restrict__ out_lcp, uint64_t* __restrict__ bflag)
{
    // bflag != null: deferred ties -- stable merge by key, TIE_SENTINEL for equal neighbours, bflag[segment] = 1 where one was written
    SHARED_ARRAY(uint64_t, skey, TILE_E);
    SHARED_ARRAY(idx_t, ssa, TILE_E);
    TL_DECL(uint64_t, pk, TILE_EPT);      // prefetched (next tile)
    TL_DECL(idx_t, ps, TILE_EPT);
    TL_DECL(uint64_t, rk, TILE_EPT);      // element being ranked
    TL_DECL(idx_t, rs, TILE_EPT);
    TL_DECL(uint32_t, rd, TILE_EPT);

    const uint32_t n_tiles = n_active[0];              // the listed tiles (merge_partition_kernel)
    uint32_t t = K_BLOCK_IDX;
    if (t >= n_tiles) return;
    TileDesc d = desc[t];
    PAR(tid) {
        UNROLL
        for (uint32_t k = 0; k < TILE_EPT; ++k) {
            const uint32_t x = tid + k * TILE_NT;
            if (x < d.na + d.nb) {
                const uint64_t src = x < d.na ? d.srcA + x : d.srcB + (x - d.na);
                TL(pk, tid, k) = in_key[src];
                TL(ps, tid, k) = in_sa[src];
            }
        }
    }
    while (true) {
        const uint32_t na = d.na, cnt = d.na + d.nb;
        const uint32_t top = pow2_above(d.na > d.nb ? d.na : d.nb);      // search depth of this tile
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t x = tid + k * TILE_NT;
                if (x < cnt) { skey[x] = TL(pk, tid, k); ssa[x] = TL(ps, tid, k); }
            }
        }
        SYNC();
        // issue the next tile's loads; they stay in flight during the rank phase
        const uint32_t t_next = t + K_GRID_DIM;
        TileDesc dn = d;
        if (t_next < n_tiles) {
            dn = desc[t_next];
            PAR(tid) {
                UNROLL
                for (uint32_t k = 0; k < TILE_EPT; ++k) {
                    const uint32_t x = tid + k * TILE_NT;
                    if (x < dn.na + dn.nb) {
                        const uint64_t src = x < dn.na ? dn.srcA + x : dn.srcB + (x - dn.na);
                        TL(pk, tid, k) = in_key[src];
                        TL(ps, tid, k) = in_sa[src];
                    }
                }
            }
        }
        PAR(tid) {
            UNROLL
            for (uint32_t g = 0; g < TILE_EPT; g += LOCK_K) {            // LOCK_K searches in lockstep
                uint64_t key[LOCK_K];
                idx_t sa[LOCK_K];
                uint32_t lo[LOCK_K], hi[LOCK_K];
                bool up[LOCK_K];
                UNROLL
                for (uint32_t k = 0; k < LOCK_K; ++k) {
                    const uint32_t x = tid + (g + k) * TILE_NT;
                    key[k] = 0;
                    sa[k] = 0;
                    lo[k] = hi[k] = 0;
                    up[k] = false;
                    if (x < cnt) {
                        key[k] = skey[x];
                        sa[k] = ssa[x];
                        const bool fromA = x < na;
                        lo[k] = fromA ? na : 0u;                         // #elements of the other piece < (key, sa)
                        hi[k] = fromA ? cnt : na;
                        up[k] = !fromA;                                  // (deferred ties: B's elements behind A's equal keys)
                    }
                }
                if (bflag) multi_key_bound<LOCK_K>(skey, key, up, lo, hi, top);
                else multi_lower_bound<idx_t, BITS, LOCK_K, RUNS>(P, n, skey, ssa, key, sa, lo, hi, top);
                UNROLL
                for (uint32_t k = 0; k < LOCK_K; ++k) {
                    // slot = own rank + rank in the other piece = (x - na) + lo for both pieces
                    // (from A: x + (lo - na); from B: (x - na) + lo; unsigned wrap-around is fine)
                    TL(rk, tid, g + k) = key[k];
                    TL(rs, tid, g + k) = sa[k];
                    TL(rd, tid, g + k) = (tid + (g + k) * TILE_NT) - na + lo[k];
                }
            }
        }
        SYNC();
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t x = tid + k * TILE_NT;
                if (x < cnt) {
                    const uint32_t dd = TL(rd, tid, k);
                    skey[dd] = TL(rk, tid, k);
                    ssa[dd] = TL(rs, tid, k);
                }
            }
        }
        SYNC();
        if (d.flags & TD_FINAL) {
            // block-uniform scalar loads: the elements preceding the two pieces
            const bool hA = d.flags & TD_HALO_A, hB = d.flags & TD_HALO_B;
            const uint64_t hAk = hA ? in_key[d.srcA - 1] : 0, hBk = hB ? in_key[d.srcB - 1] : 0;
            const uint64_t hAs = hA ? (uint64_t)in_sa[d.srcA - 1] : 0, hBs = hB ? (uint64_t)in_sa[d.srcB - 1] : 0;
            PAR(tid) {
                UNROLL
                for (uint32_t k = 0; k < TILE_EPT; ++k) {
                    const uint32_t x = tid + k * TILE_NT;
                    if (x < cnt) {
                        const uint64_t key = skey[x];
                        const uint64_t sa = (uint64_t)ssa[x];
                        uint64_t l = 0;
                        if (bflag) {
                            // equal keys anywhere before me = an equal key right before me (the pieces are sorted by key)
                            const bool tie = x ? skey[x - 1] == key : (hA && hAk == key) || (hB && hBk == key);
                            if (tie) { l = (uint64_t)tie_sentinel<idx_t>(); bflag[d.pad] = 1; }
                            else if (x) l = pair_lcp<BITS, false>(P, n, skey[x - 1], (uint64_t)ssa[x - 1], key, sa);
                            else {
                                if (hA) l = pair_lcp<BITS, false>(P, n, hAk, hAs, key, sa);
                                if (hB) { const uint64_t l2 = pair_lcp<BITS, false>(P, n, hBk, hBs, key, sa); l = l2 > l ? l2 : l; }
                            }
                        } else
                        if (x) l = pair_lcp<BITS, RUNS>(P, n, skey[x - 1], (uint64_t)ssa[x - 1], key, sa);
                        else {
                            if (hA) l = pair_lcp<BITS, RUNS>(P, n, hAk, hAs, key, sa);
                            if (hB) { const uint64_t l2 = pair_lcp<BITS, RUNS>(P, n, hBk, hBs, key, sa); l = l2 > l ? l2 : l; }
                        }
                        out_key[d.dst + x] = key;
                        out_sa[d.dst + x] = (idx_t)sa;
                        out_lcp[d.dst + x] = (idx_t)l;
                    }
                }
            }
        } else {
            PAR(tid) {
                UNROLL
                for (uint32_t k = 0; k < TILE_EPT; ++k) {
                    const uint32_t x = tid + k * TILE_NT;
                    if (x < cnt) { out_key[d.dst + x] = skey[x]; out_sa[d.dst + x] = ssa[x]; }
                }
            }
        }
        if (t_next >= n_tiles) break;
        SYNC();
        t = t_next;
        d = dn;
    }
}

// Result of a segmented sort with skip_finished: segment g sits in buffer (passes_for(len_g) & 1)
// (0 = the buffer the tile sort wrote).  finalize_kernel gathers SA and LCP of every segment
// into the caller's arrays and computes the LCP at the head of every segment but the first
// (a11: compute_partition_boundary_lcp, src/Suffix_Array.cpp:431-447); empty segments are
// skipped (the reference runs out of bounds on trailing empty partitions).
template <typename idx_t> struct PingPong {
    const uint64_t* key[2];
    const idx_t* sa[2];
    const idx_t* lcp[2];
};

template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) finalize_kernel(KCTX SegDesc sd, const uint32_t* __restrict__ P, uint64_t n, PingPong<idx_t> pp,
                                             uint32_t skip_finished, uint32_t uniform_sel, idx_t* __restrict__ dSA,
                                             idx_t* __restrict__ dLCP, FinalOut<idx_t> fin)
{
    const uint32_t b = K_BLOCK_IDX;
    if (b >= sd.tile_off[sd.G]) return;
    const uint32_t g = sd.tile_rec[b].g;
    const TileInfo t = tile_info(sd, b);
    const uint64_t seglen = t.s1 - t.s0;
    // with boundary records (fin): segments completed by the tile sort are already in place
    if (fin.sa != nullptr && seglen <= TILE_E) return;
    const uint64_t start = t.s0 + (uint64_t)t.tl * TILE_E;
    const uint32_t cnt = (uint32_t)(t.s1 - start < TILE_E ? t.s1 - start : TILE_E);
    const uint32_t sel = skip_finished ? (passes_for(seglen) & 1u) : uniform_sel;
    const idx_t* __restrict__ src_sa = pp.sa[sel];
    const idx_t* __restrict__ src_lcp = pp.lcp[sel];
    PAR(tid) {
        for (uint32_t e = tid; e < cnt; e += K_BLOCK_DIM) {
            dSA[start + e] = src_sa[start + e];
            uint64_t l = (uint64_t)src_lcp[start + e];
            if (fin.sa != nullptr) {
                if (e == 0 && t.tl == 0) { fin.first_key[g] = pp.key[sel][start]; fin.first_sa[g] = src_sa[start]; }
                if (start + e == t.s1 - 1) { fin.last_key[g] = pp.key[sel][start + e]; fin.last_sa[g] = src_sa[start + e]; }
            } else if (e == 0 && t.tl == 0 && start > 0) {
                // head of segment g: predecessor = last element of the nearest non-empty segment below
                uint32_t h = g;
                while (h > 0 && sd.seg_start[h] == sd.seg_start[h - 1]) --h;      // skip empty ones
                // seg_start[h] == start; the previous segment [seg_start[h-1], start) is non-empty (start > 0)
                const uint64_t plen = sd.seg_start[h] - sd.seg_start[h - 1];
                const uint32_t psel = skip_finished ? (passes_for(plen) & 1u) : uniform_sel;
                l = pair_lcp<BITS>(P, n, pp.key[psel][start - 1], (uint64_t)pp.sa[psel][start - 1], pp.key[sel][start],
                                   (uint64_t)src_sa[start]);
            }
            dLCP[start + e] = (idx_t)l;
        }
    }
}

// a11 with boundary records: LCP at the head of every non-empty segment but the first =
// lcp(last suffix of the nearest non-empty segment below, first suffix of this one).
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) head_lcp_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, const uint64_t* __restrict__ seg_start,
                                             uint32_t G, FinalOut<idx_t> fin, uint32_t from_text)
{
    // from_text: the boundary records hold 32-bit keys of (possibly) different groups: the head LCPs come from the text
    PAR(tid) {
        const uint64_t g = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (g < G && g > 0) {
            const uint64_t at = seg_start[g];
            if (seg_start[g + 1] > at && at > 0) {
                uint64_t h = g - 1;
                while (seg_start[h + 1] == seg_start[h]) --h;             // at > 0: a non-empty one exists below
                fin.lcp[at] = from_text ? (idx_t)deep_lcp<BITS>(P, n, (uint64_t)fin.last_sa[h], (uint64_t)fin.first_sa[g], 0)
                                        : (idx_t)pair_lcp<BITS>(P, n, fin.last_key[h], (uint64_t)fin.last_sa[h], fin.first_key[g],
                                                                (uint64_t)fin.first_sa[g]);
            }
        }
    }
}

// Maps a launch-order block id to a logical block id such that the blocks that share
// an XCD (b % 8, MI355X_MICROARCH "Workgroup dispatch") work on a contiguous range of
// logical blocks: consecutive logical blocks -- here the searches over one subarray --
// then hit the same 4 MiB L2.  Speed only; any placement is correct.
DEV_INLINE uint64_t xcd_swizzle(uint64_t b, uint64_t nb)
{
    const uint64_t q = nb / 8, r = nb % 8, x = b % 8, y = b / 8;
    return x * q + (x < r ? x : r) + y;
}

// ---- phase 2 straight from the sorted subarrays (a9 + a10 in one pass) ----------------
// The reference first gathers every partition (p^2 memcpy's, cpp:343-358) and then sorts it.
// On one GPU the bucket split of phase 2 can read the partition THROUGH the partition matrix
// instead: element x of partition j is element  PmT[j][g] + (x - rulerT[j][g])  of sorted
// subarray g, g = the run with rulerT[j][g] <= x < rulerT[j][g+1].  That removes one full
// read + write of (key, sa) -- the collate pass -- from the build.  PmT / rulerT are the
// transposes of Pm / ruler (a partition's runs are then consecutive in memory).
template <typename idx_t> struct RunSrc {
    const idx_t* PmT = nullptr;            // [(p+1) x G1]  PmT[j * G1 + g] = Pm[g][j]
    const idx_t* rulerT = nullptr;         // [p x G1]      rulerT[j * G1 + g] = ruler[g][j]
    const uint64_t* sub_start = nullptr;   // [G1 + 1] element offsets of the sorted subarrays
    const uint32_t* first_run = nullptr;   // [#tiles of the partitions] run holding the tile's first element
    uint32_t G1 = 0;
};

// out[c * rows + r] = in[r * cols + c]; 32 x 32 tiles through LDS, 256 threads.
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) transpose_kernel(KCTX const idx_t* __restrict__ in, uint32_t rows, uint32_t cols,
                                              idx_t* __restrict__ out)
{
    SHARED_ARRAY(idx_t, tile, 32 * 33);
    const uint32_t tc = (cols + 31) / 32;
    const uint64_t total = (uint64_t)tc * ((rows + 31) / 32);
    for (uint64_t v = K_BLOCK_IDX; v < total; v += K_GRID_DIM) {
        const uint32_t r0 = (uint32_t)(v / tc) * 32, c0 = (uint32_t)(v % tc) * 32;
        PAR(tid) {
            const uint32_t x = tid & 31, y = tid >> 5;             // 8 rows of 32 per step
            for (uint32_t yy = y; yy < 32; yy += 8)
                if (r0 + yy < rows && c0 + x < cols) tile[yy * 33 + x] = in[(uint64_t)(r0 + yy) * cols + c0 + x];
        }
        SYNC();
        PAR(tid) {
            const uint32_t x = tid & 31, y = tid >> 5;
            for (uint32_t yy = y; yy < 32; yy += 8)
                if (c0 + yy < cols && r0 + x < rows) out[(uint64_t)(c0 + yy) * rows + r0 + x] = tile[x * 33 + yy];
        }
        SYNC();
    }
}

// largest g in [lo, hi] with row[g] <= x   (row[lo] <= x)
template <typename idx_t>
DEV_INLINE uint32_t run_of(const idx_t* row, uint32_t lo, uint32_t hi, uint64_t x)
{
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if ((uint64_t)row[mid] <= x) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// One thread per tile of the partitions: the run that holds the tile's first element.
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) runs_plan_kernel(KCTX SegDesc sd, const idx_t* __restrict__ rulerT, uint32_t G1,
                                              uint32_t* __restrict__ first_run)
{
    PAR(tid) {
        const uint64_t b = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (b < sd.tile_off[sd.G]) {
            const uint32_t j = sd.tile_rec[b].g;
            const TileInfo t = tile_info(sd, (uint32_t)b);
            first_run[b] = run_of<idx_t>(rulerT + (uint64_t)j * G1, 0u, G1 - 1, (uint64_t)t.tl * TILE_E);
        }
    }
}

// The runs a tile of partition j spans, staged in LDS: lrow[k] = rulerT[j][ga + k],
// lsrc[k] = (offset of run ga + k in the phase-1 arrays) - lrow[k], so that element x of the
// partition is phase-1 element lsrc[k] + x.  ns = #runs (<= TILE_E to be staged).
#define RUNS_TILE_SETUP                                                                                         \
    const uint32_t run_a = FROM_RUNS ? rsrc.first_run[b] : 0u;                                                  \
    const uint32_t run_b = !FROM_RUNS ? 0u : (start + cnt >= t.s1) ? rsrc.G1 - 1 : rsrc.first_run[b + 1];       \
    const uint32_t run_ns = run_b - run_a + 1;                                                                  \
    const bool runs_staged = run_ns <= TILE_E;                                                                  \
    const idx_t* run_rowR = rsrc.rulerT + (uint64_t)g * rsrc.G1;                                                \
    const idx_t* run_rowA = rsrc.PmT + (uint64_t)g * rsrc.G1;                                                   \
    const uint64_t run_x0 = start - t.s0;                                                                       \
    (void)run_a; (void)run_b; (void)run_ns; (void)runs_staged; (void)run_rowR; (void)run_rowA; (void)run_x0;

#define RUNS_STAGE(tid)                                                                                         \
    if (runs_staged)                                                                                            \
        for (uint32_t k = tid; k < run_ns; k += K_BLOCK_DIM) {                                                  \
            const idx_t r = run_rowR[run_a + k];                                                                \
            lrow[k] = r;                                                                                        \
            lsrc[k] = rsrc.sub_start[run_a + k] + (uint64_t)run_rowA[run_a + k] - (uint64_t)r;                  \
        }

template <typename idx_t>
DEV_INLINE uint64_t run_source(const RunSrc<idx_t>& rsrc, bool staged, const idx_t* lrow, const uint64_t* lsrc, uint32_t ns,
                               const idx_t* rowR, const idx_t* rowA, uint32_t ga, uint32_t gb, uint64_t x)
{
    if (staged) return lsrc[run_of<idx_t>(lrow, 0u, ns - 1, x)] + x;
    const uint32_t r = run_of<idx_t>(rowR, ga, gb, x);
    return rsrc.sub_start[r] + (uint64_t)rowA[r] + (x - (uint64_t)rowR[r]);
}

constexpr int SRC_ARRAYS = 0, SRC_TEXT = 1, SRC_RUNS = 2;      // where a bucket split reads its elements

// ----------------------------------------------------------------------------------
// Bucketing inside a sort (a4/a10): before the tile sort, every segment longer than a tile
// is split by KEY RANGE into B = ceil(len / BUCKET_TARGET) buckets; bucket(key) is a
// monotone map of the key (linear interpolation between the segment's smallest and largest
// possible key), so the buckets are consecutive slices of the sorted segment and become the
// segments of the tile sort.  On keys that are roughly uniform inside their range (random
// DNA: exactly) every bucket fits one tile and the segment needs NO merge pass -- two
// streaming passes (count, scatter) replace log2(len / TILE_E) merge passes.  Buckets that
// come out larger than a tile (skewed or repetitive text) are finished by the LCP-merge
// passes, which only touch those buckets (skip_finished).
//
//   bucket_plan_kernel     per segment: B, range normalisation (kmin, shift, Bq)
//   bucket_count_kernel    per element: ++count[bucket]      (LDS histogram per tile when B <= BUCKET_LDS)
//   bucket_scatter_kernel  per element: slot = start[bucket] + cursor[bucket]++  (same aggregation)
//   unify_kernel           copies the (few) buckets whose last pass ended in the other ping-pong
//                          buffer back, for consumers that index a segment as one array
// ----------------------------------------------------------------------------------
#ifndef CAPS_BUCKET_EIGHTHS
#define CAPS_BUCKET_EIGHTHS 7
#endif
#ifndef CAPS_SCATTER_SWZ
#define CAPS_SCATTER_SWZ 1
#endif
#ifndef CAPS_COLLATE_SWZ
#define CAPS_COLLATE_SWZ 0
#endif
#ifndef CAPS_TILE_SWZ
#define CAPS_TILE_SWZ 0
#endif
constexpr uint32_t BUCKET_TARGET = (TILE_E * CAPS_BUCKET_EIGHTHS) / 8;   // mean bucket size (headroom for the spread of bucket sizes)
constexpr uint32_t BUCKET_LDS = TILE_BINS_;            // buckets per segment the LDS histogram can hold

// (kmin exclusive unless the first group, kmax inclusive) in 64-bit keys -> [kmin, kmax] in the group's 32-bit keys
DEV_INLINE void key32_range(uint64_t& kmin, uint64_t& kmax, bool kmin_exclusive, uint32_t cs)
{
    const uint64_t lo = kmin_exclusive ? kmin + 1 : kmin;
    if (lo > kmax || (kmin_exclusive && kmin == ~0ull)) { kmin = kmax = 0; return; }      // an empty group
    kmin = key32_of(lo, cs);
    kmax = key32_of(kmax, cs);
}

// range_mode 0: keys span the whole 64-bit range (subarrays of text positions);
// range_mode 1: segment g is partition j = part_off + g of part_total and holds keys in
// [pkey[j-1], pkey[j]] (partitions between pivots; a shard owns a slice of the partitions).
GLOBAL_FN LAUNCH_BOUNDS(256) bucket_plan_kernel(KCTX const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ seg_end,
                                                uint32_t G, uint32_t range_mode,
                                                const uint64_t* __restrict__ pkey, uint32_t part_off, uint32_t part_total,
                                                uint32_t enable, uint32_t fine, uint32_t sub,
                                                BucketParams* __restrict__ bp, uint64_t* __restrict__ segB,
                                                const uint8_t* __restrict__ gshift)
{
    // range_mode 2: as 1, in the space of the parent's 32-bit keys (key32_of with gshift[parent], text.h)
    // sub > 1: every `sub` consecutive segments are the sub-streams of ONE parent (a group of the direct path, written
    // by level A as one stream per XCD): they share the parent's key range and its buckets.  All of them get the parent's
    // map; the bucket count is credited to the LAST sub-stream only, so that the exclusive scan of segB gives every
    // sub-stream the same first bucket.
    PAR(tid) {
        const uint32_t g = K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (g < G) {
            const uint32_t parent = g / sub, first = parent * sub;
            uint64_t len = 0;
            for (uint32_t x = 0; x < sub && first + x < G; ++x) len += seg_end_of(seg_start, seg_end, first + x) - seg_start[first + x];
            uint64_t kmin = 0, kmax = ~0ull;
            if (range_mode >= 1) {                   // the parent is partition part_off + parent of part_total
                const uint32_t j = part_off + parent;
                if (j > 0) kmin = pkey[j - 1];
                if (j + 1 < part_total) kmax = pkey[j];
                if (range_mode == 2) key32_range(kmin, kmax, j > 0, gshift[j]);
            }
            uint32_t B = 1;
            if (enable && len > TILE_E && kmax > kmin) B = (uint32_t)((len + BUCKET_TARGET - 1) / BUCKET_TARGET);
            // sub-streams never take the scatter's one-bucket shortcut (an identity copy of ONE segment): at least two buckets
            if (sub > 1 && B < 2) B = 2;
            // fine > 1: the map of the equalised split's FINE buckets (bucket_group_kernel): k <= `fine` per bucket slot,
            // as many as the LDS histograms hold.  A whole number per slot, so that the plain split (k consecutive fine
            // buckets per slot) is one of the groupings and the chosen one is never worse; segments with more than
            // BUCKET_LDS / 2 slots keep one fine bucket per slot.
            if (fine > 1 && B > 1 && B <= BUCKET_LDS) {
                const uint32_t k = BUCKET_LDS / B < fine ? BUCKET_LDS / B : fine;
                B *= k;
            }
            const BucketParams q = make_bucket_params(kmin, kmax, B);
            bp[g] = q;
            segB[g] = (g % sub == sub - 1 || g == G - 1) ? B : 0;
        }
    }
}

// Equalised split (the redo path of a bucket split whose slots overflowed: keys far from uniform in the
// segment's range).  The count pass has filled fcount[] for the segment's F FINE buckets (linear map with
// `fine` times the resolution); consecutive fine buckets are grouped greedily into the segment's B bucket
// slots, each group as close to a tile as it gets without exceeding it.  gfirst[slot] = first fine bucket
// of the group (F for unused slots), count[slot] = its exact size: no second count pass.  A fine bucket
// that is larger than a tile by itself is a group of its own (LCP-merge passes finish it, as before); if
// a tile per group needs more slots than the segment has, the capacity is raised to the smallest that fits.
GLOBAL_FN LAUNCH_BOUNDS(256) bucket_group_kernel(KCTX uint32_t G, const uint64_t* __restrict__ segB, const uint64_t* __restrict__ bstart,
                                                 const uint64_t* __restrict__ fsegB, const uint64_t* __restrict__ fstart,
                                                 const uint64_t* __restrict__ fcount, uint64_t* __restrict__ count,
                                                 uint32_t* __restrict__ gfirst)
{
    // One workgroup per segment: the fine counts are staged in LDS with coalesced loads (the greedy walks are serial,
    // and a dependent global load per step made this kernel 5 ms at 8000 segments x 1680 fine buckets), thread 0 walks.
    SHARED_ARRAY(uint64_t, fc, BUCKET_LDS);
    for (uint32_t g = K_BLOCK_IDX; g < G; g += K_GRID_DIM) {
        const uint32_t B = (uint32_t)segB[g], F = (uint32_t)fsegB[g];
        const uint64_t b0 = bstart[g], f0 = fstart[g];
        if (B > BUCKET_LDS || F <= B || F > BUCKET_LDS) {   // no finer map (bucket_plan_kernel): every fine bucket is its own slot
            PAR(tid) {
                for (uint32_t i = tid; i < B; i += K_BLOCK_DIM) { count[b0 + i] = i < F ? fcount[f0 + i] : 0; gfirst[b0 + i] = i; }
            }
            continue;
        }
        PAR(tid) { for (uint32_t f = tid; f < F; f += K_BLOCK_DIM) fc[f] = fcount[f0 + f]; }
        SYNC();
        PAR(tid) {
            if (tid == 0) {
                // capacity of a group: a tile if the slots suffice; otherwise the smallest capacity that fits the segment
                // into its B slots (never worse than the B equal key ranges of the plain split, which is one such grouping)
                uint64_t cap = TILE_E, total = 0;
                for (uint32_t f = 0; f < F; ++f) total += fc[f];
                for (uint64_t lo = TILE_E, hi = total > TILE_E ? total : TILE_E;;) {
                    const uint64_t c_try = lo == TILE_E ? lo : (lo + hi) / 2;     // first probe: a tile
                    uint32_t need = 1;
                    uint64_t run = 0;
                    for (uint32_t f = 0; f < F; ++f) {
                        const uint64_t c = fc[f];
                        if (run > 0 && run + c > c_try) { ++need; run = 0; }
                        run += c;
                    }
                    if (need <= B) { cap = c_try; hi = c_try; if (c_try == TILE_E) break; }
                    else lo = c_try + 1;
                    if (lo >= hi) { cap = hi; break; }
                }
                uint32_t gi = 0;
                uint64_t sum = 0;
                gfirst[b0] = 0;
                for (uint32_t f = 0; f < F; ++f) {
                    const uint64_t c = fc[f];
                    if (sum > 0 && sum + c > cap && gi + 1 < B) {
                        count[b0 + gi] = sum;
                        ++gi;
                        gfirst[b0 + gi] = f;
                        sum = 0;
                    }
                    sum += c;
                }
                count[b0 + gi] = sum;
                for (uint32_t i = gi + 1; i < B; ++i) { count[b0 + i] = 0; gfirst[b0 + i] = F; }
            }
        }
        SYNC();                                          // fc is free for the next segment
    }
}

// Key range of every bucket (inverse of bucket_of, to within rounding: keys just outside are
// clamped into the edge bins of the tile sort, which stays correct), stored as the tile sort's
// bin map over that range.  One thread per bucket.
GLOBAL_FN LAUNCH_BOUNDS(256) bucket_ranges_kernel(KCTX const uint64_t* __restrict__ bstart, uint32_t G,
                                                  const BucketParams* __restrict__ bps, const uint64_t* __restrict__ pkey,
                                                  uint32_t range_mode, uint32_t part_off, uint32_t part_total, uint32_t sub,
                                                  BucketParams* __restrict__ tile_map,
                                                  const BucketParams* __restrict__ fbps, const uint32_t* __restrict__ gfirst,
                                                  const uint8_t* __restrict__ gshift, uint8_t* __restrict__ kshift)
{
    // range_mode 2 (32-bit keys): the ranges are in the parent's key32 space; kshift[bucket] = the parent's shift
    // gfirst != null (equalised split): bucket slot i is the group of fine buckets [gfirst[i], gfirst[i + 1]) of fbps[g]
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < bstart[G]) {
            uint32_t a = 0, b = G;                     // parent segment: largest g with bstart[g] <= i
            while (b - a > 1) {
                const uint32_t mid = (a + b) / 2;
                if (bstart[mid] <= i) a = mid; else b = mid;
            }
            const uint32_t g = a;
            const BucketParams bp = bps[g];
            uint64_t kmin = 0, kmax = ~0ull;
            if (range_mode >= 1) {
                const uint32_t j = part_off + g / sub;            // sub-streams of one parent share its buckets (bucket_plan_kernel)
                if (j > 0) kmin = pkey[j - 1];
                if (j + 1 < part_total) kmax = pkey[j];
                if (range_mode == 2) { key32_range(kmin, kmax, j > 0, gshift[j]); kshift[i] = gshift[j]; }
            }
            const uint32_t bk = (uint32_t)(i - bstart[g]);
            const double range1 = (double)(kmax - kmin) + 1.0;
            uint64_t l = kmin, h = kmax;
            if (bp.B > 1) {
                uint32_t lo = bk, hi = bk + 1, div = bp.B;
                if (gfirst) {
                    div = fbps[g].B;
                    lo = gfirst[i];
                    hi = bk + 1 < bp.B ? gfirst[i + 1] : div;
                    if (hi <= lo) hi = lo + 1;                         // an unused slot: any range will do
                }
                const double dl = (double)lo / (double)div * range1, dh = (double)hi / (double)div * range1;
                if (lo > 0) l = kmin + (uint64_t)dl;
                if (hi < div) h = kmin + (uint64_t)dh;
                if (h < l) h = l;
            }
            tile_map[i] = make_bucket_params(l, h, TILE_BINS);      // the tile sort's bin map over the bucket's key range
        }
    }
}

// Keys of a tile of consecutive text positions are cut from a copy of the tile's slice of the
// packed text in LDS (one coalesced read of ~1 KiB) instead of three dependent global loads
// per suffix.  TEXT_WIN words cover TILE_E positions + one key + alignment slack for BITS = 8.
constexpr uint32_t TEXT_WIN = TILE_E / 4 + 24;
// cells of the splitter LUT of bucket_scatter_kernel<MAP_SPLIT> (split_lut_kernel)
constexpr uint32_t SPLIT_LUT_BITS = TILE_E >= 4096 ? 12 : TILE_E >= 256 ? 8 : 4;
constexpr uint32_t SPLIT_LUT_CELLS = 1u << SPLIT_LUT_BITS;

// word index (in P) of the first staged word for a tile whose first text position is pos0
template <int BITS> HD uint64_t text_win_base(uint64_t pos0) { return pos0 / TextTraits<BITS>::CPW; }

// MAP: how a key finds its bucket inside its parent segment --
//   MAP_LINEAR   linear interpolation over the segment's key range (bps[g]);
//   MAP_GROUPED  equalised split: fine linear bucket (fbps[g]) -> group table (gfirst) -> bucket slot;
//   MAP_SPLIT    splitter table: bucket = #{splitters < key} among the bps[g].B - 1 sorted keys of the segment's table
//                (split + parent * split_stride, parent = g / in_sub; one table for all when split_stride = 0), staged in
//                LDS and searched branch-free in lockstep for the thread's elements -- through a LUT over the keys' top
//                bits when split_lut is given (level A of the direct path: one table over the whole key range), from
//                scratch otherwise (level B in quantile mode: the knots of the segment's group).  Keys only: all
//                suffixes with one key value land in one bucket, so buckets are consecutive slices of the suffix order.
constexpr int MAP_LINEAR = 0, MAP_GROUPED = 1, MAP_SPLIT = 2;
// Where entry i of a sorted table (<= 4096 entries of 8 bytes) sits in LDS.  A lock-step binary search probes, in step j, the
// entries (2 m + 1) 2^j - 1: a stride of 2^(j + 1) entries -- for j >= 3 every probe of a wave falls into the same LDS banks,
// and the upper steps of the search are serialised up to 32 ways (PMC: 87 % of the LDS cycles of the quantile count pass were
// bank conflicts).  XOR-ing the next two index nibbles into the lowest one (which picks the bank pair inside a 128-byte row)
// spreads those probes over all banks; a bijection inside every aligned block of 4096 entries.
HD uint32_t lds_swz(uint32_t i) { return i ^ ((i >> 4) & 15u) ^ ((i >> 8) & 15u); }
#ifdef CAPS_GA_NO_SWZ              /* measurement: level A's splitter table unswizzled */
#define GA_SWZ(i) (i)
#else
#define GA_SWZ(i) lds_swz(i)
#endif
constexpr uint32_t COUNT_CHUNK = 8;            // tiles per chunk of bucket_count_kernel

// Persistent workgroups (a tile is little work: launching one workgroup per tile is bound by
// the wave launch rate).
template <typename idx_t, int BITS, int SRC, int MAP = MAP_LINEAR>
GLOBAL_FN LAUNCH_BOUNDS(TILE_NT) bucket_count_kernel(KCTX SegDesc sd, const uint32_t* __restrict__ P, uint64_t n_words,
                                                     uint64_t text_base, const uint64_t* __restrict__ in_key, RunSrc<idx_t> rsrc,
                                                     const BucketParams* __restrict__ bps, const uint64_t* __restrict__ bstart,
                                                     uint64_t* __restrict__ count, const uint64_t* __restrict__ split,
                                                     uint32_t split_stride, uint32_t in_sub, uint16_t* __restrict__ bid)
{
    // bid (MAP_SPLIT, optional): bid[element] = its bucket inside the parent -- the scatter that follows reads it back
    // instead of searching the knots again (10 dependent LDS reads per suffix, as much as everything else it does)
    constexpr bool FROM_TEXT = SRC == SRC_TEXT, FROM_RUNS = SRC == SRC_RUNS;
    SHARED_ARRAY(uint32_t, hist, BUCKET_LDS);
    SHARED_ARRAY(uint64_t, stab, MAP == MAP_SPLIT ? BUCKET_LDS : 1);
    SHARED_ARRAY(uint32_t, twin, FROM_TEXT ? TEXT_WIN : 1);
    SHARED_ARRAY(uint64_t, lsrc, FROM_RUNS ? TILE_E : 1);
    SHARED_ARRAY(idx_t, lrow, FROM_RUNS ? TILE_E : 1);
    // A workgroup takes CHUNKS of consecutive tiles and keeps one LDS histogram across the tiles of a chunk that count into
    // the same buckets (consecutive tiles of a long segment, or of the sub-streams of one parent): the histogram reaches the
    // global counters once per chunk instead of once per tile -- at ~1000 buckets per segment and 4096 elements per tile
    // that flush is one global atomic per 4 elements (genome-like 3e9, quantile mode: 20 of 162 ms were this kernel).
    const uint32_t n_tiles = sd.tile_off[sd.G];
    const uint32_t want = n_tiles / (4u * K_GRID_DIM);
    const uint32_t CH = want < 1u ? 1u : want > COUNT_CHUNK ? COUNT_CHUNK : want;          // short launches keep every workgroup busy
    const uint32_t n_chunks = (n_tiles + CH - 1) / CH;
    for (uint32_t v = K_BLOCK_IDX; v < n_chunks; v += K_GRID_DIM) {
    uint64_t cur_b0 = ~0ull;                           // block-uniform: the buckets the LDS histogram counts for (~0: none)
    uint32_t cur_B = 0;
    const uint32_t b_end = (v + 1) * CH < n_tiles ? (v + 1) * CH : n_tiles;
    for (uint32_t b = v * CH; b < b_end; ++b) {
        const uint32_t g = sd.tile_rec[b].g;
        const TileInfo t = tile_info(sd, b);
        const uint64_t start = t.s0 + (uint64_t)t.tl * TILE_E;
        const uint32_t cnt = (uint32_t)(t.s1 - start < TILE_E ? t.s1 - start : TILE_E);
        const BucketParams bp = bps[g];
        const uint64_t b0 = bstart[g];
        if (bp.B == 1) {                               // the segment is its own bucket
            PAR(tid) { if (tid == 0) ATOMIC_ADD_U64(&count[b0], (uint64_t)cnt); }
            continue;
        }
        const bool lds = bp.B <= BUCKET_LDS;
        const uint64_t w0 = text_win_base<BITS>(text_base + start);
        RUNS_TILE_SETUP
        const uint32_t n_split = MAP == MAP_SPLIT ? bp.B - 1 : 0u;
        const uint64_t* tab = MAP == MAP_SPLIT ? split + (uint64_t)(g / in_sub) * split_stride : nullptr;
        // same buckets as the tile before (and, MAP_SPLIT, the same table: it belongs to the parent that owns the buckets)?
        const bool fresh = !lds || b0 != cur_b0 || bp.B != cur_B;
        PAR(tid) {
            if (fresh) {
                for (uint32_t i = tid; i < cur_B; i += K_BLOCK_DIM)
                    if (hist[i]) ATOMIC_ADD_U64(&count[cur_b0 + i], (uint64_t)hist[i]);
                if (lds) for (uint32_t i = tid; i < bp.B; i += K_BLOCK_DIM) hist[i] = 0;
                if (MAP == MAP_SPLIT && lds) for (uint32_t i = tid; i < n_split; i += K_BLOCK_DIM) stab[lds_swz(i)] = tab[i];
            }
            if (FROM_TEXT)
                for (uint32_t i = tid; i < TEXT_WIN; i += K_BLOCK_DIM) twin[i] = w0 + i < n_words ? P[w0 + i] : 0u;
            if (FROM_RUNS) { RUNS_STAGE(tid) }
        }
        if (fresh) { cur_b0 = lds ? b0 : ~0ull; cur_B = lds ? bp.B : 0u; }
        SYNC();
        const uint32_t top = pow2_above(n_split);
        PAR(tid) {
            uint64_t key[TILE_EPT];
            uint32_t bk[TILE_EPT];
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                key[k] = 0;
                if (e < cnt)
                    key[k] = FROM_TEXT ? window64<BITS>(twin, text_base + start + e - w0 * TextTraits<BITS>::CPW)
                           : FROM_RUNS ? in_key[run_source<idx_t>(rsrc, runs_staged, lrow, lsrc, run_ns, run_rowR, run_rowA,
                                                                  run_a, run_b, run_x0 + e)]
                                       : in_key[start + e];
                bk[k] = MAP == MAP_SPLIT ? 0u : bucket_of(bp, key[k]);
            }
            if (MAP == MAP_SPLIT && lds) {                                    // (two loops: one pointer for both tables would
                for (uint32_t st = top >> 1; st >= 1; st >>= 1) {             //  make every probe a flat load)
                    UNROLL
                    for (uint32_t k = 0; k < TILE_EPT; ++k) {
                        const uint32_t idx = bk[k] + st - 1;
                        const uint64_t mk = stab[lds_swz(idx < n_split ? idx : n_split - 1)];
                        if (idx < n_split && mk < key[k]) bk[k] += st;
                    }
                }
            } else if (MAP == MAP_SPLIT) {                                    // more knots than LDS holds: search in HBM
                for (uint32_t st = top >> 1; st >= 1; st >>= 1) {
                    UNROLL
                    for (uint32_t k = 0; k < TILE_EPT; ++k) {
                        const uint32_t idx = bk[k] + st - 1;
                        const uint64_t mk = tab[idx < n_split ? idx : n_split - 1];
                        if (idx < n_split && mk < key[k]) bk[k] += st;
                    }
                }
            }
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    if (lds) FETCH_ADD_U32(&hist[bk[k]], 1u);
                    else ATOMIC_ADD_U64(&count[b0 + bk[k]], 1ull);
                    if (MAP == MAP_SPLIT && bid && lds) bid[start + e] = (uint16_t)bk[k];
                }
            }
        }
        SYNC();                                        // the histogram is complete; the staging arrays are free
    }
    if (cur_B) {
        PAR(tid) {
            for (uint32_t i = tid; i < cur_B; i += K_BLOCK_DIM)
                if (hist[i]) ATOMIC_ADD_U64(&count[cur_b0 + i], (uint64_t)hist[i]);
        }
        SYNC();
    }
    }
}

// sub_start[NB+1] = exclusive scan of count (absolute element offsets of the buckets);
// cursor[NB] starts at 0.  Per tile: LDS histogram of its elements' buckets; one global
// cursor bump per (tile, non-empty bucket); the tile is then re-ordered by bucket in LDS so
// that consecutive lanes write consecutive slots (a bucket receives a run of consecutive
// elements from every tile instead of 64 scattered 8-byte stores per wave instruction).
// KT = uint32_t (SRC_ARRAYS, MAP_LINEAR only): the elements carry 32-bit keys in and out, the map lives in their space.
// SPILL (slots only; the speculative split by knots): what does not fit a slot is not dropped but appended to a stream --
// every (tile, bucket) run that reaches beyond its slot reserves its length there (one global atomic, only then).  An
// entry is (key, index, bucket, position inside the bucket): the cursor's values are positions inside the bucket for ALL its
// elements, so once the sizes are known (the cursors) and scanned, spill_gather_kernel copies the slot's part of every bucket
// that outgrew its slot to the bucket's place in the compact array and spill_place_kernel adds the stream's entries: no
// count pass, and no second scatter for a text whose buckets mostly fit (quantile knots: all but 1-2 %, + the repeats).
template <typename idx_t> struct Spill {
    uint64_t* count = nullptr;     // entries reserved so far (beyond cap: the caller redoes the split with the count pass)
    uint64_t cap = 0;
    uint32_t chunk = 0;            // != 0: tile b of the launch owns entries [b * chunk, (b + 1) * chunk) -- its first runs go there
                                   //   without a trip to `count` (which then starts behind the chunks; unused entries: bucket ~0)
    uint64_t base = 0;             // one past the last slot position: an output offset >= base names entry (offset - base)
    uint64_t* key = nullptr;
    idx_t* sa = nullptr;
    uint32_t* bucket = nullptr;    // bucket of the launch (~0: no entry)
    idx_t* rel = nullptr;          // position inside the bucket
};
constexpr uint32_t SPILL_LONG = 64;                   // runs of this length and more: the workgroup writes (bucket, position) together
constexpr uint32_t SPILL_LONG_CAP = TILE_E / SPILL_LONG;
constexpr uint32_t SPILL_CHUNK = TILE_E / 128 ? TILE_E / 128 : 1;   // a tile of a genome-like text puts 7 elements on the stream, mean

template <typename idx_t, int BITS, int SRC, int MAP, typename KT = uint64_t, bool SPILL = false>
GLOBAL_FN LAUNCH_BOUNDS(TILE_NT) bucket_scatter_kernel(KCTX SegDesc sd, const uint32_t* __restrict__ P, uint64_t n_words,
                                                       uint64_t text_base, const KT* __restrict__ in_key,
                                                       const idx_t* __restrict__ in_sa, RunSrc<idx_t> rsrc,
                                                       const BucketParams* __restrict__ bps, const uint64_t* __restrict__ bstart,
                                                       const uint64_t* __restrict__ sub_start, uint64_t slot_cap,
                                                       idx_t* __restrict__ cursor,
                                                       KT* __restrict__ out_key, idx_t* __restrict__ out_sa,
                                                       const BucketParams* __restrict__ fbps, const uint32_t* __restrict__ gfirst,
                                                       const uint64_t* __restrict__ split, const uint16_t* __restrict__ split_lut,
                                                       const uint32_t* __restrict__ split_span, uint32_t sub, uint32_t split_stride,
                                                       uint32_t in_sub, const uint16_t* __restrict__ bid, Spill<idx_t> spill)
{
    // sub > 1 (slots only; level A of the direct path): bucket i owns `sub` slots, one per sub-stream; the tiles of a
    // launch are dealt to the sub-streams round-robin by block index, i.e. (observed dispatch order, MI355X_MICROARCH
    // "Workgroup dispatch") one sub-stream per XCD: the short runs that consecutive tiles append to a stream then
    // complete each other's cache lines in ONE L2 instead of leaving partial lines in eight.  Any placement is correct.
    constexpr bool GROUPED = MAP == MAP_GROUPED;
    // gfirst != null: equalised split -- the key's FINE bucket (map fbps[g]) is looked up in the segment's group table
    // (bucket_group_kernel) to get its bucket slot.
    // slot_cap != 0 ("speculative" split, no count pass): bucket i of the launch owns the fixed slot
    // [i * slot_cap, (i + 1) * slot_cap) of the output; cursor[] ends as the exact bucket sizes, and
    // elements that do not fit their slot are dropped -- the caller sees a size > slot_cap and
    // redoes the split with the count pass.  slot_cap == 0: bucket i starts at sub_start[i].
    const bool spec = slot_cap != 0;
    const idx_t NO_SLOT = (idx_t)~(idx_t)0;
#if CAPS_SCATTER_SWZ
    const uint32_t v = sub > 1 ? K_BLOCK_IDX                             // sub-streams: tile -> stream by block index
                               : (uint32_t)xcd_swizzle(K_BLOCK_IDX, K_GRID_DIM);   // tiles of one segment on one XCD: the partial
                                                                         // lines of neighbouring runs meet in its L2
#else
    const uint32_t v = K_BLOCK_IDX;
#endif
    const uint32_t sx = sub > 1 ? K_BLOCK_IDX % sub : 0u;
    if (v >= sd.tile_off[sd.G]) return;
    const uint32_t b = v;
    const uint32_t g = sd.tile_rec[b].g;
    const TileInfo t = tile_info(sd, b);
    const uint64_t start = t.s0 + (uint64_t)t.tl * TILE_E;
    const uint32_t cnt = (uint32_t)(t.s1 - start < TILE_E ? t.s1 - start : TILE_E);
    const BucketParams bp = bps[g];
    const uint64_t b0 = bstart[g];
    const bool grouped = GROUPED && bp.B > 1 && bp.B <= BUCKET_LDS;      // GROUPED: a build of its own, the plain scatter is hot
    const BucketParams fbp = GROUPED ? fbps[g] : bp;
    SHARED_ARRAY(uint32_t, hist, TILE_BINS + 1);          // counts -> exclusive prefix inside the tile
    SHARED_ARRAY(idx_t, obase, TILE_BINS);                // global slot of the tile's first element of bucket i, minus its prefix
                                                          // (idx_t: 72 KiB of LDS at 32-bit indices -> two workgroups per CU)
    static_assert(sizeof(KT) == 8 || (SRC == SRC_ARRAYS && MAP == MAP_LINEAR), "32-bit keys: arrays in, linear map");
    SHARED_ARRAY(KT, skey, TILE_E);
    SHARED_ARRAY(idx_t, ssa, TILE_E);
    SHARED_ARRAY(uint16_t, sbk, TILE_E);
    // SPILL: the long runs of the tile that went to the stream (at, cursor value, length, bucket)
    SHARED_ARRAY(uint64_t, lr_at, SPILL ? SPILL_LONG_CAP : 1);
    SHARED_ARRAY(idx_t, lr_old, SPILL ? SPILL_LONG_CAP : 1);
    SHARED_ARRAY(uint32_t, lr_c, SPILL ? SPILL_LONG_CAP : 1);
    SHARED_ARRAY(uint32_t, lr_b, SPILL ? SPILL_LONG_CAP : 1);
    SHARED_ARRAY(uint32_t, nlong, 2);     // [1]: entries of the tile's own chunk taken so far
    static_assert(!SPILL || (MAP == MAP_SPLIT && sizeof(KT) == 8), "the spill stream: splits by knots, 64-bit keys");
    TL_DECL(KT, rk, TILE_EPT);
    TL_DECL(idx_t, rs, TILE_EPT);
    TL_DECL(uint32_t, rb, TILE_EPT);      // bucket
    TL_DECL(idx_t, rr, TILE_EPT);         // rank inside (tile, bucket) or inside the bucket (idx_t: a bucket of a
                                          // degenerate text can hold more than 2^32 suffixes at 64-bit indices)
    constexpr bool FROM_TEXT = SRC == SRC_TEXT, FROM_RUNS = SRC == SRC_RUNS;
    const bool lds = bp.B > 1 && bp.B <= BUCKET_LDS;
    // FROM_TEXT: the tile's slice of the packed text is staged in the (not yet used) key staging array;
    // FROM_RUNS: the runs the tile spans are staged in the (not yet used) key and index staging arrays
    // (64-bit keys only: the staging arrays are large enough for these guests)
    uint32_t* twin = reinterpret_cast<uint32_t*>(skey);
    uint64_t* lsrc = reinterpret_cast<uint64_t*>(skey);
    idx_t* lrow = ssa;
    uint16_t* gtab = sbk;                                  // fine bucket -> bucket slot of the segment (sbk is written after its last use)
    static_assert(TILE_E >= BUCKET_LDS && BUCKET_LDS <= 65536, "group table fits the bucket-id staging array");
    const uint64_t w0 = text_win_base<BITS>(text_base + start);
    // MAP_SPLIT: the splitter table sits in the key staging array too, behind the text window
    uint64_t* stab = reinterpret_cast<uint64_t*>(skey) + TILE_E / 4;
    uint16_t* slut = reinterpret_cast<uint16_t*>(ssa);    // ... and its LUT in the index staging array
    static_assert(TEXT_WIN * sizeof(uint32_t) <= (TILE_E / 4) * sizeof(uint64_t) && TILE_E / 4 + BUCKET_LDS <= TILE_E,
                  "text window + splitter table fit the key staging array");
    static_assert((SPLIT_LUT_CELLS + 1) * sizeof(uint16_t) <= TILE_E * sizeof(idx_t), "splitter LUT fits the index staging array");
    const uint32_t n_split = MAP == MAP_SPLIT && bp.B > 1 ? bp.B - 1 : 0u;
    const uint64_t* tab = MAP == MAP_SPLIT ? split + (uint64_t)(g / in_sub) * split_stride : nullptr;
    RUNS_TILE_SETUP
    if (lds || FROM_TEXT || FROM_RUNS) {
        PAR(tid) {
            if (lds) for (uint32_t i = tid; i <= TILE_BINS; i += K_BLOCK_DIM) hist[i] = 0;
            if (SPILL && tid < 2) nlong[tid] = 0;
            if (MAP == MAP_SPLIT && lds && !bid) {
                for (uint32_t i = tid; i < n_split; i += K_BLOCK_DIM) stab[lds_swz(i)] = tab[i];
                if (split_lut) for (uint32_t i = tid; i <= SPLIT_LUT_CELLS; i += K_BLOCK_DIM) slut[i] = split_lut[i];
            }
            if (grouped)                                   // slot i owns the fine buckets [gfirst[i], gfirst[i + 1]) (F for unused slots)
                for (uint32_t i = tid; i < bp.B; i += K_BLOCK_DIM) {
                    const uint32_t f0 = gfirst[b0 + i], f1 = i + 1 < bp.B ? gfirst[b0 + i + 1] : fbp.B;
                    for (uint32_t f = f0; f < f1 && f < fbp.B; ++f) gtab[f] = (uint16_t)i;
                }
            if (FROM_TEXT)
                for (uint32_t i = tid; i < TEXT_WIN; i += K_BLOCK_DIM) twin[i] = w0 + i < n_words ? P[w0 + i] : 0u;
            if (FROM_RUNS) { RUNS_STAGE(tid) }
        }
        SYNC();
    }
    if (MAP == MAP_SPLIT && lds && bid) {
        // the count pass has left every element's bucket (bucket_count_kernel): no table, no search
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    const uint32_t bk = bid[start + e];
                    TL(rk, tid, k) = in_key[start + e];
                    TL(rs, tid, k) = in_sa[start + e];
                    TL(rb, tid, k) = bk;
                    TL(rr, tid, k) = FETCH_ADD_U32(&hist[bk], 1u);
                }
            }
        }
    } else if (MAP == MAP_SPLIT && lds) {
        // keys first, then the TILE_EPT table searches of a thread in lockstep (independent LDS reads in flight together)
        // depth of the search: the candidates per LUT cell (block-uniform), or all of the table without a LUT
        const uint32_t top = pow2_above(split_lut ? split_span[0] : n_split);
        TL_DECL(uint32_t, rh, TILE_EPT);
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                const uint64_t src = start + e;
                uint64_t key = 0;
                idx_t sa = 0;
                if (e < cnt) {
                    key = FROM_TEXT ? window64<BITS>(twin, text_base + start + e - w0 * TextTraits<BITS>::CPW) : STREAM_LOAD(&in_key[src]);
                    sa = FROM_TEXT ? (idx_t)(text_base + start + e) : STREAM_LOAD(&in_sa[src]);
                }
                const uint32_t cell = (uint32_t)(key >> (64 - SPLIT_LUT_BITS));
                TL(rk, tid, k) = key;
                TL(rs, tid, k) = sa;
                TL(rb, tid, k) = split_lut ? slut[cell] : 0u;
                TL(rh, tid, k) = split_lut ? slut[cell + 1] : n_split;
            }
            for (uint32_t s = top >> 1; s >= 1; s >>= 1) {
                UNROLL
                for (uint32_t k = 0; k < TILE_EPT; ++k) {
                    const uint32_t idx = TL(rb, tid, k) + s - 1;
                    const uint64_t mk = stab[lds_swz(idx < n_split ? idx : n_split - 1)];
                    if (idx < TL(rh, tid, k) && mk < TL(rk, tid, k)) TL(rb, tid, k) += s;
                }
            }
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) TL(rr, tid, k) = FETCH_ADD_U32(&hist[TL(rb, tid, k)], 1u);
            }
        }
    } else {
    PAR(tid) {
        UNROLL
        for (uint32_t k = 0; k < TILE_EPT; ++k) {
            const uint32_t e = tid + k * TILE_NT;
            if (e < cnt) {
                const uint64_t src = FROM_RUNS ? run_source<idx_t>(rsrc, runs_staged, lrow, lsrc, run_ns, run_rowR, run_rowA, run_a,
                                                                   run_b, run_x0 + e)
                                               : start + e;
                const uint64_t key = FROM_TEXT ? window64<BITS>(twin, text_base + start + e - w0 * TextTraits<BITS>::CPW)
                                               : (uint64_t)STREAM_LOAD(&in_key[src]);
                const idx_t sa = FROM_TEXT ? (idx_t)(text_base + start + e) : STREAM_LOAD(&in_sa[src]);
                uint32_t bk = 0;
                idx_t r;
                if (bp.B == 1) r = e;                                        // identity: the segment is its own bucket
                else {
                    if (MAP == MAP_SPLIT) {                                  // too many splitters for the LDS table: search in HBM
                        uint32_t a = 0, z = bp.B - 1;
                        while (a < z) { const uint32_t mid = (a + z) >> 1; if (tab[mid] < key) a = mid + 1; else z = mid; }
                        bk = a;
                    } else {
                        bk = bucket_of(fbp, key);
                        if (grouped) bk = gtab[bk];
                    }
                    r = lds ? FETCH_ADD_U32(&hist[bk], 1u) : caps_fetch_add(&cursor[(b0 + bk) * (spec ? sub : 1u) + sx], (idx_t)1);
                }
                TL(rk, tid, k) = key;
                TL(rs, tid, k) = sa;
                TL(rb, tid, k) = bk;
                TL(rr, tid, k) = r;
            }
        }
    }
    }
    if (!lds) {                                            // one bucket, or too many for the LDS histogram
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t e = tid + k * TILE_NT;
                if (e < cnt) {
                    // one bucket: slot = same offset inside the segment (64-bit: such a segment can be long)
                    const uint64_t off = (bp.B == 1 ? (start - t.s0) : 0) + TL(rr, tid, k);
                    const uint64_t bi = (b0 + TL(rb, tid, k)) * (spec ? sub : 1u) + sx;
                    if (!spec || off < slot_cap) {
                        const uint64_t dst = (spec ? bi * slot_cap : sub_start[bi]) + off;
                        out_key[dst] = TL(rk, tid, k);
                        out_sa[dst] = TL(rs, tid, k);
                    }
                }
            }
            if (spec && bp.B == 1 && tid == 0) caps_fetch_add(&cursor[b0 * sub + sx], (idx_t)cnt);    // the size of a one-bucket segment
        }
        return;
    }
    SYNC();
    // One global cursor bump per (tile, non-empty bucket).  The returning atomics are only ISSUED here; their results are
    // needed for the output addresses alone, so they travel while the tile is scanned and re-ordered in LDS (barriers of
    // LDS scope in between: __syncthreads() would wait for them).
    constexpr uint32_t BPT = (TILE_BINS + TILE_NT - 1) / TILE_NT;          // buckets per thread
    TL_DECL(idx_t, co, BPT);
    TL_DECL(uint32_t, cc, BPT);
    PAR(tid) {
        UNROLL
        for (uint32_t j = 0; j < BPT; ++j) {
            const uint32_t i = tid + j * TILE_NT;
            const uint32_t c = i < bp.B ? hist[i] : 0u;
            TL(cc, tid, j) = c;
            TL(co, tid, j) = 0;
            if (c) TL(co, tid, j) = caps_fetch_add(&cursor[spec ? (b0 + i) * sub + sx : b0 + i], (idx_t)c);
        }
    }
    block_exclusive_scan_bins(KCTX_PASS hist);             // hist[i] = position of bucket i inside the re-ordered tile
    PAR(tid) {
        UNROLL
        for (uint32_t k = 0; k < TILE_EPT; ++k) {
            const uint32_t e = tid + k * TILE_NT;
            if (e < cnt) {
                const uint32_t bk = TL(rb, tid, k);
                const uint32_t q = hist[bk] + (uint32_t)TL(rr, tid, k);
                skey[q] = TL(rk, tid, k);
                ssa[q] = TL(rs, tid, k);
                sbk[q] = (uint16_t)bk;
            }
        }
        UNROLL
        for (uint32_t j = 0; j < BPT; ++j) {
            const uint32_t i = tid + j * TILE_NT;
            if (i < bp.B) {
                const uint32_t c = TL(cc, tid, j);
                const idx_t old = TL(co, tid, j);
                idx_t ob = 0;
                if (c) {
                    const uint64_t ci = spec ? (b0 + i) * sub + sx : b0 + i;
                    if (!spec) ob = (idx_t)(sub_start[b0 + i] + old);
                    else if ((uint64_t)old + c <= slot_cap) ob = (idx_t)(ci * slot_cap + old);
                    else if (SPILL && spill.count) {
                        // the run reaches beyond the slot: all of it goes to the stream (what sits in the slot stays a prefix
                        // of the bucket: the cursor only grows, so every later run of this bucket comes here too)
                        // (room in the tile's own chunk first: the trip to the global counter would sit on every tile's critical
                        // path -- measured at 3e9 genome-like: 20.8 ms with it against 19.4 for the same scatter without a stream)
                        uint64_t at = ~0ull;
                        if (spill.chunk) {
                            const uint32_t lo = FETCH_ADD_U32(&nlong[1], c);
                            if ((uint64_t)lo + c <= spill.chunk) at = (uint64_t)b * spill.chunk + lo;
                        }
                        if (at == ~0ull) at = FETCH_ADD_U64(spill.count, (uint64_t)c);
                        ob = NO_SLOT;
                        if (at + c <= spill.cap) {
                            ob = (idx_t)(spill.base + at);
                            if (c < SPILL_LONG) {
                                for (uint32_t j = 0; j < c; ++j) { spill.bucket[at + j] = (uint32_t)(b0 + i); spill.rel[at + j] = (idx_t)(old + j); }
                            } else {
                                const uint32_t q = FETCH_ADD_U32(&nlong[0], 1u);      // (< SPILL_LONG_CAP: the runs of a tile add up to TILE_E)
                                lr_at[q] = at;
                                lr_old[q] = old;
                                lr_c[q] = c;
                                lr_b[q] = (uint32_t)(b0 + i);
                            }
                        }
                    } else ob = NO_SLOT;
                }
                obase[i] = ob;
            }
        }
    }
    SYNC_LDS();
    PAR(tid) {
        UNROLL
        for (uint32_t k = 0; k < TILE_EPT; ++k) {
            const uint32_t q = tid + k * TILE_NT;
            if (q < cnt) {
                const uint32_t bk = sbk[q];
                const idx_t ob = obase[bk];
                if (SPILL && ob != NO_SLOT && (uint64_t)ob >= spill.base) {
                    const uint64_t dst = (uint64_t)ob - spill.base + (q - hist[bk]);
                    spill.key[dst] = skey[q];
                    spill.sa[dst] = ssa[q];
                } else if (ob != NO_SLOT) {
                    const uint64_t dst = (uint64_t)ob + (q - hist[bk]);
                    STREAM_STORE2(&out_key[dst], skey[q]);
                    STREAM_STORE2(&out_sa[dst], ssa[q]);
                }
            }
        }
        if (SPILL) {
            const uint32_t nl = nlong[0];
            for (uint32_t r = 0; r < nl && r < SPILL_LONG_CAP; ++r) {
                const uint64_t at = lr_at[r];
                const idx_t old = lr_old[r];
                const uint32_t c = lr_c[r], bb = lr_b[r];
                for (uint32_t j = tid; j < c; j += K_BLOCK_DIM) { spill.bucket[at + j] = bb; spill.rel[at + j] = (idx_t)(old + j); }
            }
        }
    }
}

// After a scatter with a spill stream: bucket b of the launch has seg_start[b + 1] - seg_start[b] elements (its cursor); where
// that is more than the slot holds -- or the bucket sits out the tile sort (skip: the letter-run buckets, sorted later from the
// compact array) -- the slot's part goes to the bucket's place in the compact array.  The tail of the slot's part of a bucket
// that outgrew the slot may be positions of runs that went to the stream (never written): spill_place_kernel, next on the
// stream, overwrites exactly those.
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) spill_gather_kernel(KCTX const uint64_t* __restrict__ seg_start, uint32_t NB, uint64_t slot_cap,
                                                 const uint8_t* __restrict__ skip, const uint64_t* __restrict__ slot_key,
                                                 const idx_t* __restrict__ slot_sa, uint64_t* __restrict__ dst_key,
                                                 idx_t* __restrict__ dst_sa)
{
    // a workgroup looks at 256 consecutive buckets at once (one bucket per thread: a walk bucket by bucket was one memory latency
    // per bucket for the 98 % that have nothing to copy -- 0.6 ms at 977,000 buckets), lists the few that do, copies those together
    SHARED_ARRAY(uint32_t, lst, 256);
    SHARED_ARRAY(uint32_t, ln, 1);
    for (uint64_t base = (uint64_t)K_BLOCK_IDX * 256u; base < NB; base += (uint64_t)K_GRID_DIM * 256u) {
        PAR(tid) { if (tid == 0) ln[0] = 0; }
        SYNC();
        PAR(tid) {
            const uint64_t b = base + tid;
            if (b < NB) {
                const uint64_t len = seg_start[b + 1] - seg_start[b];
                if (len > slot_cap || (len && skip && skip[b])) lst[FETCH_ADD_U32(&ln[0], 1u)] = (uint32_t)b;
            }
        }
        SYNC();
        const uint32_t m = ln[0];                                           // block-uniform
        for (uint32_t q = 0; q < m; ++q) {
            const uint32_t b = lst[q];
            const uint64_t s0 = seg_start[b], len = seg_start[b + 1] - s0;
            const uint64_t cnt = len < slot_cap ? len : slot_cap;
            PAR(tid) {
                for (uint64_t i = tid; i < cnt; i += K_BLOCK_DIM) {
                    dst_key[s0 + i] = slot_key[(uint64_t)b * slot_cap + i];
                    dst_sa[s0 + i] = slot_sa[(uint64_t)b * slot_cap + i];
                }
            }
        }
        SYNC();
    }
}

template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) spill_place_kernel(KCTX Spill<idx_t> sp, uint64_t cnt, const uint64_t* __restrict__ seg_start,
                                                uint64_t* __restrict__ dst_key, idx_t* __restrict__ dst_sa)
{
    PAR(tid) {
        for (uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; i < cnt; i += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
            const uint32_t b = sp.bucket[i];
            if (b == ~0u) continue;
            const uint64_t d = seg_start[b] + (uint64_t)sp.rel[i];
            dst_key[d] = sp.key[i];
            dst_sa[d] = sp.sa[i];
        }
    }
}

// ---- level A of the direct path: the text -> groups, in tiles of GA_E consecutive positions -------------------------
// A dedicated kernel rather than bucket_scatter_kernel<SRC_TEXT, MAP_SPLIT> (kept: it is the same computation on
// TILE_E positions and serves as its cross-check): with ~1000 groups, a 4096-element tile spends as much on its groups
// (histogram, one cursor bump each, scan) as on its elements, and appends runs of 4 elements.  Here a workgroup takes
// GA_E = 4 x TILE_E positions.  Nothing but the tile's slice of the packed text is staged: a key is cut from it when the
// element is classified and again when it is written, so the per-element state between the two is ONE register
// (group << 14 | position in the re-ordered tile), and the re-ordered tile goes out in TILE_E-element chunks through a
// 16 KB permutation buffer -- 60 KB of LDS at 2-bit codes, two workgroups per CU.
//   cursor[sx * K1 + g]  elements of group g appended so far to sub-stream sx (= block index % sub, see
//                        bucket_scatter_kernel); ends as the stream's size -- larger than slot_cap = the stream overflowed,
//                        its surplus was dropped and the caller must not use the result
//   stream (g, sx) owns [(g * sub + sx) * slot_cap, +slot_cap) of out_key / out_sa
#ifndef CAPS_GA_TILES
#define CAPS_GA_TILES 4
#endif
constexpr uint32_t GA_TILES = CAPS_GA_TILES;
constexpr uint32_t GA_E = GA_TILES * TILE_E;
constexpr uint32_t GA_EPT = GA_E / TILE_NT;
static_assert(GA_E <= (1u << 14) && BUCKET_LDS <= (1u << 11), "group << 14 | position fits a register");

// KT = uint32_t: the elements leave with 32-bit keys, key32_of(key, gshift[group]) (text.h).
template <typename idx_t, int BITS, typename KT = uint64_t>
GLOBAL_FN LAUNCH_BOUNDS2(TILE_NT, TILE_WAVES_PER_SIMD) group_scatter_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n_words,
                                                      uint64_t text_base, uint64_t len, const uint64_t* __restrict__ split, uint32_t K1,
                                                      const uint16_t* __restrict__ split_lut, const uint32_t* __restrict__ split_span,
                                                      uint32_t sub, uint64_t slot_cap, idx_t* __restrict__ cursor,
                                                      KT* __restrict__ out_key, idx_t* __restrict__ out_sa,
                                                      uint32_t tile_first, uint32_t tile_stride,
                                                      const uint64_t* __restrict__ region_start, const uint64_t* __restrict__ region_cap,
                                                      const uint8_t* __restrict__ gshift, uint32_t own_lo, uint32_t own_hi)
{
    // own_lo .. own_hi: only the elements of these groups are kept (counted, re-ordered, written); stream (g, sx) of a kept group
    // owns [((g - own_lo) * sub + sx) * slot_cap, + slot_cap).  A rank of a sharded build scatters the WHOLE text and keeps the
    // groups it owns: no element ever crosses a link (shard.h); one GPU keeps them all (own_lo = 0, own_hi = K1).
    constexpr bool K32 = sizeof(KT) == 4;
    constexpr uint32_t DROP = ~0u;
    SHARED_ARRAY(uint8_t, scs, K32 ? BUCKET_LDS : 1);
    // region_start != null: stream s = g * sub + sx owns [region_start[s], + region_cap[s]) instead of the uniform
    // [s * slot_cap, + slot_cap) -- groups that share a frequent key get the room of all of them (group_caps_kernel).
    // Workgroup b takes tile tile_first + b * tile_stride of [text_base, text_base + len): a rank of a sharded build takes
    // every world-th tile (its share of the text is then a fine-grained interleave, balanced whatever the text's composition
    // does along its length); one GPU: tile_first = 0, tile_stride = 1.
    constexpr uint32_t CPW = TextTraits<BITS>::CPW;
    constexpr uint32_t WIN = GA_E / CPW + 8;                   // words covering GA_E positions + one key
    const idx_t NO_SLOT = (idx_t)~(idx_t)0;
    SHARED_ARRAY(uint32_t, twin, WIN);
    SHARED_ARRAY(uint64_t, stab, BUCKET_LDS);
    SHARED_ARRAY(uint16_t, slut, SPLIT_LUT_CELLS + 2);
    SHARED_ARRAY(uint32_t, hist, TILE_BINS + 1);
    SHARED_ARRAY(idx_t, obase, TILE_BINS);
    SHARED_ARRAY(uint32_t, perm, TILE_E);
    TL_DECL(uint32_t, pk, GA_EPT);
    const uint64_t start = ((uint64_t)tile_first + (uint64_t)K_BLOCK_IDX * tile_stride) * GA_E;
    if (start >= len) return;
    const uint32_t cnt = (uint32_t)(len - start < GA_E ? len - start : GA_E);
    const uint32_t sx = sub > 1 ? K_BLOCK_IDX % sub : 0u;
    const uint32_t n_split = K1 - 1;
    const uint64_t pos0 = text_base + start;
    const uint64_t w0 = pos0 / CPW;
    const uint32_t top = pow2_above(split_span[0]);
    PAR(tid) {
        for (uint32_t i = tid; i <= TILE_BINS; i += K_BLOCK_DIM) hist[i] = 0;
        for (uint32_t i = tid; i < WIN; i += K_BLOCK_DIM) twin[i] = w0 + i < n_words ? P[w0 + i] : 0u;
        for (uint32_t i = tid; i < n_split; i += K_BLOCK_DIM) stab[GA_SWZ(i)] = split[i];
        for (uint32_t i = tid; i <= SPLIT_LUT_CELLS; i += K_BLOCK_DIM) slut[i] = split_lut[i];
        if (K32) for (uint32_t i = tid; i < K1; i += K_BLOCK_DIM) scs[i] = gshift[i];
    }
    SYNC();
    // ---- classify: group = #{splitters < key} (LUT cell -> a few candidates -> branch-free search), rank inside (tile, group)
    // 2-bit codes: a thread classifies GA_EPT CONSECUTIVE positions (when a tile starts on a word boundary: always, GA_E is a
    // multiple of the 16 bases of a word) -- their keys are one 96-bit window of the staged text shifted along, three LDS reads
    // for all of them instead of three each; byte codes keep one position per thread and round
    const bool RUN_OF_POS = BITS == 2 && GA_EPT == CPW && pos0 % CPW == 0;      // block-uniform
    PAR(tid) {
        uint64_t wA = 0;
        uint32_t wC = 0;
        if (RUN_OF_POS) {
            const uint32_t wi = (uint32_t)(pos0 / CPW - w0) + tid;          // the word of my first position
            wA = ((uint64_t)twin[wi] << 32) | twin[wi + 1];
            wC = twin[wi + 2];
        }
        UNROLL
        for (uint32_t kk = 0; kk < GA_EPT; kk += 4) {
            uint64_t key[4];
            uint32_t lo[4], hi[4];
            UNROLL
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t e = RUN_OF_POS ? tid * GA_EPT + kk + j : tid + (kk + j) * TILE_NT;
                const uint32_t sh = BITS * (kk + j);                        // < 32
                key[j] = e >= cnt ? 0 : !RUN_OF_POS ? window64<BITS>(twin, pos0 + e - w0 * CPW)
                                                    : sh ? (wA << sh) | (uint64_t)(wC >> (32u - sh)) : wA;
                const uint32_t cell = (uint32_t)(key[j] >> (64 - SPLIT_LUT_BITS));
                lo[j] = slut[cell];
                hi[j] = slut[cell + 1];
            }
            for (uint32_t st = top >> 1; st >= 1; st >>= 1) {
                UNROLL
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t idx = lo[j] + st - 1;
                    const uint64_t mk = stab[GA_SWZ(idx < n_split ? idx : (n_split ? n_split - 1 : 0))];
                    if (idx < hi[j] && mk < key[j]) lo[j] += st;
                }
            }
            UNROLL
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t e = RUN_OF_POS ? tid * GA_EPT + kk + j : tid + (kk + j) * TILE_NT;
                uint32_t v = DROP;
                if (e < cnt && lo[j] >= own_lo && lo[j] < own_hi) v = (lo[j] << 14) | FETCH_ADD_U32(&hist[lo[j]], 1u);
                TL(pk, tid, kk + j) = v;
            }
        }
    }
    SYNC();
    // ---- one cursor bump per (tile, non-empty group), issued here and consumed after the scan (the returning atomics
    //      travel meanwhile: LDS-scope barriers in between); hist -> first position of every group in the re-ordered tile
    constexpr uint32_t GPT = (TILE_BINS + TILE_NT - 1) / TILE_NT;          // groups per thread
    TL_DECL(idx_t, co, GPT);
    TL_DECL(uint32_t, cc, GPT);
    PAR(tid) {
        UNROLL
        for (uint32_t j = 0; j < GPT; ++j) {
            const uint32_t i = tid + j * TILE_NT;
            const uint32_t c = i < K1 ? hist[i] : 0u;
            TL(cc, tid, j) = c;
            TL(co, tid, j) = 0;
            if (c) TL(co, tid, j) = caps_fetch_add(&cursor[(uint64_t)sx * K1 + i], (idx_t)c);
        }
    }
    block_exclusive_scan_bins(KCTX_PASS hist);
    PAR(tid) {
        UNROLL
        for (uint32_t k = 0; k < GA_EPT; ++k) {
            const uint32_t v = TL(pk, tid, k), g = v >> 14;
            if (v != DROP) TL(pk, tid, k) = (g << 14) | (hist[g] + (v & 0x3FFFu));   // group << 14 | position q in the re-ordered tile
        }
        UNROLL
        for (uint32_t j = 0; j < GPT; ++j) {
            const uint32_t i = tid + j * TILE_NT;
            if (i < K1) {
                const uint32_t c = TL(cc, tid, j);
                idx_t ob = 0;
                if (c) {
                    const idx_t old = TL(co, tid, j);
                    const uint64_t st = (uint64_t)(i - own_lo) * sub + sx;             // (c != 0: a kept group)
                    // (region_start / region_cap point at the first kept stream's entries; the output starts at its region)
                    const uint64_t r0 = region_start ? region_start[st] - region_start[0] : st * slot_cap, rc = region_start ? region_cap[st] : slot_cap;
                    ob = (uint64_t)old + c <= rc ? (idx_t)(r0 + old) : NO_SLOT;
                }
                obase[i] = ob;
            }
        }
    }
    // ---- write the re-ordered tile (its kept elements), TILE_E positions at a time: perm[q] = group << 14 | element
    const uint32_t kept = hist[TILE_BINS];                              // block-uniform (the scan's total)
    for (uint32_t c0 = 0; c0 < kept; c0 += TILE_E) {
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < GA_EPT; ++k) {
                const uint32_t e = RUN_OF_POS ? tid * GA_EPT + k : tid + k * TILE_NT;
                const uint32_t v = TL(pk, tid, k), q = v & 0x3FFFu;
                if (v != DROP && q - c0 < TILE_E) perm[q - c0] = (v & ~0x3FFFu) | e;
            }
        }
        SYNC_LDS();
        PAR(tid) {
            UNROLL
            for (uint32_t k = 0; k < TILE_EPT; ++k) {
                const uint32_t q = c0 + tid + k * TILE_NT;
                if (q < kept) {
                    const uint32_t v = perm[q - c0], g = v >> 14, e = v & 0x3FFFu;
                    const idx_t ob = obase[g];
                    if (ob != NO_SLOT) {
                        const uint64_t dst = (uint64_t)ob + (q - hist[g]);
                        const uint64_t key = window64<BITS>(twin, pos0 + e - w0 * CPW);
                        STREAM_STORE2(&out_key[dst], K32 ? (KT)key32_of(key, scs[g]) : (KT)key);
                        STREAM_STORE2(&out_sa[dst], (idx_t)(pos0 + e));
                    }
                }
            }
        }
        SYNC_LDS();                                  // the chunk's stores need not land before the next chunk is staged
    }
}

// Report of a rank's level A for the other ranks (sharded direct path): out[0 .. n_streams) = stream sizes,
// out[n_streams] = flag word (pivot-key ties), out[n_streams + 1] = size of the largest stream that outgrew its region (0: none).
// region_cap != null (quantile mode): the room of stream (g, sx) is region_cap[(g - own_lo) * sub + sx] instead of cap (K1 = groups,
// the cursors are stream-major: s = sx * K1 + g; streams of groups outside own_lo .. own_hi are empty).
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) stream_report_kernel(KCTX const idx_t* __restrict__ cursor, uint32_t n_streams, uint64_t cap,
                                                  const uint32_t* __restrict__ flag, uint64_t* __restrict__ out,
                                                  const uint64_t* __restrict__ region_cap, uint32_t K1, uint32_t sub, uint32_t own_lo,
                                                  uint32_t own_hi)
{
    PAR(tid) {
        const uint32_t s = K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (s < n_streams) {
            const uint64_t z = (uint64_t)cursor[s];
            out[s] = z;
            uint64_t room = cap;
            if (region_cap) {
                const uint32_t g = s % K1, sx = s / K1;
                room = g >= own_lo && g < own_hi ? region_cap[(uint64_t)(g - own_lo) * sub + sx] : 0;
            }
            if (z > room) ATOMIC_MAX_U64(&out[n_streams + 1], z);
        }
        if (s == n_streams) out[n_streams] = flag[0];
    }
}

// Bucket sizes of a speculative split: cursor (idx_t) -> count (u64), the input of the scan.
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) widen_kernel(KCTX const idx_t* __restrict__ in, uint64_t cnt, uint64_t* __restrict__ out)
{
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < cnt) out[i] = (uint64_t)in[i];
    }
}

// Moves every bucket whose data ended in ping-pong buffer 1 (odd number of merge passes) back
// into buffer 0, so that a parent segment is one sorted array again.
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) unify_kernel(KCTX SegDesc sd, const uint64_t* __restrict__ key1, const idx_t* __restrict__ sa1,
                                          uint64_t* __restrict__ key0, idx_t* __restrict__ sa0)
{
    const uint32_t b = K_BLOCK_IDX;
    if (b >= sd.tile_off[sd.G]) return;
    const TileInfo t = tile_info(sd, b);
    if ((passes_for(t.s1 - t.s0) & 1u) == 0) return;
    const uint64_t start = t.s0 + (uint64_t)t.tl * TILE_E;
    const uint32_t cnt = (uint32_t)(t.s1 - start < TILE_E ? t.s1 - start : TILE_E);
    PAR(tid) {
        for (uint32_t e = tid; e < cnt; e += K_BLOCK_DIM) {
            key0[start + e] = key1[start + e];
            sa0[start + e] = sa1[start + e];
        }
    }
}

// ----------------------------------------------------------------------------------
// a6: samples and pivots (reference: sample_pivots/select_pivots, cpp:187-222).
// Unlike the reference's truncated gap (cpp:191: the top of every subarray is never
// sampled), sample k sits at the CENTRE of the k-th of ppp equal slices of the
// sorted subarray: the rank of a pivot among the samples is then an unbiased estimate of
// its rank among all suffixes, and the first/last partitions are not p*gap/2 oversized
// (samples at slice ends made them ~10x the mean at C2).  The output does not depend on
// the pivots.
// ----------------------------------------------------------------------------------
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) sample_kernel(KCTX const uint64_t* __restrict__ seg_start, uint32_t G, uint32_t ppp,
                                           const uint64_t* __restrict__ key, const idx_t* __restrict__ sa,
                                           uint64_t* __restrict__ out_key, idx_t* __restrict__ out_sa)
{
    PAR(tid) {
        const uint64_t t = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (t < (uint64_t)G * ppp) {
            const uint32_t g = (uint32_t)(t / ppp), k = (uint32_t)(t % ppp);
            const uint64_t s0 = seg_start[g], len = seg_start[g + 1] - s0;
            const uint64_t at = s0 + ((uint64_t)(2 * k + 1) * len) / (2 * (uint64_t)ppp);
            out_key[t] = key[at];
            out_sa[t] = sa[at];
        }
    }
}

template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) pick_pivots_kernel(KCTX const uint64_t* __restrict__ skey, const idx_t* __restrict__ ssa, uint64_t m,
                                                uint32_t p, uint64_t* __restrict__ pkey, idx_t* __restrict__ psa)
{
    PAR(tid) {
        const uint64_t j = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (j + 1 < p) {
            const uint64_t at = ((j + 1) * m) / p;
            pkey[j] = skey[at];
            psa[j] = ssa[at];
        }
    }
}

// ---- direct path (pipeline.h, Builder::run_direct): pivots straight from the text ----------------
// The reference samples its SORTED subarrays (cpp:187-195) because phase 1 has sorted them anyway.  The
// direct path draws the same number of samples (p * ppp) from the text itself: sample t is one position of
// the t-th of m equal strata of [0, n) (jittered by a hash of t, so that a periodic text does not alias with
// the stride).  Strata are disjoint, hence the sampled suffixes are distinct.
DEV_INLINE uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// pos0 = first text position of the sampled range (a shard samples its own slice), len = its length, m <= len
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) sample_text_kernel(KCTX const uint32_t* __restrict__ P, uint64_t pos0, uint64_t len, uint64_t m,
                                                uint64_t* __restrict__ out_key, idx_t* __restrict__ out_sa)
{
    PAR(tid) {
        const uint64_t t = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (t < m) {
            const uint64_t q = len / m, r = len % m;                    // floor(t * len / m) = t * q + floor(t * r / m)
            const uint64_t lo = t * q + (t * r) / m, hi = (t + 1) * q + ((t + 1) * r) / m;
            const uint64_t pos = pos0 + lo + mix64(t) % (hi - lo);      // hi - lo >= q >= 1
            out_key[t] = window64<BITS>(P, pos);
            out_sa[t] = (idx_t)pos;
        }
    }
}

// Groups of PG consecutive partitions: gkey[g] = the pivot key that closes group g (g < K1 - 1), i.e. group g holds
// the keys in (gkey[g-1], gkey[g]].  flag[0] |= 1 when two consecutive pivot KEYS are equal (some key value is so
// frequent -- a long repeat, an N-block -- that splitting by key alone cannot balance the groups: the caller then
// takes the samplesort path, whose exact comparator splits such stretches).
GLOBAL_FN LAUNCH_BOUNDS(256) group_keys_kernel(KCTX const uint64_t* __restrict__ pkey, uint32_t p, uint32_t PG, uint32_t K1,
                                               uint64_t* __restrict__ gkey, uint32_t* __restrict__ flag)
{
    // flag[3] = longest run of pivots with one key (how much of the text a single key value covers, in partitions)
    PAR(tid) {
        const uint32_t j = K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (j + 2 < p && pkey[j] >= pkey[j + 1]) {
            flag[0] = 1;                                                // benign race: every writer stores 1
            if (j == 0 || pkey[j - 1] != pkey[j]) {                     // head of a run
                uint32_t len = 2;
                while (j + len < p - 1 && pkey[j + len] == pkey[j]) ++len;
#ifdef CAPS_EMUL
                if (len > flag[3]) flag[3] = len;
#else
                atomicMax(&flag[3], len);
#endif
            }
        }
        if (j + 1 < K1) gkey[j] = pkey[(uint64_t)(j + 1) * PG - 1];
        if (j + 1 == K1) gkey[j] = ~0ull;                               // (as a knot table of K1 buckets: the last one is open above)
    }
}

// 32-bit keys (text.h key32_of): gshift[g] = bits all keys of group g share = the common bit prefix of its smallest
// possible key (the previous group key + 1) and its largest (its own group key), at most 32.
GLOBAL_FN LAUNCH_BOUNDS(256) group_shift_kernel(KCTX const uint64_t* __restrict__ gkey, uint32_t K1, uint8_t* __restrict__ gshift)
{
    PAR(tid) {
        const uint32_t g = K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (g < K1) {
            const uint64_t lo = g ? gkey[g - 1] + 1 : 0, hi = g + 1 < K1 ? gkey[g] : ~0ull;
            const uint64_t x = lo ^ hi;
            const uint32_t c = x ? (uint32_t)caps_clz64(x) : 32u;
            gshift[g] = (uint8_t)(lo > hi ? 0u : c < 32u ? c : 32u);       // (lo > hi: gkey[g-1] was the largest key; the group is empty)
        }
    }
}

// Splitter search accelerator of level A (bucket_scatter_kernel<MAP_SPLIT>): the top SPLIT_LUT_BITS bits of a key select a
// cell; lut[c] = #{splitters < smallest key of cell c} (lut[cells] = n_split), so a key of cell c belongs to group
// lut[c] + #{splitters in [lut[c], lut[c+1]) that are < key}: a search over lut[c+1] - lut[c] candidates instead of all of
// them -- 0 to 2 on pivots that are roughly uniform in the key range, whatever it takes on skewed ones: span[0] = the largest
// such count, the (block-uniform) depth of the branch-free search.

GLOBAL_FN LAUNCH_BOUNDS(256) split_lut_kernel(KCTX const uint64_t* __restrict__ split, uint32_t n_split, uint16_t* __restrict__ lut,
                                              uint32_t* __restrict__ span)
{
    PAR(tid) {
        const uint32_t c = K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (c <= SPLIT_LUT_CELLS) {
            uint32_t lo = 0, hi = n_split;                               // #{splitters < cell_lo(c)}; cell_lo(cells) = 2^64
            if (c < SPLIT_LUT_CELLS) {
                const uint64_t cell_lo = (uint64_t)c << (64 - SPLIT_LUT_BITS);
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (split[mid] < cell_lo) lo = mid + 1; else hi = mid; }
            } else lo = n_split;
            lut[c] = (uint16_t)lo;
            if (c < SPLIT_LUT_CELLS) {
                uint32_t a = lo, z = n_split;                            // lut[c + 1], recomputed (no dependence on another thread)
                if (c + 1 < SPLIT_LUT_CELLS) {
                    const uint64_t next_lo = (uint64_t)(c + 1) << (64 - SPLIT_LUT_BITS);
                    while (a < z) { const uint32_t mid = (a + z) >> 1; if (split[mid] < next_lo) a = mid + 1; else z = mid; }
                } else a = n_split;
#ifdef CAPS_EMUL
                if (a - lo > span[0]) span[0] = a - lo;
#else
                if (a > lo) atomicMax(&span[0], a - lo);
#endif
            }
        }
    }
}

// The groups as segments in fixed-capacity regions: group g = [g * cap, g * cap + size_g) of the level-A output.
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) slot_segments_kernel(KCTX const idx_t* __restrict__ sizes, uint32_t K1, uint32_t sub, uint32_t stream_major,
                                                  uint64_t cap, const uint64_t* __restrict__ region_start,
                                                  const uint64_t* __restrict__ region_cap, uint64_t* __restrict__ seg_start,
                                                  uint64_t* __restrict__ seg_end, uint64_t* __restrict__ total)
{
    // segment s = g * sub + x is sub-stream x of group g; its size is sizes[x * K1 + g] (stream_major: group_scatter_kernel's
    // cursors) or sizes[s] (bucket_scatter_kernel's); its region: [s * cap, + cap), or region_start / region_cap when given
    PAR(tid) {
        const uint32_t G = K1 * sub;
        const uint32_t s = K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (s < G) {
            const uint64_t z = (uint64_t)sizes[stream_major ? (uint64_t)(s % sub) * K1 + s / sub : s];
            const uint64_t r0 = region_start ? region_start[s] : (uint64_t)s * cap, rc = region_start ? region_cap[s] : cap;
            seg_start[s] = r0;
            seg_end[s] = r0 + (z < rc ? z : rc);                       // an overflowed stream is reported through total[1]
            ATOMIC_ADD_U64(total, z);
            if (z > rc) ATOMIC_MAX_U64(total + 1, z);
        }
        if (s == G) seg_start[s] = region_start ? region_start[G - 1] + region_cap[G - 1] : (uint64_t)G * cap;
    }
}

// ---- quantile mode of the direct path (skewed key distributions: every real genome) ------------------------
// skew_probe_kernel: the pivots are quantiles (equal numbers of suffixes between consecutive ones); level B's linear
// bucket map over a group's key range assumes they are also equally SPACED inside the group.  flag[0] |= 1 when some
// partition's share of its group's key range is below SKEW_MIN / (partitions in the group): its buckets would overflow.
constexpr double SKEW_MIN = 0.7;          // uniform keys: 1 +- 4 % (sampling noise of ppp = 700 samples per partition)

GLOBAL_FN LAUNCH_BOUNDS(256) skew_probe_kernel(KCTX const uint64_t* __restrict__ pkey, uint32_t p, uint32_t PG, uint32_t K1,
                                               uint32_t* __restrict__ flag)
{
    PAR(tid) {
        const uint32_t j = K_BLOCK_IDX * K_BLOCK_DIM + tid;           // partition j = keys in (pkey[j-1], pkey[j]]
        if (j < p) {
            const uint32_t g = j / PG, j0 = g * PG, j1 = (j0 + PG < p ? j0 + PG : p);      // its group: partitions [j0, j1)
            if (j1 - j0 >= 2) {
                const double glo = j0 ? (double)pkey[j0 - 1] : 0.0, ghi = j1 < p ? (double)pkey[j1 - 1] : 18446744073709551615.0;
                const double lo = j ? (double)pkey[j - 1] : 0.0, hi = j + 1 < p ? (double)pkey[j] : 18446744073709551615.0;
                if (ghi > glo && (hi - lo) * (double)(j1 - j0) < SKEW_MIN * (ghi - glo)) flag[0] = 1;
            }
        }
    }
}

// knots[k] (k < NB - 1) = the sample at quantile (k + 1) / NB of the sorted samples: bucket k holds the keys in
// (knots[k-1], knots[k]]; the groups are runs of KPG consecutive buckets, gkey[g] = knots[(g + 1) * KPG - 1].
// A FREQUENT key K -- the quantile of two or more consecutive knots: a long repeat, an N-block -- gets a bucket of its own:
// the first knot of its run becomes K - 1, so bucket (K - 1, K] = {K} holds nothing else (run_bucket_mark_kernel finds such
// buckets by knots[k - 1] + 1 == knots[k]) and the keys between the previous knot and K have theirs.
GLOBAL_FN LAUNCH_BOUNDS(256) knots_kernel(KCTX const uint64_t* __restrict__ skey, uint64_t m, uint64_t NB, uint32_t KPG, uint32_t K1,
                                          uint64_t* __restrict__ knots, uint64_t* __restrict__ gkey)
{
    PAR(tid) {
        const uint64_t k = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (k < NB) {
            // the last bucket of the last group ends at the top of the key range
            uint64_t v = k + 1 < NB ? skey[((k + 1) * m) / NB] : ~0ull;
            if (k + 2 < NB) {
                const uint64_t nx = skey[((k + 2) * m) / NB];
                const uint64_t pv = k ? skey[(k * m) / NB] : 0;
                if (nx == v && (k == 0 ? v > 0 : pv + 1 < v)) --v;        // head of a run of equal knots, with room below
            }
            knots[k] = v;
            if ((k + 1) % KPG == 0 && (k + 1) / KPG < K1) gkey[(k + 1) / KPG - 1] = v;
        }
    }
}

// Capacity of the streams of every group in quantile mode.  Every bucket is expected to hold n / NB suffixes -- except
// that all suffixes of a frequent key land in the FIRST of the buckets whose knot is that key (the others, (key, key],
// stay empty).  So group g is expected to hold (KPG - lead + tail) buckets' worth: lead = its leading buckets that repeat
// the previous group's last knot (empty), tail = the buckets of later groups that repeat its own last knot (theirs go to g).
// caps[g * sub + x] = base * (KPG - lead + tail) / KPG + token: the shares add up to base * K1.
GLOBAL_FN LAUNCH_BOUNDS(256) group_caps_kernel(KCTX const uint64_t* __restrict__ knots, uint64_t NB, uint32_t KPG, uint32_t K1,
                                               uint32_t sub, uint64_t base, uint64_t token, uint64_t* __restrict__ caps)
{
    PAR(tid) {
        const uint32_t g = K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (g < K1) {
            const uint64_t k0 = (uint64_t)g * KPG, k1 = k0 + KPG;       // the group's buckets [k0, k1)
            uint64_t lead = 0, tail = 0;
            if (g > 0) while (k0 + lead < k1 && knots[k0 + lead] == knots[k0 - 1]) ++lead;
            if (lead < KPG) while (k1 + tail < NB && knots[k1 + tail] == knots[k1 - 1]) ++tail;
            const uint64_t worth = KPG - lead + tail;
            const uint64_t c = base * worth / KPG + token;              // base < 2^33, worth < 2^22
            for (uint32_t x = 0; x < sub; ++x) caps[(uint64_t)g * sub + x] = c;
        }
    }
}

// Quantile mode's counterpart of bucket_plan_kernel: every parent (group) has exactly KPG buckets; all its `sub`
// sub-streams carry that map, the bucket count is credited to the last one (see bucket_plan_kernel).
GLOBAL_FN LAUNCH_BOUNDS(256) knot_plan_kernel(KCTX uint32_t G, uint32_t sub, uint32_t KPG, BucketParams* __restrict__ bp,
                                              uint64_t* __restrict__ segB)
{
    PAR(tid) {
        const uint32_t g = K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (g < G) {
            BucketParams q = make_bucket_params(0, ~0ull, KPG);     // only B is used by MAP_SPLIT
            bp[g] = q;
            segB[g] = (g % sub == sub - 1 || g == G - 1) ? KPG : 0;
        }
    }
}

// The tile sort's bin map of every bucket in quantile mode: linear over (knots[i-1], knots[i]].
// has_prev: knots[-1] exists (a shard's slice of the knots that does not start at bucket 0): the lower end of bucket 0.
GLOBAL_FN LAUNCH_BOUNDS(256) knot_ranges_kernel(KCTX const uint64_t* __restrict__ knots, uint64_t NB, BucketParams* __restrict__ tile_map,
                                                uint32_t has_prev)
{
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < NB) {
            const bool below = i > 0 || has_prev != 0;
            const uint64_t lo = below ? knots[(int64_t)i - 1] : 0, hi = knots[i];
            tile_map[i] = make_bucket_params(lo < hi ? lo + (below ? 1 : 0) : hi, hi, TILE_BINS);
        }
    }
}

// ---- letter-run buckets (text.h "letter runs") ------------------------------------------------------------------
// run_bucket_mark_kernel: bucket i of a quantile split is a LETTER-RUN BUCKET when it holds one key only (knots[i - 1] + 1 ==
// knots[i]: knots_kernel gives every frequent key such a bucket), that key is one letter repeated, and it is larger than a
// tile -- all the suffixes deep inside the N-blocks of a genome (the CLI maps N to G, src/main.cpp:61-68).  Such a bucket
// sits out the tile sort and the LCP-merge passes (skip[i] = 1: no tiles) and is listed ({bucket, start, end, key} in
// list[1 + 4 j ..], list[0] = how many) for run_rekey_kernel / run_emit_kernel, which order it by (terminator class,
// what is left of the run, text behind the run) -- no suffix of it is compared with another through the run.
constexpr uint32_t RUN_BUCKET_MAX = 64;        // listed buckets (4 letters at 2 bits per char; any more stay ordinary)
GLOBAL_FN LAUNCH_BOUNDS(256) run_bucket_mark_kernel(KCTX const uint64_t* __restrict__ knots, uint64_t NB, uint32_t has_prev, uint32_t bits,
                                                    const uint64_t* __restrict__ seg_start, uint8_t* __restrict__ skip,
                                                    uint64_t* __restrict__ list)
{
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < NB) {
            const uint64_t hi = knots[i], len = seg_start[i + 1] - seg_start[i];
            const bool pure = (i > 0 || has_prev != 0) ? knots[(int64_t)i - 1] + 1 == hi : hi == 0;
            const bool letter = bits == 2 ? is_letter_key<2>(hi) : is_letter_key<8>(hi);
            uint8_t s = 0;
            if (pure && letter && len > TILE_E) {
                const uint64_t j = FETCH_ADD_U64(&list[0], 1);
                if (j < RUN_BUCKET_MAX) {
                    list[1 + 4 * j] = i;
                    list[2 + 4 * j] = seg_start[i];
                    list[3 + 4 * j] = seg_start[i + 1];
                    list[4 + 4 * j] = hi;
                    s = 1;
                }
            }
            skip[i] = s;
        }
    }
}

// key[x] = run_key(sa[x]) for the elements [s0, s1) of a letter-run bucket of key K (in place: the bucket's keys are all K).
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) run_rekey_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, uint64_t s0, uint64_t s1, uint64_t K,
                                              uint64_t* __restrict__ key, const idx_t* __restrict__ sa)
{
    PAR(tid) {
        for (uint64_t x = s0 + (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; x < s1; x += (uint64_t)K_GRID_DIM * K_BLOCK_DIM)
            key[x] = run_key<(int)sizeof(idx_t), BITS>(P, n, (uint64_t)sa[x], K);
    }
}

// The sorted letter-run bucket [s0, s1) (run keys + positions) -> SA and LCP.  LCPs from the run keys (text.h run_pair_lcp);
// the bucket's head and the head of what follows it in the suffix array (at `total`: the end of the array) are compared with
// their neighbours in dSA through the text -- the segment records of a bucket that sat out the sort hold nothing.
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) run_emit_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, uint64_t s0, uint64_t s1, uint64_t total,
                                             const uint64_t* __restrict__ key, const idx_t* __restrict__ sa,
                                             idx_t* __restrict__ dSA, idx_t* __restrict__ dLCP)
{
    PAR(tid) {
        for (uint64_t x = s0 + (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; x <= s1; x += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
            if (x == s1) {                                       // the element behind the bucket: its LCP with the bucket's last
                if (s1 < total) dLCP[s1] = (idx_t)deep_lcp<BITS, true>(P, n, (uint64_t)sa[s1 - 1], (uint64_t)dSA[s1], 0);
                continue;
            }
            const uint64_t b = (uint64_t)sa[x];
            uint64_t l = 0;
            if (x > s0) l = run_pair_lcp<(int)sizeof(idx_t), BITS>(P, n, key[x - 1], (uint64_t)sa[x - 1], key[x], b);
            else if (s0 > 0) l = deep_lcp<BITS, true>(P, n, (uint64_t)dSA[s0 - 1], b, 0);
            dSA[x] = (idx_t)b;
            dLCP[x] = (idx_t)l;
        }
    }
}

// ----------------------------------------------------------------------------------
// a7/a8: locate every pivot in every sorted segment (reference: upper_bound,
// cpp:252-297, locate_pivots, cpp:225-249).  Pm is the reference's `P` matrix:
// Pm[g*(np+2) .. ] = {0, ub(pivot_0), ..., ub(pivot_{np-1}), len_g}.
// One thread per (segment, pivot); exact comparator (no 65,536-char cutoff, cpp:261).
// ----------------------------------------------------------------------------------
template <typename idx_t, int BITS, bool GALLOP = false>
GLOBAL_FN LAUNCH_BOUNDS(256) locate_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n,
                                           const uint64_t* __restrict__ seg_start, uint32_t G,
                                           const uint64_t* __restrict__ key, const idx_t* __restrict__ sa,
                                           const uint64_t* __restrict__ pkey, const idx_t* __restrict__ psa, uint32_t np,
                                           idx_t* __restrict__ Pm)
{
    const uint32_t bpr = (np + K_BLOCK_DIM - 1) / K_BLOCK_DIM;            // blocks per segment row
    const uint64_t total = (uint64_t)G * bpr;                             // logical blocks (~p^2 / 256: may exceed a dispatch)
    for (uint64_t v = K_BLOCK_IDX; v < total; v += K_GRID_DIM) {
    const uint64_t L = K_GRID_DIM == total ? xcd_swizzle(v, total) : v;
    const uint32_t g = (uint32_t)(L / bpr), jb = (uint32_t)(L % bpr);
    const uint64_t s0 = seg_start[g], len = seg_start[g + 1] - s0;
    PAR(tid) {
        const uint32_t j = jb * K_BLOCK_DIM + tid;
        idx_t* row = Pm + (uint64_t)g * (np + 2);
        if (j < np) {
            const uint64_t pk = pkey[j], ps = (uint64_t)psa[j];
            uint64_t lo = 0, hi = len;                                    // first element > pivot
            // The pivots are quantiles of all suffixes, so pivot j sits near rank (j + 1) * len / (np + 1) of any subarray,
            // whatever the key distribution (binomial spread).  Short subarrays (fewer than 8 elements per pivot: C2 has 4)
            // gallop from that guess to a bracket and bisect there: C2's locate 1.48 -> 0.87 ms.  Long ones keep the plain
            // bisection: its upper levels probe the same few cache lines for all 64 pivots of a wave, while every
            // galloping probe of a wave touches 64 lines (C3, 47 elements per pivot: 4.6 -> 8 ms with the gallop).
            if (GALLOP && len > 64 && len < 8 * (uint64_t)np) {        // a build of its own: the plain one keeps its registers
                const uint64_t g0 = ((uint64_t)(j + 1) * len) / ((uint64_t)np + 1);          // < len
                uint64_t step = 32;
                if (suffix_less<BITS>(P, n, pk, ps, key[s0 + g0], (uint64_t)sa[s0 + g0])) {   // element g0 > pivot: answer <= g0
                    hi = g0;
                    while (hi > 0) {
                        const uint64_t probe = hi > step ? hi - step : 0;
                        if (suffix_less<BITS>(P, n, pk, ps, key[s0 + probe], (uint64_t)sa[s0 + probe])) { hi = probe; step *= 2; }
                        else { lo = probe + 1; break; }
                    }
                } else {                                                                      // answer > g0
                    lo = g0 + 1;
                    while (lo < len) {
                        const uint64_t probe = lo + step - 1 < len ? lo + step - 1 : len - 1;
                        if (!suffix_less<BITS>(P, n, pk, ps, key[s0 + probe], (uint64_t)sa[s0 + probe])) { lo = probe + 1; step *= 2; }
                        else { hi = probe; break; }
                    }
                }
            }
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (suffix_less<BITS>(P, n, pk, ps, key[s0 + mid], (uint64_t)sa[s0 + mid])) hi = mid;
                else lo = mid + 1;
            }
            row[j + 1] = (idx_t)lo;
        }
        if (j == 0) { row[0] = 0; row[np + 1] = (idx_t)len; }
    }
    }
}

// ----------------------------------------------------------------------------------
// a9: partition sizes and the "ruler" (reference: partition_sub_subarrays, cpp:300-368).
// Column j of Pm: ruler[g*p + j] = offset of sub-subarray (g, j) inside partition j
// (exclusive scan over g, cpp:340-349), sizes[j] = partition size (cpp:305-316).
// ----------------------------------------------------------------------------------
// Two launches over a (column block) x (row chunk) grid: partial column sums per row chunk,
// then every chunk walks its rows again with the sum of the chunks above it as base.
constexpr uint32_t PART_CHUNKS = 64;

template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) partition_partial_kernel(KCTX const idx_t* __restrict__ Pm, uint32_t G, uint32_t p,
                                                      uint64_t* __restrict__ partial)
{
    const uint32_t ncb = (p + K_BLOCK_DIM - 1) / K_BLOCK_DIM;
    const uint32_t cb = K_BLOCK_IDX % ncb, rc = K_BLOCK_IDX / ncb;
    const uint32_t rows = (G + PART_CHUNKS - 1) / PART_CHUNKS;
    const uint32_t g0 = rc * rows, g1 = g0 + rows < G ? g0 + rows : G;
    PAR(tid) {
        const uint32_t j = cb * K_BLOCK_DIM + tid;
        if (j < p) {
            uint64_t run = 0;
            for (uint32_t g = g0; g < g1; ++g) {
                const idx_t* row = Pm + (uint64_t)g * (p + 1);
                run += (uint64_t)(row[j + 1] - row[j]);
            }
            partial[(uint64_t)rc * p + j] = run;
        }
    }
}

template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) partition_sizes_kernel(KCTX const idx_t* __restrict__ Pm, uint32_t G, uint32_t p,
                                                    const uint64_t* __restrict__ partial, idx_t* __restrict__ ruler,
                                                    uint64_t* __restrict__ sizes)
{
    const uint32_t ncb = (p + K_BLOCK_DIM - 1) / K_BLOCK_DIM;
    const uint32_t cb = K_BLOCK_IDX % ncb, rc = K_BLOCK_IDX / ncb;
    const uint32_t rows = (G + PART_CHUNKS - 1) / PART_CHUNKS;
    const uint32_t g0 = rc * rows, g1 = g0 + rows < G ? g0 + rows : G;
    PAR(tid) {
        const uint32_t j = cb * K_BLOCK_DIM + tid;
        if (j < p) {
            uint64_t run = 0;
            for (uint32_t c = 0; c < rc; ++c) run += partial[(uint64_t)c * p + j];
            for (uint32_t g = g0; g < g1; ++g) {               // G sorted subarrays (all p, or a shard's slice)
                const idx_t* row = Pm + (uint64_t)g * (p + 1);
                ruler[(uint64_t)g * p + j] = (idx_t)run;
                run += (uint64_t)(row[j + 1] - row[j]);
            }
            if (rc == PART_CHUNKS - 1) sizes[j] = run;
        }
    }
}

// Collate: move every element of sorted subarray g to its slot in its partition
// (reference: the p^2 memcpy's at cpp:343-358).  Element x of subarray g belongs to
// partition j with Pm[g][j] <= x < Pm[g][j+1]; slot = part_start[j] + ruler[g][j] + (x - Pm[g][j]).
// collate_plan_kernel finds the partition of every tile's first element; collate_kernel stages
// the slice of the subarray's Pm row that its tile spans (with the slot bases) in LDS and
// resolves every element there.  Reads are coalesced; writes land in runs of consecutive
// slots.  Only (key, sa) move: phase 2 rebuilds the LCPs (the reference resets run-head LCPs
// here, cpp:356).
constexpr uint32_t COLLATE_ROW = TILE_E + 2;     // row entries a tile can span without empty partitions in between

template <typename idx_t>
DEV_INLINE uint32_t partition_of(const idx_t* __restrict__ row, uint32_t p, uint64_t x)
{
    uint32_t lo = 0, hi = p;                                     // j = #{j' in [1,p] : row[j'] <= x}
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t)row[mid + 1] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) collate_plan_kernel(KCTX SegDesc sd, uint32_t p, const idx_t* __restrict__ Pm,
                                                 uint32_t* __restrict__ first_part)
{
    PAR(tid) {
        const uint64_t b = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (b < sd.tile_off[sd.G]) {
            const uint32_t g = sd.tile_rec[b].g;
            const TileInfo t = tile_info(sd, (uint32_t)b);
            first_part[b] = partition_of<idx_t>(Pm + (uint64_t)g * (p + 1), p, (uint64_t)t.tl * TILE_E);
        }
    }
}

template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(TILE_NT) collate_kernel(KCTX SegDesc sd, uint32_t p, const idx_t* __restrict__ Pm,
                                                const idx_t* __restrict__ ruler, const uint64_t* __restrict__ part_start,
                                                const uint32_t* __restrict__ first_part,
                                                const uint64_t* __restrict__ in_key, const idx_t* __restrict__ in_sa,
                                                uint64_t* __restrict__ out_key, idx_t* __restrict__ out_sa)
{
#if CAPS_COLLATE_SWZ
    const uint32_t b = (uint32_t)xcd_swizzle(K_BLOCK_IDX, K_GRID_DIM);
#else
    const uint32_t b = K_BLOCK_IDX;
#endif
    if (b >= sd.tile_off[sd.G]) return;
    const uint32_t g = sd.tile_rec[b].g;
    const TileInfo t = tile_info(sd, b);
    const uint64_t x0 = (uint64_t)t.tl * TILE_E;                  // index of the tile's first element in the subarray
    const uint64_t start = t.s0 + x0;
    const uint32_t cnt = (uint32_t)(t.s1 - start < TILE_E ? t.s1 - start : TILE_E);
    const idx_t* row = Pm + (uint64_t)g * (p + 1);
    const uint32_t j0 = first_part[b];
    // last partition the tile touches: the next tile's first one, or p - 1 at the end of the subarray
    const bool last_tile = start + cnt >= t.s1;
    const uint32_t j1 = last_tile ? p - 1 : first_part[b + 1];
    const uint32_t ns = j1 - j0 + 1;                               // partitions spanned
    SHARED_ARRAY(idx_t, lrow, COLLATE_ROW);                        // row[j0 .. j1 + 1]
    SHARED_ARRAY(uint64_t, lbase, COLLATE_ROW);                    // slot of (partition j, x = 0)
    const bool staged = ns + 1 <= COLLATE_ROW;
    if (staged) {
        PAR(tid) {
            for (uint32_t k = tid; k <= ns; k += K_BLOCK_DIM) {
                const uint32_t j = j0 + k;
                lrow[k] = row[j];
                if (k < ns) lbase[k] = part_start[j] + (uint64_t)ruler[(uint64_t)g * p + j] - (uint64_t)row[j];
            }
        }
        SYNC();
    }
    PAR(tid) {
        UNROLL
        for (uint32_t k = 0; k < TILE_EPT; ++k) {
            const uint32_t e = tid + k * TILE_NT;
            if (e < cnt) {
                const uint64_t x = x0 + e;
                uint64_t dst;
                if (staged) {
                    uint32_t lo = 0, hi = ns - 1;                 // largest k with lrow[k] <= x  (lrow[0] <= x0 <= x)
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi + 1) >> 1;
                        if ((uint64_t)lrow[mid] <= x) lo = mid; else hi = mid - 1;
                    }
                    dst = lbase[lo] + x;
                } else {
                    const uint32_t j = partition_of<idx_t>(row, p, x);
                    dst = part_start[j] + (uint64_t)ruler[(uint64_t)g * p + j] + (x - (uint64_t)row[j]);
                }
                out_key[dst] = in_key[start + e];
                out_sa[dst] = in_sa[start + e];
            }
        }
    }
}

// a2 as a stand-alone entry point: lcp of suffix pairs (a[i], b[i]).
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) lcp_pairs_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, const idx_t* __restrict__ a,
                                              const idx_t* __restrict__ b, uint64_t cnt, idx_t* __restrict__ out)
{
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < cnt) {
            const uint64_t x = a[i], y = b[i];
            out[i] = (idx_t)(x == y ? n - x : deep_lcp<BITS>(P, n, x, y, 0));
        }
    }
}

// ----------------------------------------------------------------------------------
// Deferred ties, resolved (pipeline.h msd_refine): MSD refinement of the groups of equal keys a sort has left open.
//
// A group = a maximal run of entries of the finished arrays whose LCPs are TIE_SENTINEL, plus the entry before the run (its
// head, whose LCP with ITS predecessor is already right): suffixes that share their first D = KCH chars, at positions
// [pos, pos + len) of SA that are theirs whatever their order turns out to be.  The members are copied to compact work arrays
// (wsa; seg_start / gpos: where a group starts there and in SA), and then, level by level:
//   re-key      key = the KCH chars behind the D known-equal ones: one window of the text per member
//   sort        every group by (key, position descending) -- the ordinary segmented sort, text-free (SortOpts::keys_only)
//   classify    neighbours with different keys are settled: LCP = D + what the two keys share.  Equal keys: still open, unless
//               the upper one's suffix ends inside this window -- then it is a prefix of the lower one (the pad behind the
//               text is the smallest code), sorts first (it has the larger position) and the LCP is its length
//   compact     what is still open (runs of >= 2) moves on, D += KCH
// Groups of at most MSD_FIN_MAX members leave the loop at once: one wave ranks them by direct comparison from depth D on
// (deep_scan: run-table aware, any depth).  The reference settles every one of these pairs by scanning both suffixes from
// their known common prefix (src/Suffix_Array.cpp:69-80 through LCP<8>, include/Suffix_Array.hpp:195-241).
// ----------------------------------------------------------------------------------
#ifndef CAPS_MSD_FIN_MAX
#define CAPS_MSD_FIN_MAX 32
#endif
constexpr uint32_t MSD_FIN_MAX = CAPS_MSD_FIN_MAX;
constexpr uint64_t MSD_TILE_BIT = 1ull << 40;             // tile counts: elements in the low 40 bits, tiles above
constexpr uint64_t MSD_ELEM_MASK = MSD_TILE_BIT - 1;

// tcnt[b] = (elements of tile b) | MSD_TILE_BIT if the tile's segment holds deferred ties, else 0
GLOBAL_FN LAUNCH_BOUNDS(256) msd_tile_counts_kernel(KCTX SegDesc sd, const uint64_t* __restrict__ bflag, uint64_t* __restrict__ tcnt)
{
    PAR(tid) {
        const uint64_t b = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (b < sd.tile_off[sd.G]) {
            const TileInfo t = tile_info(sd, (uint32_t)b);
            const uint64_t start = t.s0 + (uint64_t)t.tl * TILE_E;
            const uint64_t cnt = t.s1 - start < TILE_E ? t.s1 - start : TILE_E;
            tcnt[b] = bflag[t.g] ? (cnt | MSD_TILE_BIT) : 0;
        }
    }
}

// ftile[j] = the j-th flagged tile (toff = exclusive scan of tcnt)
GLOBAL_FN LAUNCH_BOUNDS(256) msd_tile_list_kernel(KCTX const uint64_t* __restrict__ tcnt, const uint64_t* __restrict__ toff, uint32_t n_tiles,
                                                  uint32_t* __restrict__ ftile)
{
    PAR(tid) {
        const uint64_t b = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (b < n_tiles && tcnt[b]) ftile[toff[b] >> 40] = (uint32_t)b;
    }
}

// suffix `sa` ends inside (or before) the window of chars [D, D + KCH) of the group's members
template <int BITS> HD bool msd_ending(uint64_t n, uint64_t sa, uint64_t D) { return n - sa <= D + TextTraits<BITS>::KCH; }

// Level 0: the members of the flagged tiles.  flags[toff(tile) + e] = member | head << 32 for entry e of the tile: an entry is
// tied to its predecessor when its LCP is the sentinel.  SA / LCP: the slice the sort wrote (segment offsets).
// (A suffix that ends inside the key is a member like any other -- the merge passes may have left it anywhere in its group;
// it is a prefix of the other members, and the levels / the comparisons below place it first.)
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) msd_flags0_kernel(KCTX SegDesc sd, const uint32_t* __restrict__ ftile, const uint64_t* __restrict__ toff,
                                               const idx_t* __restrict__ LCP, uint64_t* __restrict__ flags)
{
    const uint32_t b = ftile[K_BLOCK_IDX];
    const TileInfo t = tile_info(sd, b);
    const uint64_t start = t.s0 + (uint64_t)t.tl * TILE_E;
    const uint32_t cnt = (uint32_t)(t.s1 - start < TILE_E ? t.s1 - start : TILE_E);
    const uint64_t f0 = toff[b] & MSD_ELEM_MASK;
    const idx_t SENT = tie_sentinel<idx_t>();
    PAR(tid) {
        for (uint32_t e = tid; e < cnt; e += K_BLOCK_DIM) {
            const uint64_t p = start + e;
            const bool tp = LCP[p] == SENT;                  // (a segment head never carries the sentinel)
            const bool tn = p + 1 < t.s1 && LCP[p + 1] == SENT;
            const bool member = tp || tn;
            flags[f0 + e] = (member ? 1ull : 0ull) | (member && !tp ? 1ull << 32 : 0ull);
        }
    }
}

// ... and their copy into the work arrays: wsa[c] = SA[p], a head opens group g': seg_start[g'] = c, gpos[g'] = p
// (offs = exclusive scan of flags: members before in the low word, heads before in the high one)
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) msd_compact0_kernel(KCTX SegDesc sd, const uint32_t* __restrict__ ftile, const uint64_t* __restrict__ toff,
                                                 const idx_t* __restrict__ SA, const uint64_t* __restrict__ flags,
                                                 const uint64_t* __restrict__ offs, idx_t* __restrict__ wsa,
                                                 uint64_t* __restrict__ seg_start, uint64_t* __restrict__ gpos, uint32_t* __restrict__ wgid,
                                                 uint64_t* __restrict__ gdepth, uint64_t D0, uint64_t* __restrict__ gplen)
{
    // gplen[g] = the size of the group this one was split off from (0: none): a group as large as its parent made no progress
    // wgid[c] = the member's group (heads before it, itself included); gdepth[g] = D0: the chars the group's members are known to share
    const uint32_t b = ftile[K_BLOCK_IDX];
    const TileInfo t = tile_info(sd, b);
    const uint64_t start = t.s0 + (uint64_t)t.tl * TILE_E;
    const uint32_t cnt = (uint32_t)(t.s1 - start < TILE_E ? t.s1 - start : TILE_E);
    const uint64_t f0 = toff[b] & MSD_ELEM_MASK;
    PAR(tid) {
        for (uint32_t e = tid; e < cnt; e += K_BLOCK_DIM) {
            const uint64_t f = flags[f0 + e];
            if (f & 1u) {
                const uint64_t o = offs[f0 + e], c = o & 0xFFFFFFFFull;
                wsa[c] = SA[start + e];
                wgid[c] = (uint32_t)((o >> 32) + (f >> 32)) - 1u;
                if (f >> 32) { seg_start[o >> 32] = c; gpos[o >> 32] = start + e; gdepth[o >> 32] = D0; gplen[o >> 32] = 0; }
            }
        }
    }
}

// closes a group table: seg_start[G] = m, out3 = {G, m, 0, 0, 0} (total = offs[count]: members | groups << 32)
GLOBAL_FN LAUNCH_BOUNDS(64) msd_close_kernel(KCTX const uint64_t* __restrict__ total, uint64_t* __restrict__ seg_start, uint64_t* __restrict__ out3)
{
    PAR(tid) {
        if (tid == 0 && K_BLOCK_IDX == 0) {
            const uint64_t m = total[0] & 0xFFFFFFFFull, G = total[0] >> 32;
            seg_start[G] = m;
            out3[0] = G;
            out3[1] = m;
            out3[2] = 0;                               // groups larger than MSD_FIN_MAX (msd_finish_kernel counts them),
            out3[3] = 0;                               //   the largest of them,
            out3[4] = 0;                               //   (the same count: they are sorted tile by tile),
            out3[5] = 0;                               //   members listed for msd_finish_kernel,
            out3[6] = out3[7] = out3[8] = 0;           //   groups listed for msd_quick_kernel, by capacity class
        }
    }
}

// ---- one level.  Everything below is one THREAD per member of the work array (wgid: its group; skip[g] != 0: the group is
// finished -- its members are dead weight until the next compaction drops them).
constexpr uint32_t MSD_QK_MAX = TILE_E;       // groups up to this size are finished in LDS (msd_quick_kernel); only larger ones see another level
// (three capacities, so that small groups do not occupy a compute unit each: 8 / 32 / 128 KB of LDS per workgroup)
constexpr uint32_t MSD_QK_CAP[3] = {TILE_E / 16, TILE_E / 4, TILE_E};
constexpr uint32_t MSD_QK_NT[3] = {TILE_NT / 4, TILE_NT / 4, TILE_NT};

// key = the KCH chars behind the D known-equal ones
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) msd_rekey_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, const uint64_t* __restrict__ gdepth, uint64_t m,
                                              const uint32_t* __restrict__ wgid, const uint8_t* __restrict__ skip,
                                              const idx_t* __restrict__ wsa, uint64_t* __restrict__ wkey)
{
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < m && !skip[wgid[i]]) {
            const uint64_t pos = (uint64_t)wsa[i] + gdepth[wgid[i]];
            wkey[i] = pos < n ? window64<BITS>(P, pos) : 0;       // (a suffix that has ended: the pad, the smallest key)
        }
    }
}

// The groups sorted by (key, position descending): what is settled goes to SA / LCP, what is not is flagged for the next level.
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) msd_classify_kernel(KCTX uint64_t n, const uint64_t* __restrict__ gdepth, uint64_t m, const uint64_t* __restrict__ seg_start,
                                                 const uint32_t* __restrict__ wgid, const uint8_t* __restrict__ skip,
                                                 const uint64_t* __restrict__ key, const idx_t* __restrict__ sa,
                                                 const uint64_t* __restrict__ gpos, idx_t* __restrict__ SA, idx_t* __restrict__ LCP,
                                                 uint64_t* __restrict__ flags)
{
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < m) {
            const uint32_t g = wgid[i];
            uint64_t f = 0;
            if (!skip[g]) {
                const uint64_t s0 = seg_start[g], s1 = seg_start[g + 1], D = gdepth[g];
                const uint64_t k = key[i], a = (uint64_t)sa[i];
                const uint64_t pos = gpos[g] + (i - s0);
                bool tp = false;
                SA[pos] = (idx_t)a;
                if (i > s0) {
                    const uint64_t kp = key[i - 1], ap = (uint64_t)sa[i - 1];
                    if (kp != k) {
                        const uint64_t l = D + (uint32_t)caps_clz64(kp ^ k) / BITS, room = n - (a > ap ? a : ap);
                        LCP[pos] = (idx_t)(l < room ? l : room);
                    } else if (msd_ending<BITS>(n, ap, D)) LCP[pos] = (idx_t)(n - ap);     // the upper one is a prefix of this one
                    else tp = true;
                }
                const bool tn = i + 1 < s1 && key[i + 1] == k && !msd_ending<BITS>(n, a, D);
                const bool member = tp || tn;
                f = (member ? 1ull : 0ull) | (member && !tp ? 1ull << 32 : 0ull);
            }
            flags[i] = f;
        }
    }
}

// ... and its members move on (as msd_compact0_kernel, from the sorted work arrays; wgid_out may not alias wgid)
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) msd_compact_kernel(KCTX uint64_t m, const uint64_t* __restrict__ seg_in, const uint32_t* __restrict__ wgid,
                                                const idx_t* __restrict__ sa, const uint64_t* __restrict__ flags,
                                                const uint64_t* __restrict__ offs, const uint64_t* __restrict__ gpos_in,
                                                idx_t* __restrict__ wsa, uint64_t* __restrict__ seg_start, uint64_t* __restrict__ gpos,
                                                uint32_t* __restrict__ wgid_out, const uint64_t* __restrict__ gdepth_in,
                                                uint64_t* __restrict__ gdepth, uint32_t kch, uint64_t* __restrict__ gplen)
{
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < m) {
            const uint64_t f = flags[i];
            if (f & 1u) {
                const uint32_t g = wgid[i];
                const uint64_t o = offs[i], c = o & 0xFFFFFFFFull;
                wsa[c] = sa[i];
                wgid_out[c] = (uint32_t)((o >> 32) + (f >> 32)) - 1u;
                if (f >> 32) { seg_start[o >> 32] = c; gpos[o >> 32] = gpos_in[g] + (i - seg_in[g]); gdepth[o >> 32] = gdepth_in[g] + kch; gplen[o >> 32] = seg_in[g + 1] - seg_in[g]; }
            }
        }
    }
}

// A group that came out of a level as large as it went in: its members agree far beyond the next window (copies of a long exact
// repeat, suffixes at one offset of equally long letter runs: 32 chars per level would take thousands of levels).  Its depth jumps to
// what ALL its members share: msd_jump_kernel -- one thread per member, its common prefix with the group's first member from the
// known depth on (deep_scan: a periodic stretch costs one step), the minimum per group in gmin (stored inverted: an atomic
// max; preset to 0) -- then msd_jump_apply_kernel rounds it down to whole windows.
template <typename idx_t, int BITS, bool RUNS>
GLOBAL_FN LAUNCH_BOUNDS(256) msd_jump_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, uint64_t m, const uint64_t* __restrict__ seg_start,
                                             const uint32_t* __restrict__ wgid, const uint8_t* __restrict__ skip, const idx_t* __restrict__ wsa,
                                             const uint64_t* __restrict__ gdepth, const uint64_t* __restrict__ gplen, uint64_t* __restrict__ gmin)
{
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < m) {
            const uint32_t g = wgid[i];
            const uint64_t s0 = seg_start[g];
            if (!skip[g] && i != s0 && seg_start[g + 1] - s0 == gplen[g]) {
                const uint64_t l = deep_lcp<BITS, RUNS>(P, n, (uint64_t)wsa[s0], (uint64_t)wsa[i], gdepth[g]);
                ATOMIC_MAX_U64(&gmin[g], ~l);
            }
        }
    }
}
GLOBAL_FN LAUNCH_BOUNDS(256) msd_jump_apply_kernel(KCTX const uint64_t* __restrict__ out3, const uint8_t* __restrict__ skip,
                                                   const uint64_t* __restrict__ gmin, uint32_t kch, uint64_t* __restrict__ gdepth)
{
    const uint64_t G = out3[0];
    PAR(tid) {
        for (uint64_t g = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; g < G; g += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
            if (!skip[g] && gmin[g]) {
                const uint64_t d = (~gmin[g] / kch) * kch;
                if (d > gdepth[g]) gdepth[g] = d;
            }
        }
    }
}

// The two ends of every level-0 group.  The LCP at a group's head (with the suffix before the group) and the LCP of the suffix
// behind the group were emitted by the sort from the members that stood there THEN; they depend on the member only through its
// length (all members share the key) -- but a member whose suffix ends inside the key is shorter than that, and the stable merges
// may have left it anywhere.  edges[2g], edges[2g + 1] = the group's first position and the one behind it; once every group is
// in its final order both LCPs are taken from the text again (two suffixes with different keys: one window).
GLOBAL_FN LAUNCH_BOUNDS(256) msd_edges_kernel(KCTX const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ gpos,
                                              const uint64_t* __restrict__ out3, uint64_t* __restrict__ edges)
{
    const uint64_t G = out3[0];
    PAR(tid) {
        for (uint64_t g = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; g < G; g += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
            edges[2 * g] = gpos[g];
            edges[2 * g + 1] = gpos[g] + (seg_start[g + 1] - seg_start[g]);
        }
    }
}
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(256) msd_fix_edges_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, const uint64_t* __restrict__ edges, uint64_t count,
                                                  uint64_t total, const idx_t* __restrict__ SA, idx_t* __restrict__ LCP)
{
    PAR(tid) {
        const uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid;
        if (i < count) {
            const uint64_t p = edges[i];
            if (p > 0 && p < total) LCP[p] = (idx_t)deep_lcp<BITS, false>(P, n, (uint64_t)SA[p - 1], (uint64_t)SA[p], 0);
        }
    }
}

// The groups of a level, one thread each.  At most MSD_FIN_MAX members: listed for msd_finish_kernel (flist: the indices of its
// members, out3[5] of them in all).  At most MSD_QK_MAX: listed for msd_quick_kernel by capacity class (qlist[c]: the group,
// out3[6 + c] of them).  Both are
// taken out (skip[g] = skip_tiles[g] = 1).  Larger ones go through another level: out3[2] = out3[4] counts them, out3[3] = the largest.
GLOBAL_FN LAUNCH_BOUNDS(256) msd_groups_kernel(KCTX const uint64_t* __restrict__ seg_start, uint8_t* __restrict__ skip, uint8_t* __restrict__ skip_tiles,
                                               uint32_t* __restrict__ flist, uint32_t* __restrict__ qlist0, uint32_t* __restrict__ qlist1,
                                               uint32_t* __restrict__ qlist2, uint64_t* __restrict__ out3)
{
    // (a workgroup reserves its share of the lists and adds its counts with ONE global atomic each: a million groups bumping the
    // counters one by one took longer than finishing them)
    SHARED_ARRAY(uint32_t, acc, 6);            // [0] members listed by this workgroup, [1] larger groups, [2] the largest, [3..5] groups per quick class
    SHARED_ARRAY(uint64_t, at0, 4);
    const uint64_t G = out3[0];
    for (uint64_t gb = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM; gb < G; gb += (uint64_t)K_GRID_DIM * K_BLOCK_DIM) {
        TL_DECL(uint32_t, mine, 1);
        TL_DECL(uint32_t, mineq, 1);
        PAR(tid) { if (tid < 6) acc[tid] = 0; }
        SYNC();
        PAR(tid) {
            const uint64_t g = gb + tid;
            TL(mine, tid, 0) = ~0u;
            TL(mineq, tid, 0) = ~0u;
            if (g < G) {
                const uint64_t len = seg_start[g + 1] - seg_start[g];
                if (len <= MSD_QK_MAX) {
                    if (len <= MSD_FIN_MAX) TL(mine, tid, 0) = FETCH_ADD_U32(&acc[0], (uint32_t)len);
                    else {
                        const uint32_t c = len <= MSD_QK_CAP[0] ? 0u : len <= MSD_QK_CAP[1] ? 1u : 2u;
                        TL(mineq, tid, 0) = FETCH_ADD_U32(&acc[3 + c], 1u) | (c << 30);
                    }
                    skip[g] = 1;
                    skip_tiles[g] = 1;
                } else {
                    skip[g] = 0;
                    skip_tiles[g] = 0;
                    FETCH_ADD_U32(&acc[1], 1u);
                    ATOMIC_MAX_U32(&acc[2], (uint32_t)(len < 0xFFFFFFFFull ? len : 0xFFFFFFFFull));
                }
            }
        }
        SYNC();
        PAR(tid) {
            if (tid == 0) {
                at0[0] = acc[0] ? FETCH_ADD_U64(&out3[5], (uint64_t)acc[0]) : 0;
                for (uint32_t c = 0; c < 3; ++c) at0[1 + c] = acc[3 + c] ? FETCH_ADD_U64(&out3[6 + c], (uint64_t)acc[3 + c]) : 0;
                if (acc[1]) { ATOMIC_ADD_U64(&out3[2], (uint64_t)acc[1]); ATOMIC_ADD_U64(&out3[4], (uint64_t)acc[1]); }
                if (acc[2]) ATOMIC_MAX_U64(&out3[3], (uint64_t)acc[2]);
            }
        }
        SYNC();
        PAR(tid) {
            const uint64_t g = gb + tid;
            if (g < G && TL(mine, tid, 0) != ~0u) {
                const uint64_t s0 = seg_start[g], len = seg_start[g + 1] - s0, at = at0[0] + TL(mine, tid, 0);
                for (uint64_t k = 0; k < len; ++k) flist[at + k] = (uint32_t)(s0 + k);
            }
            if (g < G && TL(mineq, tid, 0) != ~0u) {
                const uint32_t c = TL(mineq, tid, 0) >> 30, at = TL(mineq, tid, 0) & 0x3FFFFFFFu;
                (c == 0 ? qlist0 : c == 1 ? qlist1 : qlist2)[at0[1 + c] + at] = (uint32_t)g;
            }
        }
        SYNC();
    }
}

// The listed groups are finished by direct comparison: MSD_FIN_LANES threads per member share the other members of its group (the
// comparisons of a member in flight together: a compare is a chain of dependent reads of the text), each compares from depth D
// on (deep_scan: run-table aware, any depth); the member's rank is the number of smaller ones, its LCP the largest one it shares
// with a smaller one (the predecessor's).  The head's LCP with ITS predecessor stands.
constexpr uint32_t MSD_FIN_LANES = 8;
constexpr uint32_t MSD_FIN_NT = 256, MSD_FIN_MEMBERS = MSD_FIN_NT / MSD_FIN_LANES;
template <typename idx_t, int BITS, bool RUNS>
GLOBAL_FN LAUNCH_BOUNDS(MSD_FIN_NT) msd_finish_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, const uint64_t* __restrict__ gdepth, const uint64_t* __restrict__ seg_start,
                                               const uint64_t* __restrict__ gpos, const idx_t* __restrict__ wsa, const uint32_t* __restrict__ wgid,
                                               const uint32_t* __restrict__ flist, idx_t* __restrict__ SA, idx_t* __restrict__ LCP,
                                               const uint64_t* __restrict__ out3)
{
    SHARED_ARRAY(uint32_t, rk, MSD_FIN_MEMBERS);
    SHARED_ARRAY(uint64_t, best, MSD_FIN_MEMBERS);
    const uint64_t m = out3[5];
    for (uint64_t base = (uint64_t)K_BLOCK_IDX * MSD_FIN_MEMBERS; base < m; base += (uint64_t)K_GRID_DIM * MSD_FIN_MEMBERS) {
        PAR(tid) { if (tid < MSD_FIN_MEMBERS) { rk[tid] = 0; best[tid] = 0; } }
        SYNC();
        PAR(tid) {
            const uint32_t q = tid / MSD_FIN_LANES, lane = tid % MSD_FIN_LANES;
            if (base + q < m) {
                const uint64_t i = flist[base + q];
                const uint32_t g = wgid[i];
                const uint64_t s0 = seg_start[g], len = seg_start[g + 1] - s0, D = gdepth[g];
                const uint64_t a = (uint64_t)wsa[i];
                uint32_t cnt = 0;
                uint64_t bl = 0;
                for (uint64_t j = s0 + lane; j < s0 + len; j += MSD_FIN_LANES) {
                    if (j == i) continue;
                    const uint64_t c = (uint64_t)wsa[j];
                    const uint64_t maxlen = n - (a > c ? a : c);
                    uint64_t wa = 0, wc = 0;
                    uint64_t l = deep_scan<BITS, RUNS>(P, n, c, a, D, maxlen, wc, wa);
                    const bool c_first = l < maxlen ? wc < wa : c > a;         // (one a prefix of the other: the shorter first)
                    l = l < maxlen ? l : maxlen;
                    if (c_first) { ++cnt; bl = l > bl ? l : bl; }
                }
                if (cnt) { FETCH_ADD_U32(&rk[q], cnt); ATOMIC_MAX_LDS_U64(&best[q], bl); }
            }
        }
        SYNC();
        PAR(tid) {
            if (tid < MSD_FIN_MEMBERS && base + tid < m) {
                const uint64_t i = flist[base + tid];
                const uint64_t p = gpos[wgid[i]] + rk[tid];
                SA[p] = wsa[i];
                if (rk[tid]) LCP[p] = (idx_t)best[tid];
            }
        }
        SYNC();                                        // rk / best are free for the next batch
    }
}

// A group of at most MSD_QK_MAX members, finished in ONE launch: multikey quicksort in LDS, every level of the refinement without a
// trip through global memory but the windows of the text.  The workgroup holds the group's positions; a SUBSEGMENT is a run of
// them known to share D0 + lvl windows (at first: the whole group, lvl 0).  Every round, every open subsegment is split three
// ways around its first member's window at that depth: smaller | equal | larger (a member whose suffix ends inside the window is
// a prefix of the others: by position, the shorter first).  The equal part moves on to the next window, the other two are
// split again at the same depth; a part of one member is settled.  A subsegment that comes out whole (all windows equal) jumps
// to what all its members share (deep_lcp: a periodic stretch is one step).  The LCP of a member with its predecessor follows
// from the round that separated the two: the depth of that round + what their windows there share (sep[i]: its level).
// Replaces ~45 launches-and-syncs deep level loops over the thousands of groups of a tandem array (25 of 119 ms on the
// GRCh38-shaped workload) by one kernel.  Every round waits for the text once (one workgroup: ~2 us), so the compute unit
// must hold several groups: capacities of 256 / 1024 / 4096 members (MSD_QK_CAP: 8 / 32 / 128 KB of LDS).
constexpr uint32_t MSD_QK_SHIFT = 21;             // cnt[start] = smaller | equal << 21 | larger << 42
constexpr uint32_t MSD_QK_MASK = (1u << MSD_QK_SHIFT) - 1u;
constexpr uint32_t MSD_QK_WHOLE = 1u << 31;       // (in a member's class | rank << 2 word)
static_assert(TILE_E < (1u << MSD_QK_SHIFT) && TILE_E <= 65535, "counts fit their fields, subsegment starts fit 16 bits");
template <typename idx_t, int BITS, bool RUNS, uint32_t CAP, uint32_t NT>
GLOBAL_FN LAUNCH_BOUNDS(NT) msd_quick_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, const uint32_t* __restrict__ qlist,
                                                  const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ gpos,
                                                  const uint64_t* __restrict__ gdepth, const idx_t* __restrict__ wsa, idx_t* __restrict__ SA,
                                                  idx_t* __restrict__ LCP)
{
    constexpr uint32_t KCH = TextTraits<BITS>::KCH, EPT = CAP / NT;
    static_assert(CAP % NT == 0 && CAP <= MSD_QK_MAX, "capacity classes");
    SHARED_ARRAY(idx_t, ssa, CAP);
    SHARED_ARRAY(uint16_t, sid, CAP);          // the start of the member's subsegment
    SHARED_ARRAY(uint16_t, slen, CAP);         // [start] members (1: settled)
    SHARED_ARRAY(uint32_t, slvl, CAP);         // [start] windows behind D0 its members are known to share
    SHARED_ARRAY(uint32_t, sep, CAP);          // [i] the level of the round that separated member i from member i - 1
    SHARED_ARRAY(uint64_t, cnt, CAP);          // [start] the three counts of a round (zero between rounds)
    SHARED_ARRAY(uint64_t, jacc, CAP / 2);     // [start / 2] what a subsegment that came out whole shares, inverted minimum (zero between rounds;
                                               //   an open subsegment has two members at least: start / 2 is its own)
    SHARED_ARRAY(uint32_t, open, 2);           // [round & 1] a subsegment of two or more members is left
    TL_DECL(idx_t, ra, EPT);
    TL_DECL(idx_t, rp, EPT);                 // the first member of the subsegment this round (the one the others are compared with)
    TL_DECL(uint32_t, rs, EPT);              // the member's subsegment (start), ~0: settled
    TL_DECL(uint32_t, rc, EPT);              // class | rank << 2 | MSD_QK_WHOLE
    TL_DECL(uint32_t, rl, EPT);              // the subsegment's level in this round
    const uint32_t g = qlist[K_BLOCK_IDX];
    const uint64_t s0 = seg_start[g], D0 = gdepth[g], p0 = gpos[g];
    const uint32_t N = (uint32_t)(seg_start[g + 1] - s0);
    PAR(tid) {
        for (uint32_t e = tid; e < N; e += NT) { ssa[e] = wsa[s0 + e]; sid[e] = 0; sep[e] = 0; cnt[e] = 0; jacc[e >> 1] = 0; }
        if (tid == 0) { slen[0] = (uint16_t)N; slvl[0] = 0; open[0] = 0; open[1] = N >= 2 ? 1u : 0u; }
    }
    // Three barriers per round (a round is as long as its trip to the text and its barriers: the launch lasts as long as its deepest
    // group, hundreds of rounds): every member fetches its own window AND the first member's (no hand-over through LDS), classes,
    // moves and records share a phase (everything they read was read before the barrier in front of them).
    for (uint32_t round = 0;; ++round) {
        SYNC();
        if (open[(round + 1u) & 1u] == 0) break;                               // block-uniform; set in the round before (or at the start)
        PAR(tid) {                                 // 1: windows, class, rank inside the class (one LDS atomic per member)
            if (tid == 0) open[round & 1u] = 0;                                //    (this round's flag: last read two barriers ago)
            UNROLL
            for (uint32_t k = 0; k < EPT; ++k) {
                const uint32_t e = tid + k * NT;
                TL(rs, tid, k) = ~0u;
                if (e < N) {
                    const uint32_t s = sid[e];
                    if (slen[s] >= 2) {
                        const uint32_t lvl = slvl[s];
                        const idx_t a = ssa[e], apx = ssa[s];
                        const uint64_t depth = D0 + (uint64_t)lvl * KCH, ap = (uint64_t)apx;
                        const uint64_t pa = (uint64_t)a + depth, pp = ap + depth;
                        const uint64_t key = pa < n ? window64<BITS>(P, pa) : 0, kp = pp < n ? window64<BITS>(P, pp) : 0;
                        uint32_t c = 1;
                        if (e != s) {
                            if (key != kp) c = key < kp ? 0u : 2u;
                            else if (msd_ending<BITS>(n, (uint64_t)a, depth) || msd_ending<BITS>(n, ap, depth)) c = (uint64_t)a > ap ? 0u : 2u;
                        }
                        const uint64_t old = FETCH_ADD_U64(&cnt[s], 1ull << (MSD_QK_SHIFT * c));
                        TL(ra, tid, k) = a;
                        TL(rp, tid, k) = apx;
                        TL(rs, tid, k) = s;
                        TL(rl, tid, k) = lvl;
                        TL(rc, tid, k) = c | ((uint32_t)((old >> (MSD_QK_SHIFT * c)) & MSD_QK_MASK) << 2);
                    }
                }
            }
        }
        SYNC();
        PAR(tid) {                                 // 2: the members move to their classes; the first of every class opens the class's record; a
            UNROLL                                 //    subsegment that came out whole measures what its members share with its first one
            for (uint32_t k = 0; k < EPT; ++k) {
                const uint32_t s = TL(rs, tid + 0u, k);
                if (s == ~0u) continue;
                const uint64_t pk = cnt[s];
                const uint32_t n0 = (uint32_t)(pk & MSD_QK_MASK), n1 = (uint32_t)((pk >> MSD_QK_SHIFT) & MSD_QK_MASK), n2 = (uint32_t)(pk >> (2 * MSD_QK_SHIFT));
                const uint32_t c = TL(rc, tid, k) & 3u, r = (TL(rc, tid, k) >> 2) & MSD_QK_MASK, lvl = TL(rl, tid, k);
                const uint32_t base = s + (c == 0 ? 0u : c == 1 ? n0 : n0 + n1);
                ssa[base + r] = TL(ra, tid, k);
                sid[base + r] = (uint16_t)base;
                if (r == 0) {
                    const uint32_t nc = c == 0 ? n0 : c == 1 ? n1 : n2;
                    slen[base] = (uint16_t)nc;
                    slvl[base] = lvl + (c == 1 ? 1u : 0u);
                    if (base != s) sep[base] = lvl;
                    if (nc >= 2) open[round & 1u] = 1;
                }
                if (n0 == 0 && n2 == 0 && n1 >= 2) {
                    TL(rc, tid, k) |= MSD_QK_WHOLE;
                    const uint64_t a = (uint64_t)TL(ra, tid, k), ap = (uint64_t)TL(rp, tid, k);
                    if (a != ap) {
                        const uint64_t l = deep_lcp<BITS, RUNS>(P, n, ap, a, D0 + ((uint64_t)lvl + 1u) * KCH);
                        ATOMIC_MAX_LDS_U64(&jacc[s >> 1], ~((l - D0) / KCH));                    // (the minimum, inverted)
                    }
                }
            }
        }
        SYNC();
        PAR(tid) {                                 // 3: ... and jumps there; the counters are zero again
            UNROLL
            for (uint32_t k = 0; k < EPT; ++k) {
                const uint32_t s = TL(rs, tid, k), e = tid + k * NT;
                if (s != ~0u && e == s) {
                    cnt[s] = 0;
                    if ((TL(rc, tid, k) & MSD_QK_WHOLE) && jacc[s >> 1]) {
                        const uint64_t w = ~jacc[s >> 1];
                        jacc[s >> 1] = 0;
                        if (w > slvl[s]) slvl[s] = (uint32_t)(w < 0xFFFFFFFFull ? w : 0xFFFFFFFFull);
                    }
                }
            }
        }
    }
    PAR(tid) {                                     // the group in its final order: SA, and the LCPs from the separating rounds
        for (uint32_t e = tid; e < N; e += NT) {
            const uint64_t a = (uint64_t)ssa[e];
            SA[p0 + e] = (idx_t)a;
            if (e) {
                const uint64_t b = (uint64_t)ssa[e - 1], depth = D0 + (uint64_t)sep[e] * KCH;
                const uint64_t wa = a + depth < n ? window64<BITS>(P, a + depth) : 0, wb = b + depth < n ? window64<BITS>(P, b + depth) : 0;
                const uint64_t l = depth + (wa == wb ? KCH : (uint32_t)caps_clz64(wa ^ wb) / BITS), room = n - (a > b ? a : b);
                LCP[p0 + e] = (idx_t)(l < room ? l : room);
            }
        }
    }
}

// ----------------------------------------------------------------------------------
// Multi-GPU exchange helpers.  After the all-to-all-v a rank holds, per source
// rank, that rank's sub-subarrays of the partitions it now owns; regroup_kernel moves run
// d (desc[3d] = source offset, desc[3d+1] = destination offset, desc[3d+2] = length) so
// that every partition is contiguous.  One workgroup per run.
// ----------------------------------------------------------------------------------
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) regroup_kernel(KCTX const uint64_t* __restrict__ desc, const uint64_t* __restrict__ in_key,
                                            const idx_t* __restrict__ in_sa, uint64_t* __restrict__ out_key,
                                            idx_t* __restrict__ out_sa)
{
    const uint64_t src = desc[3 * (uint64_t)K_BLOCK_IDX], dst = desc[3 * (uint64_t)K_BLOCK_IDX + 1],
                   len = desc[3 * (uint64_t)K_BLOCK_IDX + 2];
    PAR(tid) {
        for (uint64_t i = tid; i < len; i += K_BLOCK_DIM) {
            out_key[dst + i] = in_key[src + i];
            out_sa[dst + i] = in_sa[src + i];
        }
    }
}

// LCP of the first suffix of a rank's slice with the last suffix of the previous
// non-empty slice (the a11 boundary between GPUs).
// SA[base - 1] and SA[base] are neighbours in the suffix array but were sorted by different waves: their LCP from the text
template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(64) wave_head_lcp_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, const idx_t* __restrict__ dSA,
                                                 idx_t* __restrict__ dLCP, uint64_t base)
{
    PAR(tid) {
        if (tid == 0 && K_BLOCK_IDX == 0)
            dLCP[base] = (idx_t)deep_lcp<BITS, true>(P, n, (uint64_t)dSA[base - 1], (uint64_t)dSA[base], 0);
    }
}

template <typename idx_t, int BITS>
GLOBAL_FN LAUNCH_BOUNDS(64) first_lcp_kernel(KCTX const uint32_t* __restrict__ P, uint64_t n, uint64_t prev_sa,
                                             const idx_t* __restrict__ sa, idx_t* __restrict__ lcp)
{
    PAR(tid) {
        if (tid == 0 && K_BLOCK_IDX == 0) lcp[0] = (idx_t)deep_lcp<BITS>(P, n, prev_sa, (uint64_t)sa[0], 0);
    }
}

// ----------------------------------------------------------------------------------
// Verifier (idea of the reference's never-called is_sorted,
// cpp:512-536): works on the RAW text with byte loops, independent of the packed text
// and keys.  err[0] += #violations; seen is an n-bit set for the permutation check.
// ----------------------------------------------------------------------------------
// cnt entries of SA / LCP are checked (cnt = n: the whole arrays; cnt < n: a slice of them, e.g. one rank's -- then
// "no value twice" is all the permutation check can say).  head != 0: entry 0 is the head of the suffix array (LCP 0).
template <typename idx_t>
GLOBAL_FN LAUNCH_BOUNDS(256) verify_kernel(KCTX const int8_t* __restrict__ T, uint64_t n, const idx_t* __restrict__ SA,
                                           const idx_t* __restrict__ LCP, uint64_t cnt, uint32_t head, uint32_t* __restrict__ seen,
                                           uint64_t* __restrict__ err)
{
    PAR(tid) {
        const uint64_t stride = (uint64_t)K_GRID_DIM * K_BLOCK_DIM;
        uint64_t bad = 0;
        for (uint64_t i = (uint64_t)K_BLOCK_IDX * K_BLOCK_DIM + tid; i < cnt; i += stride) {
            const uint64_t b = SA[i];
            if (b >= n) { ++bad; continue; }
            const uint32_t bit = 1u << (b & 31);
#ifdef CAPS_EMUL
            const uint32_t old = seen[b >> 5]; seen[b >> 5] |= bit;
#else
            const uint32_t old = atomicOr(&seen[b >> 5], bit);
#endif
            if (old & bit) ++bad;
            if (i == 0) { if (head && LCP[0] != 0) ++bad; continue; }
            const uint64_t a = SA[i - 1];
            if (a >= n) continue;                        // counted by its own thread
            const uint64_t cap = n - (a > b ? a : b);
            uint64_t l = 0;
            while (l < cap && T[a + l] == T[b + l]) ++l;
            if (l != (uint64_t)LCP[i]) ++bad;
            if (l == cap) { if (a < b) ++bad; }
            else if (!(T[a + l] < T[b + l])) ++bad;
        }
        if (bad) ATOMIC_ADD_U64(&err[0], bad);
    }
}

// ---- f2: LCP values leave the device as BYTES (capi_impl.h HostCopySink) -------------------------------------------------------
// The D2H of SA and LCP is what a host-buffer build waits for (24 GB at C3: 390 of 494 ms), and LCP values are small -- mean 12 on
// random DNA, a few dozen on a genome.  out8[i] = lcp[i] if it is below 255, else 255 with (base + i) << 32 | lcp[i] appended to the
// list of exceptions (one atomic per wave; exc_count may run past exc_cap: the host then takes the whole array at full width).  The
// host widens the bytes into the caller's array while the next slice is on the link.  32-bit indices only (positions fit 32 bits).
constexpr uint32_t NARROW_PER = 16;            // consecutive values per thread: four 16-byte loads, one 16-byte store
GLOBAL_FN LAUNCH_BOUNDS(256) lcp_narrow_kernel(KCTX const uint32_t* __restrict__ lcp, uint64_t cnt, uint64_t base, uint8_t* __restrict__ out8,
                                               uint64_t* __restrict__ exc_count, uint64_t exc_cap, uint64_t* __restrict__ exc)
{
    // a workgroup reserves room on the list once per round for all its exceptions (inside a satellite array EVERY value is one: an
    // atomic per thread on the one counter would serialise millions of them)
    SHARED_ARRAY(uint32_t, bsum, 1);
    SHARED_ARRAY(uint64_t, bbase, 1);
    TL_DECL(uint32_t, v, NARROW_PER);
    TL_DECL(uint32_t, off, 1);
    const uint64_t per_round = (uint64_t)K_BLOCK_DIM * NARROW_PER, stride = (uint64_t)K_GRID_DIM * per_round;
    for (uint64_t r0 = (uint64_t)K_BLOCK_IDX * per_round; r0 < cnt; r0 += stride) {               // block-uniform
        PAR(tid) { if (tid == 0) bsum[0] = 0; }
        SYNC();
        PAR(tid) {
            const uint64_t i0 = r0 + (uint64_t)tid * NARROW_PER;
            uint32_t big = 0;
            UNROLL
            for (uint32_t k = 0; k < NARROW_PER; ++k) {
                TL(v, tid, k) = i0 + k < cnt ? lcp[i0 + k] : 0u;
                big += TL(v, tid, k) >= 255u ? 1u : 0u;
            }
            TL(off, tid, 0) = big ? FETCH_ADD_U32(&bsum[0], big) : 0u;
        }
        SYNC();
        PAR(tid) { if (tid == 0 && bsum[0]) bbase[0] = FETCH_ADD_U64(exc_count, (uint64_t)bsum[0]); }
        SYNC();
        PAR(tid) {
            const uint64_t i0 = r0 + (uint64_t)tid * NARROW_PER;
            uint64_t at = bsum[0] ? bbase[0] + TL(off, tid, 0) : 0;
            UNROLL
            for (uint32_t k = 0; k < NARROW_PER; ++k) {
                const uint32_t x = TL(v, tid, k);
                if (x >= 255u) {
                    if (at < exc_cap) exc[at] = ((base + i0 + k) << 32) | x;
                    ++at;
                }
                if (i0 + k < cnt) out8[i0 + k] = (uint8_t)(x < 255u ? x : 255u);
            }
        }
        SYNC();
    }
}

}  // namespace caps
