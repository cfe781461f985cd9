// caps-sa_amd/csrc/pipeline.h
//
// Host orchestration of the construction path: the GPU counterpart of
// Suffix_Array::construct() (reference: src/Suffix_Array.cpp:466-494).  It only
// sequences kernel launches on one HIP stream; all arithmetic is in kernels.h.
//
//   pack text -> [phase 1] sort p subarrays (tile sort + merge passes)
//             -> sample, sort samples, pick p-1 pivots
//             -> locate pivots in every subarray -> partition sizes / ruler / offsets
//             -> collate sub-subarrays into partitions
//             -> [phase 2] sort every partition (tile sort + merge passes)
//             -> partition-boundary LCPs -> SA, LCP
//
// Written against a tiny backend interface (alloc / copies / launch) so that the same
// sequence can be driven by the host emulation used in tests/emul (kernel_lang.h).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.h"
#include "bounded.h"
#include "../../include/caps_sa_hip.h"

// The backend (device memory, copies, launches, events): hip_backend.h in the product.  A translation unit that has
// already defined one (CAPS_BACKEND_DEFINED: the host emulation of tests/emul, which includes its own before this header)
// keeps its own -- the product headers never reach into tests/.
#ifndef CAPS_BACKEND_DEFINED
#include "hip_backend.h"
#endif

namespace caps {

template <typename idx_t> struct ElemBuf {
    uint64_t* key = nullptr;
    idx_t* sa = nullptr;
    idx_t* lcp = nullptr;
    size_t region_bytes = 0;      // != 0: key | sa | lcp are one contiguous allocation of this many bytes starting at key
                                  // (it can then be viewed as a slot buffer of a speculative bucket split)
};

// Bump allocator over one device allocation (or a dry run that only measures).
struct Arena {
    char* base = nullptr;
    size_t off = 0;
    template <typename T> T* take(size_t count)
    {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

struct SegBufs {
    uint64_t* seg_start = nullptr;   // [G+1]
    uint64_t* seg_end = nullptr;     // null: segment g ends at seg_start[g+1]; else [G] ends of segments in fixed-capacity regions
    uint32_t* tile_off = nullptr;    // [G+1]
    TileInfo* tile_rec = nullptr;    // [tile capacity]
    uint64_t* out2 = nullptr;        // [2] = {#tiles, max segment length}
    uint32_t G = 0;
    SegDesc desc() const { return SegDesc{seg_start, tile_off, tile_rec, G}; }
};

constexpr uint32_t kMaxPasses = 96;

// Sub-streams per group of the direct path's level A (one per XCD of the MI355X, see bucket_scatter_kernel).
#ifndef CAPS_DIRECT_SUB
#define CAPS_DIRECT_SUB 8
#endif
constexpr uint32_t DIRECT_SUB = CAPS_DIRECT_SUB;
// Quantile mode of the direct path (skewed keys): mean bucket size (sampling noise on top: sigma = 1 / sqrt(QUANTILE_SPB)) and
// samples drawn per bucket.
#ifndef CAPS_BUCKET_Q_32NDS
#define CAPS_BUCKET_Q_32NDS 24     /* measured at 3e9 genome-like, count-free split: 22 / 24 / 26 thirty-seconds of a tile */
#endif
#ifndef CAPS_QUANTILE_SPB
#define CAPS_QUANTILE_SPB 64       /* ... and 32 / 48 / 64 samples per bucket: 89.4 / 83.6 / 82.1 ms (a bucket above a tile costs a merge pass and more) */
#endif
constexpr uint32_t BUCKET_Q = (TILE_E * CAPS_BUCKET_Q_32NDS) / 32;
constexpr uint32_t QUANTILE_SPB = CAPS_QUANTILE_SPB;

// Fine buckets per bucket slot of the equalised split (bucket_group_kernel), as far as the LDS histograms hold them.
#ifndef CAPS_EQ_FINE
#define CAPS_EQ_FINE 16
#endif
constexpr uint32_t EQ_FINE = CAPS_EQ_FINE;

// Scratch of the bucketing stage of one sort (kernels.h "Bucketing inside a sort").
struct BucketBufs {
    BucketParams* params = nullptr;   // [G]
    uint64_t* segB = nullptr;         // [G]    buckets per parent segment
    uint64_t* bstart = nullptr;       // [G+1]  exclusive scan of segB
    uint64_t* count = nullptr;        // [nb_cap] elements per bucket
    void* cursor = nullptr;           // [nb_cap] idx_t
    uint64_t* scan_tmp = nullptr;     // [2 * (nb_cap / SCAN_CHUNK + 2)]
    uint64_t *first_key = nullptr, *last_key = nullptr;   // [nb_cap] boundary records of the sorted segments
    BucketParams* tile_map = nullptr;                     // [nb_cap] bin map of the tile sort over every bucket's key range
    void *first_sa = nullptr, *last_sa = nullptr;         // [nb_cap] idx_t
    SegBufs sub;                      // the buckets as segments (G = nb_cap, trailing ones empty)
    // equalised split (redo path): fine buckets of every segment and their grouping into the bucket slots
    BucketParams* fparams = nullptr;  // [G]
    uint64_t* fsegB = nullptr;        // [G]
    uint64_t* fstart = nullptr;       // [G+1]
    uint64_t* fcount = nullptr;       // [EQ_FINE * nb_cap]
    uint32_t* gfirst = nullptr;       // [nb_cap] first fine bucket of every slot's group
    uint8_t* kshift = nullptr;        // [nb_cap] 32-bit keys: the shift of every bucket's parent group
    uint8_t* skip = nullptr;          // [nb_cap] buckets that sit out the tile sort and the merge passes (letter-run buckets)
    uint64_t* run_list = nullptr;     // [1 + 4 * RUN_BUCKET_MAX] the letter-run buckets of a quantile split (run_bucket_mark_kernel)
    uint32_t nb_cap = 0;
    uint64_t tile_cap = 0;
    // buckets of a sort: mean size BUCKET_TARGET (linear maps) or BUCKET_Q (quantile mode), at least two per segment
    static uint32_t bucket_bound(uint64_t n_elems, uint32_t G) { return (uint32_t)(n_elems / BUCKET_Q + 2 * (uint64_t)G + 2); }
};

template <typename idx_t> struct Plan {
    uint64_t n = 0;
    int text_bits = 8;       // code width the text arena holds (make_plan)
    uint32_t p = 0;          // effective subproblem count (0/1 -> single segment)
    uint32_t ppp = 0;        // samples per subarray
    uint64_t m = 0;          // number of samples
    uint64_t tile_cap = 0;   // capacity of the per-tile arrays
    uint32_t* P = nullptr;
    ElemBuf<idx_t> A, B, SA_, SB_;
    uint64_t* pkey = nullptr;
    idx_t* psa = nullptr;
    idx_t* Pm = nullptr;
    idx_t* ruler = nullptr;
    idx_t* PmT = nullptr;            // transposes (phase 2 reads partitions through them)
    idx_t* rulerT = nullptr;
    uint64_t* sizes = nullptr;
    uint64_t* gkey = nullptr;        // [p]   direct path: the pivot keys that close the groups
    uint64_t* dstat = nullptr;       // [4]   direct path: {elements scattered, largest overflowed stream, (u32 pivot-tie flag, u32 LUT span),
                                     //         (u32 skew flag, u32 longest run of equal pivot keys)}
    idx_t* dcur = nullptr;           // [gseg] direct path: level A's cursors = sizes of the groups' sub-streams
    uint16_t* glut = nullptr;        // [SPLIT_LUT_CELLS + 1] direct path: LUT over the group keys (split_lut_kernel)
    uint8_t* gshift = nullptr;       // [gseg] direct path, 32-bit keys: the groups' shifts (group_shift_kernel)
    uint64_t* knots = nullptr;       // [nb_cap] quantile mode: upper key of every bucket
    uint64_t* rcap = nullptr;        // [gseg]   quantile mode: capacity of every stream's region
    uint64_t* rstart = nullptr;      // [gseg+1]                 ... and where it starts
    uint64_t* partial = nullptr;     // [PART_CHUNKS * p] partial column sums of Pm
    SegBufs seg1, seg2, segS;
    TileDesc* desc = nullptr;        // [tile_cap] per-pass tile descriptors
    BucketBufs bk;                   // bucketing scratch (shared by phase 1 and phase 2)
    uint64_t* pass_elems = nullptr;  // [kMaxPasses] elements merged by each timed pass
    uint32_t* present = nullptr;     // [8]
    uint8_t* lut = nullptr;          // [256]
    size_t bytes = 0;
};

// Effective subproblem count and samples per subarray: the reference's constructor,
// src/Suffix_Array.cpp:24-27 (default 8192: include/Suffix_Array.hpp:42).
inline void effective_params(uint64_t n, uint64_t p_arg, uint32_t* p_eff, uint32_t* ppp)
{
    uint64_t p = p_arg > 0 ? p_arg : 8192;
    if (n / 16 < p) p = n / 16;
    if (p < 2) { *p_eff = (uint32_t)p; *ppp = 0; return; }
    const uint64_t a = (uint64_t)std::ceil(32.0 * std::log((double)n));
    const uint64_t b = n / p - 1;
    *p_eff = (uint32_t)p;
    *ppp = (uint32_t)(a < b ? a : b);
}

// text_bits: the code width the text arena is sized for (8: any text; 2: texts of at most 4 distinct bytes -- a quarter of it)
template <typename idx_t>
Plan<idx_t> make_plan(uint64_t n, uint64_t p_arg, char* base, int text_bits = 8)
{
    Plan<idx_t> pl;
    pl.n = n;
    pl.text_bits = text_bits;
    effective_params(n, p_arg, &pl.p, &pl.ppp);
    Arena ar;
    ar.base = base;
    const uint64_t nn = n ? n : 1;
    const bool split = pl.p >= 2;
    const uint32_t p = split ? pl.p : 1;
    // per-segment tables also serve the direct path's level B: up to DIRECT_SUB sub-streams for each of <= BUCKET_LDS groups
    const uint32_t gseg = std::max<uint32_t>(p, DIRECT_SUB * std::min<uint32_t>(p, BUCKET_LDS));
    pl.m = split ? (uint64_t)p * pl.ppp : 0;
    pl.bk.nb_cap = BucketBufs::bucket_bound(nn, p);
    pl.bk.tile_cap = nn / TILE_E + pl.bk.nb_cap + 2;
    pl.tile_cap = pl.bk.tile_cap;
    pl.P = ar.take<uint32_t>(text_alloc_words(nn, text_bits));
    for (ElemBuf<idx_t>* b : {&pl.A, &pl.B}) {
        b->key = ar.take<uint64_t>(nn);
        const size_t at = ar.off - nn * sizeof(uint64_t);
        b->sa = ar.take<idx_t>(nn);
        b->lcp = ar.take<idx_t>(nn);
        b->region_bytes = ar.off - at;
    }
    auto segs = [&](SegBufs& s, uint32_t G, uint64_t cap) {
        s.G = G;
        s.seg_start = ar.take<uint64_t>((size_t)G + 1);
        s.tile_off = ar.take<uint32_t>((size_t)G + 1);
        s.tile_rec = ar.take<TileInfo>(cap);
        s.out2 = ar.take<uint64_t>(2);
    };
    segs(pl.seg1, p, pl.tile_cap);
    if (split) {
        for (ElemBuf<idx_t>* b : {&pl.SA_, &pl.SB_}) {
            b->key = ar.take<uint64_t>(pl.m);
            b->sa = ar.take<idx_t>(pl.m);
            b->lcp = ar.take<idx_t>(pl.m);
        }
        pl.pkey = ar.take<uint64_t>(p);
        pl.psa = ar.take<idx_t>(p);
        pl.Pm = ar.take<idx_t>((size_t)p * (p + 1));
        pl.ruler = ar.take<idx_t>((size_t)p * p);
        pl.PmT = ar.take<idx_t>((size_t)p * (p + 1));
        pl.rulerT = ar.take<idx_t>((size_t)p * p);
        pl.sizes = ar.take<uint64_t>(p);
        pl.gkey = ar.take<uint64_t>(p);
        pl.dstat = ar.take<uint64_t>(4);
        pl.dcur = ar.take<idx_t>(gseg);
        pl.glut = ar.take<uint16_t>(SPLIT_LUT_CELLS + 2);
        pl.gshift = ar.take<uint8_t>(gseg);
        pl.knots = ar.take<uint64_t>(pl.bk.nb_cap);
        pl.rcap = ar.take<uint64_t>(gseg);
        pl.rstart = ar.take<uint64_t>((size_t)gseg + 1);
        pl.partial = ar.take<uint64_t>((size_t)PART_CHUNKS * p);
        segs(pl.seg2, gseg, pl.tile_cap);
        pl.seg2.G = p;
        pl.seg2.seg_end = ar.take<uint64_t>((size_t)gseg + 1);
        segs(pl.segS, 1, pl.m / TILE_E + 3);
    }
    pl.desc = ar.take<TileDesc>(pl.tile_cap + 1);
    pl.bk.params = ar.take<BucketParams>(gseg);
    pl.bk.segB = ar.take<uint64_t>(gseg);
    pl.bk.bstart = ar.take<uint64_t>((size_t)gseg + 1);
    pl.bk.count = ar.take<uint64_t>(pl.bk.nb_cap);
    pl.bk.cursor = ar.take<idx_t>(pl.bk.nb_cap);
    pl.bk.scan_tmp = ar.take<uint64_t>(2 * ((size_t)pl.bk.nb_cap / SCAN_CHUNK + 2));
    pl.bk.tile_map = ar.take<BucketParams>(pl.bk.nb_cap);
    pl.bk.first_key = ar.take<uint64_t>(pl.bk.nb_cap);
    pl.bk.last_key = ar.take<uint64_t>(pl.bk.nb_cap);
    pl.bk.first_sa = ar.take<idx_t>(pl.bk.nb_cap);
    pl.bk.last_sa = ar.take<idx_t>(pl.bk.nb_cap);
    segs(pl.bk.sub, pl.bk.nb_cap, pl.bk.tile_cap);
    pl.bk.fparams = ar.take<BucketParams>(gseg);
    pl.bk.fsegB = ar.take<uint64_t>(gseg);
    pl.bk.fstart = ar.take<uint64_t>((size_t)gseg + 1);
    pl.bk.fcount = ar.take<uint64_t>((size_t)EQ_FINE * pl.bk.nb_cap);
    pl.bk.gfirst = ar.take<uint32_t>(pl.bk.nb_cap);
    pl.bk.kshift = ar.take<uint8_t>(pl.bk.nb_cap);
    pl.bk.skip = ar.take<uint8_t>(pl.bk.nb_cap);
    pl.bk.run_list = ar.take<uint64_t>(1 + 4 * RUN_BUCKET_MAX);
    pl.pass_elems = ar.take<uint64_t>(kMaxPasses);
    pl.present = ar.take<uint32_t>(16);
    pl.lut = ar.take<uint8_t>(256);
    pl.bytes = ar.off + 256;
    return pl;
}

// Timing of one kernel family across a build (HIP events on the build's stream).
struct KernelClock {
    std::vector<std::pair<BackendEvent, BackendEvent>> spans;
    std::vector<uint64_t> elems;
};

// byte -> code preserving signed-char order (text.h).  Returns BITS.
inline int build_lut(const uint32_t present[8], uint8_t lut[256])
{
    int sigma = 0;
    for (int c = 0; c < 256; ++c) sigma += (present[c >> 5] >> (c & 31)) & 1;
    if (sigma <= 4) {
        for (int c = 0; c < 256; ++c) lut[c] = 0xFF;          // not in the alphabet (pack_kernel's validation; never read otherwise)
        int rank = 0;
        for (int v = -128; v < 128; ++v) {          // signed-char order
            const int c = v & 0xFF;
            if ((present[c >> 5] >> (c & 31)) & 1) lut[c] = (uint8_t)rank++;
        }
        return 2;
    }
    for (int c = 0; c < 256; ++c) lut[c] = (uint8_t)(c ^ 0x80);
    return 8;
}

// Launch of a kernel that exists with and without the run-table comparators (text.h): the build
// for the text prepared last (Backend::long_runs).  targs: the leading template arguments, in parentheses.
#define CAPS_UNPAREN(...) __VA_ARGS__
#define CAPS_LAUNCH_RUNS(kernel, targs, grid, block, be, ...)                                            \
    do {                                                                                                 \
        if ((be).long_runs) CAPS_LAUNCH((kernel<CAPS_UNPAREN targs, true>), grid, block, be, __VA_ARGS__); \
        else CAPS_LAUNCH((kernel<CAPS_UNPAREN targs, false>), grid, block, be, __VA_ARGS__);             \
    } while (0)

// Run table of the packed text (text.h): per-block values, then -- only if some periodic stretch
// spans two whole blocks, decided on the device -- the three-kernel suffix scan that resolves them.
inline void build_run_table(Backend& be, uint32_t* P, uint64_t n, int bits)
{
    const uint64_t entries = run_table_entries(n, bits);
    const uint32_t chunks = (uint32_t)run_table_chunks(entries);
    uint64_t* R = run_table_mut(P, n, bits);
    uint64_t* flag = R - 2;
    uint64_t* longest = R - 1;
    uint64_t* head = R + entries;
    uint64_t* carry = head + chunks;
    be.memset(flag, 0, 2 * sizeof(uint64_t));
    const uint64_t want = (entries + 255) / 256;
    const uint32_t grid = (uint32_t)(want < 16384 ? want : 16384);
    if (bits == 2) CAPS_LAUNCH(run_blocks_kernel<2>, grid, 256, be, (const uint32_t*)P, n, entries, R, flag);
    else CAPS_LAUNCH(run_blocks_kernel<8>, grid, 256, be, (const uint32_t*)P, n, entries, R, flag);
    CAPS_LAUNCH(run_chunk_heads_kernel, chunks, RUN_NT, be, (const uint64_t*)R, entries, (const uint64_t*)flag, head);
    CAPS_LAUNCH(run_carry_kernel, 1, 1024, be, (const uint64_t*)head, chunks, (const uint64_t*)flag, carry);
    CAPS_LAUNCH(run_resolve_kernel, chunks, RUN_NT, be, R, entries, (const uint64_t*)flag, (const uint64_t*)carry,
                (uint32_t)(64 / bits), longest);
    uint64_t longest_host = 0;
    be.d2h(&longest_host, longest, sizeof longest_host);
    be.sync();
    be.long_runs = longest_host >= RUN_LONG;
}

// Alphabet scan + packing of the raw device text dT into P (input
// preparation on device).  present_dev: 8 x u32, lut_dev: 256 bytes.  Returns BITS.
// arena_bits: the code width the arena at P was sized for; a text that needs more is refused (CAPS_SA_EALPHABET thrown as
// std::length_error: the caller sized the workspace for a smaller alphabet than the text has).
struct AlphabetError : std::length_error { AlphabetError() : std::length_error("the text has more than 4 distinct bytes, the workspace was sized for 2-bit codes") {} };
inline int prepare_text(Backend& be, const uint8_t* dT, uint64_t n, uint32_t* P, uint32_t* present_dev, uint8_t* lut_dev, int arena_bits = 8)
{
    // present_dev: 16 x u32 -- [0..8) the presence bits, [8] pack_kernel's "unknown byte" flag
    auto alphabet = [&](uint64_t len, uint32_t present[8]) {
        be.memset(present_dev, 0, 16 * sizeof(uint32_t));
        const uint64_t want = (len + 16 * 256 - 1) / (16 * 256);
        const uint32_t grid = (uint32_t)(want < 4096 ? (want ? want : 1) : 4096);
        CAPS_LAUNCH(alphabet_kernel, grid, 256, be, dT, len, present_dev);
        be.d2h(present, present_dev, 8 * sizeof(uint32_t));
        be.sync();
    };
    auto pack = [&](int bits, const uint8_t lut[256], uint32_t* bad) {
        be.h2d(lut_dev, lut, 256);
        const uint64_t n_words = packed_words(n, bits);
        const uint64_t want = (n_words / (bits == 2 ? 1 : 4) + 255) / 256 + 1;
        const uint32_t grid = (uint32_t)(want < 65536 ? want : 65536);
        if (bits == 2) CAPS_LAUNCH(pack_kernel<2>, grid, 256, be, dT, n, (const uint8_t*)lut_dev, P, n_words, bad);
        else CAPS_LAUNCH(pack_kernel<8>, grid, 256, be, dT, n, (const uint8_t*)lut_dev, P, n_words, bad);
    };
    uint32_t present[8];
    uint8_t lut[256];
    // Large texts: the alphabet of the first MiB is tried first -- a scan of all of T costs as much as packing it (0.68 of the
    // 1.8 ms this function takes at 3e9 bytes).  The pack kernel checks every byte against it; a byte the sample did not show
    // (the codes are ranks in the WHOLE text's alphabet) sends us to the exact path below.
#ifdef CAPS_EMUL
    constexpr uint64_t SAMPLE = 1u << 12;             // small, so that the CPU logic tests take this route (and its redo)
#else
    constexpr uint64_t SAMPLE = 1u << 20;
#endif
    if (n >= 16 * SAMPLE && !std::getenv("CAPS_SA_FULL_ALPHABET")) {
        alphabet(SAMPLE, present);
        if (build_lut(present, lut) == 2) {
            pack(2, lut, present_dev + 8);
            uint32_t bad = 0;
            be.d2h(&bad, present_dev + 8, sizeof bad);
            build_run_table(be, P, n, 2);             // ends with a sync: `bad` has arrived, lut[] may go
            if (!bad) return 2;
        }
    }
    alphabet(n, present);
    const int bits = build_lut(present, lut);
    if (bits > arena_bits) throw AlphabetError();
    pack(bits, lut, nullptr);
    build_run_table(be, P, n, bits);              // ends with a sync (lut[] lives on this stack frame)
    return bits;
}

// out[0..G] = exclusive scan of in[0..G) (out[G] = total).  Small inputs: one workgroup;
// large ones (the buckets): chunked scan over many workgroups.  tmp: u64[G / SCAN_CHUNK + 2] x 2.
template <typename OutT>
inline void device_exclusive_scan(Backend& be, const uint64_t* in, uint32_t G, OutT* out, uint64_t* tmp)
{
    const uint32_t chunks = (G + SCAN_CHUNK - 1) / SCAN_CHUNK;
    uint64_t* sums = tmp;
    uint64_t* offs = tmp + chunks + 1;
    CAPS_LAUNCH((block_scan_kernel<OutT>), chunks ? chunks : 1, 1024, be, in, (uint64_t)G, out, sums);
    CAPS_LAUNCH(scan_sizes_kernel, 1, 1024, be, (const uint64_t*)sums, chunks, offs);
    CAPS_LAUNCH((add_offsets_kernel<OutT>), chunks ? chunks : 1, 1024, be, out, (uint64_t)G, (const uint64_t*)offs, chunks);
}

#ifdef CAPS_EMUL
constexpr uint32_t kSmallScan = 48;        // tiny, so that the CPU logic tests reach the chunked scan
#else
constexpr uint32_t kSmallScan = 32768;
#endif

// use_end: the segments sit in fixed-capacity regions (s.seg_end holds their ends)
// skip: segments that get no tiles (seg_prepare_kernel)
inline void prepare_segments(Backend& be, const SegBufs& s, uint64_t tile_bound, uint64_t* big_tmp = nullptr,
                             uint64_t* big_cnt = nullptr, bool use_end = false, const uint8_t* skip = nullptr)
{
    be.memset(s.out2, 0, 2 * sizeof(uint64_t));
    const uint64_t* send = use_end ? s.seg_end : nullptr;
    if (s.G <= kSmallScan || !big_tmp) {
        CAPS_LAUNCH(seg_prepare_kernel, 1, 1024, be, (const uint64_t*)s.seg_start, send, s.G, s.tile_off, s.out2, skip);
    } else {
        CAPS_LAUNCH(tile_count_kernel, (s.G + 255) / 256, 256, be, (const uint64_t*)s.seg_start, send, s.G, big_cnt, s.out2, skip);
        device_exclusive_scan<uint32_t>(be, big_cnt, s.G, s.tile_off, big_tmp);
        CAPS_LAUNCH(tile_total_kernel, 1, 64, be, (const uint32_t*)s.tile_off, s.G, s.out2);
    }
    const uint32_t grid = (uint32_t)((tile_bound + 255) / 256);
    CAPS_LAUNCH(tile_map_kernel, grid ? grid : 1, 256, be, (const uint64_t*)s.seg_start, send, (const uint32_t*)s.tile_off, s.G, s.tile_rec);
}

inline uint32_t tiles_of(uint64_t len) { return (uint32_t)((len + TILE_E - 1) / TILE_E); }

// Result of segmented_sort: data of segment g is in buf[sel(g)], sel = passes_for(len_g) & 1
// when skip_finished, else passes & 1 for every segment (buf[0] = the tile sort's output).
template <typename idx_t> struct SortResult {
    ElemBuf<idx_t> buf[2];
    uint32_t passes = 0;
    bool skip_finished = false;
    bool unified = false;                                             // everything was gathered into buf[0]
    bool k32 = false;                                                 // sorted with 32-bit keys (boundary LCPs come from the text)
    bool failed = false;                                              // k32 only: a slot overflowed, nothing was sorted
    SegBufs segs;                                                     // the segments that were sorted (buckets, if bucketed)
    // letter-run buckets (text.h "letter runs"): they sat out the sort; finalize() orders them by their run keys
    struct RunBucket { uint64_t s0, s1, key; };
    std::vector<RunBucket> run_buckets;
    SegBufs run_tables;                                               // segment tables for their sort (the bucket tables: idle by then)
    const BucketBufs* run_bk = nullptr;
    TileDesc* run_desc = nullptr;
    uint64_t total = 0;                                               // elements of the whole sort (= end of the result arrays)
    // deferred ties (kernels.h "Deferred ties"): flags of the sorted segments that hold sentinel LCPs; finalize() orders those
    // groups by re-keying them deeper (msd_refine) in the idle work buffers.  msd_failed: the groups did not fit there (finalize())
    uint64_t* defer_flags = nullptr;
    TileDesc* defer_desc = nullptr;
    uint64_t cap = 0;                                                 // elements buf[0] / buf[1] hold
    mutable bool msd_failed = false;
    // clocks of finalize()'s three parts (null: untimed): gather + head LCPs, letter-run buckets, deferred ties
    KernelClock* finish_clock = nullptr;
    KernelClock* runb_clock = nullptr;
    KernelClock* msd_clock = nullptr;
    mutable uint64_t msd_groups = 0, msd_elems = 0;
    mutable uint32_t msd_levels = 0;
    uint32_t n_tiles = 0;
    FinalOut<idx_t> fin;                                              // direct final output (may be empty)
    ElemBuf<idx_t> uniform() const { return unified ? buf[0] : buf[passes & 1]; }   // valid when unified or !skip_finished
    PingPong<idx_t> pingpong() const
    {
        PingPong<idx_t> pp;
        for (int i = 0; i < 2; ++i) { pp.key[i] = buf[i].key; pp.sa[i] = buf[i].sa; pp.lcp[i] = buf[i].lcp; }
        return pp;
    }
};

// Options of one segmented sort.
struct SortOpts {
    bool from_text = false;       // elements are consecutive text positions (keys cut from the packed text)
    uint64_t text_base = 0;       // element i of the arrays = text position text_base + i
    bool need_lcp = false;        // emit LCPs (by the step that completes each segment)
    bool skip_finished = false;   // finished segments sit out later passes (result spread over both buffers)
    bool no_equalise = false;     // measurement: plain count split on the redo path (CAPS_SA_NO_EQUALISE=1)
    const BucketBufs* bk = nullptr;   // non-null: split long segments into key-range buckets first
    uint32_t range_mode = 0;      // bucket_plan_kernel: 0 full key range, 1 between pivots
    const uint64_t* pkey = nullptr;
    uint32_t part_off = 0, part_total = 0;    // range_mode 1: segment g = partition part_off + g of part_total
    bool speculate = false;       // try the bucket split without its count pass first (slots carved from `oth`)
    uint32_t* slot_stats = nullptr;   // [2] host counters: splits done with slots / redone with the count pass
    bool seg_ends = false;        // the segments sit in fixed-capacity regions: their ends are s.seg_end (direct path, level B)
    uint32_t sub = 1;             // > 1: every `sub` consecutive segments are sub-streams of one parent and share its buckets
    const uint64_t* in_key = nullptr;   // non-null: the elements are read from these arrays (indexed like the segments)
    const void* in_sa = nullptr;        //   instead of `cur`, which then only receives results
    const uint8_t* gshift = nullptr;    // non-null with range_mode 2: 32-bit keys (text.h key32_of), the parents' shifts
    bool k32 = false;                   // the elements (in_key, read as uint32_t) carry 32-bit keys; slots only: when a slot
                                        //   overflows the sort gives up (SortResult::failed) and the caller falls back to 64-bit keys
    bool keys_only = false;             // order by key, equal keys by index: no comparison reads the text (the samples of the direct
                                        //   path: only their keys are used -- pivots, knots -- and a text with long N-blocks has
                                        //   millions of samples with one key).  Done by sorting as if the text had length 0: every
                                        //   comparator settles a pair with an index beyond the text at once, by index (text.h)
    bool comparison_only = false;       // every tile straight to the comparison sort (tile_sort_general_kernel): the levels of msd_refine
    bool skewed_keys = false;           // the keys are far from uniform inside the buckets (skew probe): every tile goes to
                                        //   tile_sort_eq_kernel straight away, tile_sort_kernel's linear map is not tried
    const uint64_t* knots = nullptr;    // non-null (quantile mode): parent q's buckets are (knots[q * KPG + i - 1], knots[q * KPG + i]],
    uint32_t knots_per_parent = 0;      //   i < KPG = knots_per_parent -- count pass + exact scatter, no slots, no equalising
    bool knots_have_prev = false;       // knots[-1] exists: the slice of a shard that does not own the first group
    bool spill_slots = false;           // split by knots WITHOUT the count pass (needs final_sa / final_lcp: the spill stream borrows
                                        //   them until the tile sort writes results there): slots over `oth`, what does not fit a
                                        //   slot on the stream, the buckets that outgrew their slots put together in `cur`
    TileInfo* hot_tile_rec = nullptr;   //   room for a second tile table (as many records as the bucket tables' tile capacity)
    uint64_t* spill_stats = nullptr;    // [3] host counters: splits by knots done with slots / redone with the count pass, stream entries
    uint64_t in_extent = 0;             // one past the largest element index of in_key / in_sa (0: unknown).  Quantile splits then
                                        //   keep every element's bucket id between the count pass and the scatter (u16 per element, in
                                        //   the idle LCP array of the scatter's destination) instead of searching the knots twice
    const SegBufs* run_tables = nullptr;   // segment tables (one segment, tiles of the largest bucket) for the sorts of letter-run buckets
    uint64_t* defer_flags = nullptr;    // non-null (final sorts with direct output only): ties are DEFERRED (kernels.h "Deferred ties") --
                                        //   the comparison sort of a tile and the merge passes order equal keys without reading the
                                        //   text, the LCPs of such neighbours are sentinels, defer_flags[sorted segment] = 1 where
                                        //   there are some (u64 per sorted segment), and finalize() settles the groups (msd_refine)
    const void* runs = nullptr;   // RunSrc<idx_t>*: the segments are partitions still spread over the sorted subarrays in
                                  //   `cur` (requires the bucket split: bk != null and max_len > TILE_E)
    bool unify = false;           // gather the result into buf[0] (consumers that index whole segments)
    void* final_sa = nullptr;     // non-null (with need_lcp): completed segments are written straight to the
    void* final_lcp = nullptr;    //   caller's SA / LCP arrays; boundary records in `bnd` (6 arrays of G entries)
    struct { uint64_t *first_key, *last_key; void *first_sa, *last_sa; } bnd = {nullptr, nullptr, nullptr, nullptr};
    KernelClock* tile_clock = nullptr;
    KernelClock* merge_clock = nullptr;
    KernelClock* scatter_clock = nullptr;
    KernelClock* count_clock = nullptr;
    uint64_t* pass_counters = nullptr;
};

// Sort every segment of `s`:  [bucket split]  ->  tile sort  ->  LCP-merge passes.
//   not bucketed: input (unless from_text) is in `cur`, sorted in place, passes ping-pong cur <-> oth;
//   bucketed:     the buckets become the segments (result.segs).  Slot split (o.speculate, `oth` large enough):
//                 input -> fixed-capacity slots viewed over `oth` -> tile sort -> compact in `cur`.
//                 Count split (a slot overflowed, or no speculation): input (unless from_text) is in `cur`,
//                 scattered by bucket into `oth`, which then plays the role of `cur`.
//                 o.runs: the input elements are still spread over the sorted subarrays in `cur` (phase 2).
// n_tiles / max_len describe `s` (host-known for phase 1 and the samples, read back for phase 2).
template <typename idx_t, int BITS>
SortResult<idx_t> segmented_sort(Backend& be, const uint32_t* P, uint64_t n, TileDesc* desc, const SegBufs& s, uint32_t n_tiles,
                                 uint64_t max_len, ElemBuf<idx_t> cur, ElemBuf<idx_t> oth, uint64_t n_elems, const SortOpts& o)
{
    if (o.keys_only) {
        if (o.from_text || o.need_lcp || o.runs) throw std::invalid_argument("a keys-only sort takes (key, index) arrays and emits no LCPs");
        n = 0;
    }
    SortResult<idx_t> r;
    r.buf[0] = cur;
    r.buf[1] = oth;
    r.skip_finished = o.skip_finished;
    r.segs = s;
    r.n_tiles = n_tiles;
    if (n_tiles == 0) return r;
    const bool dbg = std::getenv("CAPS_SA_DEBUG") != nullptr;
    auto mark = [&](const char* what) { if (dbg) { be.sync(); std::fprintf(stderr, "[sort] %s\n", what); } };
    bool from_text = o.from_text;
    const BucketParams* seg_map = nullptr;                        // bin maps of the sorted segments, if bucketed
    SegBufs segs = s;
    bool skip = o.skip_finished;
    uint32_t slot_cap = 0;                           // != 0: the tile sort's input sits in fixed-capacity slots
    uint64_t* slot_key = nullptr;
    idx_t* slot_sa = nullptr;
    if (o.bk && (max_len > TILE_E || o.seg_ends)) {      // segments in regions are always bucketed: results are written compactly
        const BucketBufs& bk = *o.bk;
        const SegDesc psd = s.desc();
        const uint64_t* s_end = o.seg_ends ? (const uint64_t*)s.seg_end : (const uint64_t*)nullptr;
        const bool by_knots = o.knots != nullptr;
        if (by_knots) {
            const uint64_t NB = (uint64_t)(s.G / o.sub) * o.knots_per_parent;
            if (NB > bk.nb_cap) throw std::invalid_argument("more knot buckets than the bucket tables hold");
            CAPS_LAUNCH(knot_plan_kernel, (s.G + 255) / 256, 256, be, s.G, o.sub, o.knots_per_parent, bk.params, bk.segB);
            CAPS_LAUNCH(scan_sizes_kernel, 1, 1024, be, (const uint64_t*)bk.segB, s.G, bk.bstart);
            CAPS_LAUNCH(knot_ranges_kernel, (uint32_t)((NB + 255) / 256), 256, be, o.knots, NB, bk.tile_map, o.knots_have_prev ? 1u : 0u);
        } else {
        CAPS_LAUNCH(bucket_plan_kernel, (s.G + 255) / 256, 256, be, (const uint64_t*)s.seg_start, s_end, s.G, o.range_mode, o.pkey, o.part_off,
                    o.part_total ? o.part_total : s.G, 1u, 1u, o.sub,
                    bk.params, bk.segB, o.gshift);
        CAPS_LAUNCH(scan_sizes_kernel, 1, 1024, be, (const uint64_t*)bk.segB, s.G, bk.bstart);
        CAPS_LAUNCH(bucket_ranges_kernel, (bk.nb_cap + 255) / 256, 256, be, (const uint64_t*)bk.bstart, s.G,
                    (const BucketParams*)bk.params, o.pkey, o.range_mode, o.part_off, o.part_total ? o.part_total : s.G, o.sub,
                    bk.tile_map, (const BucketParams*)nullptr, (const uint32_t*)nullptr, o.gshift, bk.kshift);
        }
        seg_map = bk.tile_map;
        mark("bucket plan");
        const uint32_t pgrid = n_tiles < be.persistent_blocks() ? n_tiles : be.persistent_blocks();
        const bool runs = o.runs != nullptr;             // phase 2 reading the sorted subarrays through the partition matrix
        const RunSrc<idx_t> rsrc = runs ? *static_cast<const RunSrc<idx_t>*>(o.runs) : RunSrc<idx_t>();
        const uint64_t n_words = from_text ? packed_words(n, BITS) : 0;
        const uint64_t tbase = from_text ? o.text_base : 0;
        // where the elements are read from (SRC_ARRAYS / SRC_RUNS): `cur`, unless the caller keeps them elsewhere
        const uint64_t* src_key = o.in_key ? o.in_key : (const uint64_t*)cur.key;
        const idx_t* src_sa = o.in_key ? static_cast<const idx_t*>(o.in_sa) : (const idx_t*)cur.sa;
        // bucket ids kept between count and scatter (quantile splits of arrays): in the LCP array of the scatter's destination
        // (`oth`: nothing writes LCPs there before the scatter has read the ids back)
        uint16_t* bid = nullptr;
        if (by_knots && o.in_extent && !from_text && !o.runs && oth.region_bytes && !std::getenv("CAPS_SA_NO_BUCKET_IDS")) {
            const size_t room = (size_t)((reinterpret_cast<char*>(oth.key) + oth.region_bytes) - reinterpret_cast<char*>(oth.lcp));
            const uint64_t extent = o.in_key ? o.in_extent : n_elems;
            if (room >= extent * sizeof(uint16_t)) bid = reinterpret_cast<uint16_t*>(oth.lcp);
        }
        auto scatter = [&](const uint64_t* sub_start, uint32_t cap, uint64_t* okey, idx_t* osa, bool grouped = false) {
            const BucketParams* fbps = grouped ? bk.fparams : nullptr;
            const uint32_t* gfirst = grouped ? bk.gfirst : nullptr;
            BackendEvent s0 = be.record();
#define CAPS_SCATTER_LAUNCH(SRC_, MAP_, ikey, isa)                                                                          \
            CAPS_LAUNCH((bucket_scatter_kernel<idx_t, BITS, SRC_, MAP_>), n_tiles, TILE_NT, be, psd, P, n_words, tbase, ikey, isa, rsrc, \
                        (const BucketParams*)bk.params, (const uint64_t*)bk.bstart, sub_start, (uint64_t)cap, static_cast<idx_t*>(bk.cursor),       \
                        okey, osa, fbps, gfirst, o.knots, (const uint16_t*)nullptr, (const uint32_t*)nullptr, 1u, o.knots_per_parent, o.sub, \
                        (const uint16_t*)bid, Spill<idx_t>())
            const uint64_t* nokey = nullptr;
            const idx_t* nosa = nullptr;
            if (from_text) { if (grouped) CAPS_SCATTER_LAUNCH(SRC_TEXT, MAP_GROUPED, nokey, nosa); else CAPS_SCATTER_LAUNCH(SRC_TEXT, MAP_LINEAR, nokey, nosa); }
            else if (runs) { if (grouped) CAPS_SCATTER_LAUNCH(SRC_RUNS, MAP_GROUPED, src_key, src_sa);
                             else CAPS_SCATTER_LAUNCH(SRC_RUNS, MAP_LINEAR, src_key, src_sa); }
            else if (by_knots) CAPS_SCATTER_LAUNCH(SRC_ARRAYS, MAP_SPLIT, src_key, src_sa);
            else { if (grouped) CAPS_SCATTER_LAUNCH(SRC_ARRAYS, MAP_GROUPED, src_key, src_sa);
                   else CAPS_SCATTER_LAUNCH(SRC_ARRAYS, MAP_LINEAR, src_key, src_sa); }
#undef CAPS_SCATTER_LAUNCH
            BackendEvent s1 = be.record();
            if (o.scatter_clock) { o.scatter_clock->spans.push_back({s0, s1}); o.scatter_clock->elems.push_back(n_elems); }
        };
        // the buckets become the segments (trailing unused ones are empty): sizes -> offsets -> tiles
        uint64_t out2[2] = {0, 0};
        auto adopt_buckets = [&]() {
            device_exclusive_scan<uint64_t>(be, bk.count, bk.nb_cap, bk.sub.seg_start, bk.scan_tmp);
        };
        // letter-run buckets (quantile splits of a final sort only): marked on the device, they get no tiles -- the tile sort
        // and the merge passes never see them -- and finalize() orders them by their run keys (text.h "letter runs")
        const bool run_buckets = by_knots && o.need_lcp && o.final_sa && bk.skip && !std::getenv("CAPS_SA_NO_RUN_BUCKETS") &&
                                 n < (1ull << RunKey<(int)sizeof(idx_t)>::RB);
        auto bucket_tiles = [&]() {
            segs = bk.sub;
            segs.G = bk.nb_cap;
            uint64_t rl[1 + 4 * RUN_BUCKET_MAX];
            if (run_buckets) {
                const uint64_t NB = (uint64_t)(s.G / o.sub) * o.knots_per_parent;
                be.memset(bk.skip, 0, bk.nb_cap);
                be.memset(bk.run_list, 0, sizeof(uint64_t));
                CAPS_LAUNCH(run_bucket_mark_kernel, (uint32_t)((NB + 255) / 256), 256, be, o.knots, NB, o.knots_have_prev ? 1u : 0u, (uint32_t)BITS,
                            (const uint64_t*)bk.sub.seg_start, bk.skip, bk.run_list);
            }
            prepare_segments(be, segs, bk.tile_cap, bk.scan_tmp, bk.count, false, run_buckets ? bk.skip : nullptr);   // count[] is free again: scratch
            be.d2h(out2, segs.out2, sizeof out2);
            if (run_buckets) be.d2h(rl, bk.run_list, sizeof rl);
            be.sync();
            if (run_buckets) {
                const uint64_t cnt = rl[0] < RUN_BUCKET_MAX ? rl[0] : RUN_BUCKET_MAX;
                for (uint64_t j = 0; j < cnt; ++j) r.run_buckets.push_back({rl[2 + 4 * j], rl[3 + 4 * j], rl[4 + 4 * j]});
                std::sort(r.run_buckets.begin(), r.run_buckets.end(),
                          [](const typename SortResult<idx_t>::RunBucket& a, const typename SortResult<idx_t>::RunBucket& b) { return a.s0 < b.s0; });
                // tables for the sort of every such bucket: the caller's (o.run_tables), else the buckets' own -- idle once finalize()
                // has gathered the sorted buckets, but then nothing may read the bucket tables after the run buckets have been
                // sorted (msd_refine does: sorts that defer ties pass tables of their own)
                r.run_tables = o.run_tables ? *o.run_tables : bk.sub;
                r.run_bk = &bk;
                r.run_desc = desc;
                r.total = n_elems;
            }
        };
        // ---- speculative split: no count pass.  Bucket i gets the fixed slot [i * TILE_E, (i + 1) * TILE_E) of a
        // slot buffer viewed over `oth`; the scatter's cursors end as the exact bucket sizes.  If every bucket fits
        // its slot (keys roughly uniform inside their ranges: random DNA) the tile sort reads the slots and writes
        // the buckets compactly into `cur`; otherwise the split is redone below with the count pass.
        // buckets this split can have: at most len / BUCKET_TARGET + 2 per parent segment (bucket_plan_kernel)
        const uint64_t nb_here = std::min<uint64_t>(bk.nb_cap, n_elems / BUCKET_TARGET + 2 * (uint64_t)(s.G / o.sub) + 2);
        const uint64_t slot_elems = nb_here * TILE_E;
        bool slots = false;
        if (by_knots && (from_text || runs)) throw std::invalid_argument("quantile buckets read (key, sa) arrays");
        if (o.k32 && (from_text || runs || by_knots || !o.speculate || !o.in_key || !o.final_sa))
            throw std::invalid_argument("32-bit keys: a slot split of (key32, sa) arrays with direct output");
        const bool can_slot = o.speculate && !by_knots &&
                              oth.region_bytes >= slot_elems * ((o.k32 ? sizeof(uint32_t) : sizeof(uint64_t)) + sizeof(idx_t)) &&
                              slot_elems + TILE_E < (uint64_t)std::numeric_limits<idx_t>::max();
        if (o.k32 && !can_slot) { r.failed = true; return r; }          // 32-bit keys exist in slots only
        if (can_slot) {
            slot_key = oth.key;
            slot_sa = reinterpret_cast<idx_t*>(reinterpret_cast<char*>(oth.key) + slot_elems * sizeof(uint64_t));
            be.memset(bk.cursor, 0, (size_t)bk.nb_cap * sizeof(idx_t));
            if (o.k32) {             // 32-bit keys in and out: the slot view holds (u32 key | idx) per element
                slot_sa = reinterpret_cast<idx_t*>(reinterpret_cast<char*>(oth.key) + slot_elems * sizeof(uint32_t));
                BackendEvent s0 = be.record();
                CAPS_LAUNCH((bucket_scatter_kernel<idx_t, BITS, SRC_ARRAYS, MAP_LINEAR, uint32_t>), n_tiles, TILE_NT, be, psd, P, (uint64_t)0,
                            (uint64_t)0, reinterpret_cast<const uint32_t*>(o.in_key), src_sa, rsrc, (const BucketParams*)bk.params,
                            (const uint64_t*)bk.bstart, (const uint64_t*)nullptr, (uint64_t)TILE_E, static_cast<idx_t*>(bk.cursor),
                            reinterpret_cast<uint32_t*>(slot_key), slot_sa, (const BucketParams*)nullptr, (const uint32_t*)nullptr,
                            (const uint64_t*)nullptr, (const uint16_t*)nullptr, (const uint32_t*)nullptr, 1u, 0u, 1u, (const uint16_t*)nullptr,
                            Spill<idx_t>());
                BackendEvent s1 = be.record();
                if (o.scatter_clock) { o.scatter_clock->spans.push_back({s0, s1}); o.scatter_clock->elems.push_back(n_elems); }
            } else
            scatter(nullptr, TILE_E, slot_key, slot_sa);
            CAPS_LAUNCH((widen_kernel<idx_t>), (bk.nb_cap + 255) / 256, 256, be, (const idx_t*)static_cast<idx_t*>(bk.cursor),
                        (uint64_t)bk.nb_cap, bk.count);
            adopt_buckets();
            bucket_tiles();
            slots = out2[1] <= TILE_E;
            if (o.slot_stats) ++o.slot_stats[slots ? 0 : 1];
            if (o.k32 && !slots) { r.failed = true; return r; }
            if (dbg) std::fprintf(stderr, "[sort] slot split: largest bucket %llu -> %s\n", (unsigned long long)out2[1], slots ? "kept" : "redone");
        }
        // ---- speculative split by knots (quantile mode, level B of the direct path): slots + a spill stream, no count pass.
        // Quantile buckets average BUCKET_Q = 3/4 of a tile with ~12 % spread: all but a percent fit a slot of nearly
        // TILE_E; the rest (and the buckets of repeats, any size) outgrow it, and what they do not get into the slot goes to
        // the stream (bucket_scatter_kernel SPILL).  Afterwards the cursors are the exact sizes as after a count pass, the
        // outgrown buckets are put together at their places in `cur` (spill_gather_kernel, spill_place_kernel) and the tile
        // sort reads a bucket from its slot or, outgrown, from `cur` in place (kernels.h tile_src).  The stream lives in the
        // caller's SA / LCP slice of this sort, idle until the tile sort emits.  A stream that runs full (a text that is
        // mostly repeats): the count split below, as before.
        if (by_knots && o.spill_slots && o.hot_tile_rec && !from_text && !runs && o.final_sa && o.final_lcp && o.need_lcp &&
            !std::getenv("CAPS_SA_NO_SPILL_SLOTS")) {
            const uint64_t NB = (uint64_t)(s.G / o.sub) * o.knots_per_parent;
            const uint64_t idx_max = (uint64_t)std::numeric_limits<idx_t>::max() - TILE_E - 1;
            // the stream: keys in the LCP slice, (index, bucket, position) in the SA slice
            char* lcp_lo = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(o.final_lcp) + 15) & ~uintptr_t(15));
            char* lcp_hi = reinterpret_cast<char*>(static_cast<idx_t*>(o.final_lcp) + n_elems);
            char* sa_lo = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(o.final_sa) + 15) & ~uintptr_t(15));
            char* sa_hi = reinterpret_cast<char*>(static_cast<idx_t*>(o.final_sa) + n_elems);
            uint64_t scap = 0;
            if (lcp_hi > lcp_lo + 64 && sa_hi > sa_lo + 64)
                scap = std::min<uint64_t>((uint64_t)(lcp_hi - lcp_lo) / sizeof(uint64_t), (uint64_t)(sa_hi - sa_lo - 32) / (2 * sizeof(idx_t) + sizeof(uint32_t)));
            const char* cap_env = std::getenv("CAPS_SA_TEST_SPILL_CAP");          // tests: a stream that runs full
            if (cap_env) scap = std::min<uint64_t>(scap, std::strtoull(cap_env, nullptr, 10));
            // slot capacity: a tile, unless `oth` or the index type has less room (then a little less: still well above the mean)
            uint64_t cap_s = TILE_E;
            if (NB) cap_s = std::min<uint64_t>(cap_s, oth.region_bytes / (sizeof(uint64_t) + sizeof(idx_t)) / NB);
            const uint64_t want_stream = std::min<uint64_t>(scap, n_elems / 8 + 1024);
            if (NB && idx_max > want_stream) cap_s = std::min<uint64_t>(cap_s, (idx_max - want_stream) / NB);
            cap_s &= ~uint64_t(31);
            const char* slot_env = std::getenv("CAPS_SA_TEST_SPILL_SLOT");        // tests: tiny slots (everything outgrows them)
            if (slot_env) cap_s = std::min<uint64_t>(cap_s, std::strtoull(slot_env, nullptr, 10));
            const uint64_t slot_total = NB * cap_s;
            if (slot_total < idx_max) scap = std::min<uint64_t>(scap, idx_max - slot_total);
            const bool fits = NB != 0 && NB <= bk.nb_cap && o.knots_per_parent <= BUCKET_LDS && (cap_env ? scap >= 1 : scap >= 1024) && slot_total < idx_max &&
                              (slot_env || cap_s >= (BUCKET_Q / 6) * 7) &&
                              oth.region_bytes >= slot_total * (sizeof(uint64_t) + sizeof(idx_t)) && bk.run_list;
            if (fits) {
                slot_key = oth.key;
                slot_sa = reinterpret_cast<idx_t*>(reinterpret_cast<char*>(oth.key) + slot_total * sizeof(uint64_t));
                Spill<idx_t> sp;
                sp.count = bk.run_list;                  // (the list of letter-run buckets is made after the stream's count has been read)
                sp.cap = scap;
                sp.base = slot_total;
                sp.key = reinterpret_cast<uint64_t*>(lcp_lo);
                sp.sa = reinterpret_cast<idx_t*>(sa_lo);
                sp.rel = sp.sa + scap;
                sp.bucket = reinterpret_cast<uint32_t*>(sp.rel + scap);
                // every tile's own chunk at the head of the stream (kernels.h Spill::chunk), the counter behind them
                const uint64_t fixed = (uint64_t)n_tiles * SPILL_CHUNK <= scap / 4 && !std::getenv("CAPS_SA_NO_SPILL_CHUNKS") ? (uint64_t)n_tiles * SPILL_CHUNK : 0;
                sp.chunk = fixed ? SPILL_CHUNK : 0u;
                be.memset(bk.cursor, 0, (size_t)bk.nb_cap * sizeof(idx_t));
                be.h2d(sp.count, &fixed, sizeof(uint64_t));
                if (fixed) be.memset(sp.bucket, 0xFF, (size_t)fixed * sizeof(uint32_t));
                BackendEvent s0 = be.record();
                CAPS_LAUNCH((bucket_scatter_kernel<idx_t, BITS, SRC_ARRAYS, MAP_SPLIT, uint64_t, true>), n_tiles, TILE_NT, be, psd, P, (uint64_t)0,
                            (uint64_t)0, src_key, src_sa, rsrc, (const BucketParams*)bk.params, (const uint64_t*)bk.bstart,
                            (const uint64_t*)nullptr, cap_s, static_cast<idx_t*>(bk.cursor), slot_key, slot_sa,
                            (const BucketParams*)nullptr, (const uint32_t*)nullptr, o.knots, (const uint16_t*)nullptr,
                            (const uint32_t*)nullptr, 1u, o.knots_per_parent, o.sub, (const uint16_t*)nullptr, sp);
                uint64_t spilled = 0;
                be.d2h(&spilled, sp.count, sizeof spilled);
                CAPS_LAUNCH((widen_kernel<idx_t>), (bk.nb_cap + 255) / 256, 256, be, (const idx_t*)static_cast<idx_t*>(bk.cursor),
                            (uint64_t)bk.nb_cap, bk.count);
                adopt_buckets();
                bucket_tiles();                          // (syncs: `spilled` is here)
                slots = spilled <= scap;
                if (o.spill_stats) { ++o.spill_stats[slots ? 0 : 1]; o.spill_stats[2] += spilled; }
                if (dbg) std::fprintf(stderr, "[sort] split by knots with slots of %llu: %llu on the stream (room for %llu), largest bucket %llu -> %s\n",
                                      (unsigned long long)cap_s, (unsigned long long)spilled, (unsigned long long)scap, (unsigned long long)out2[1],
                                      slots ? "kept" : "redone");
                if (slots) {
                    const uint32_t ggrid_ = (uint32_t)std::min<uint64_t>((NB + 255) / 256, 16384);
                    CAPS_LAUNCH((spill_gather_kernel<idx_t>), ggrid_, 256, be, (const uint64_t*)bk.sub.seg_start, (uint32_t)NB, cap_s,
                                run_buckets ? (const uint8_t*)bk.skip : (const uint8_t*)nullptr, (const uint64_t*)slot_key, (const idx_t*)slot_sa,
                                cur.key, cur.sa);
                    if (spilled)
                        CAPS_LAUNCH((spill_place_kernel<idx_t>), (uint32_t)std::min<uint64_t>((spilled + 255) / 256, 16384), 256, be, sp, spilled,
                                    (const uint64_t*)bk.sub.seg_start, cur.key, cur.sa);
                    BackendEvent s1 = be.record();
                    if (o.scatter_clock) { o.scatter_clock->spans.push_back({s0, s1}); o.scatter_clock->elems.push_back(n_elems); }
                    slot_cap = (uint32_t)cap_s;
                } else {
                    BackendEvent s1 = be.record();
                    if (o.scatter_clock) { o.scatter_clock->spans.push_back({s0, s1}); o.scatter_clock->elems.push_back(n_elems); }
                }
            }
        }
        if (!slots) {
            // Count split.  Equalised (bk.fcount): the count pass fills EQ_FINE times finer buckets, bucket_group_kernel
            // packs consecutive fine buckets into the segment's bucket slots (exact sizes, balanced whatever the key
            // distribution inside the segment), and the scatter looks the slot up from the fine bucket.
            const bool equalise = !by_knots && bk.fcount != nullptr && !o.no_equalise && !std::getenv("CAPS_SA_NO_EQUALISE");
            const BucketParams* cparams = bk.params;
            const uint64_t* cstart = bk.bstart;
            uint64_t* ccount = bk.count;
            if (equalise) {
                CAPS_LAUNCH(bucket_plan_kernel, (s.G + 255) / 256, 256, be, (const uint64_t*)s.seg_start, s_end, s.G, o.range_mode, o.pkey, o.part_off,
                            o.part_total ? o.part_total : s.G, 1u, EQ_FINE, o.sub, bk.fparams, bk.fsegB, o.gshift);
                CAPS_LAUNCH(scan_sizes_kernel, 1, 1024, be, (const uint64_t*)bk.fsegB, s.G, bk.fstart);
                be.memset(bk.fcount, 0, (size_t)EQ_FINE * bk.nb_cap * sizeof(uint64_t));
                cparams = bk.fparams;
                cstart = bk.fstart;
                ccount = bk.fcount;
            }
            be.memset(bk.count, 0, (size_t)bk.nb_cap * sizeof(uint64_t));
            be.memset(bk.cursor, 0, (size_t)bk.nb_cap * sizeof(idx_t));
            BackendEvent c0 = be.record();
            const uint64_t* no_tab = nullptr;
            if (from_text)
                CAPS_LAUNCH((bucket_count_kernel<idx_t, BITS, SRC_TEXT>), pgrid, TILE_NT, be, psd, P, n_words, tbase, (const uint64_t*)nullptr, rsrc,
                            cparams, cstart, ccount, no_tab, 0u, 1u, (uint16_t*)nullptr);
            else if (runs)
                CAPS_LAUNCH((bucket_count_kernel<idx_t, BITS, SRC_RUNS>), pgrid, TILE_NT, be, psd, P, n_words, tbase, src_key, rsrc,
                            cparams, cstart, ccount, no_tab, 0u, 1u, (uint16_t*)nullptr);
            else if (by_knots)
                CAPS_LAUNCH((bucket_count_kernel<idx_t, BITS, SRC_ARRAYS, MAP_SPLIT>), pgrid, TILE_NT, be, psd, P, n_words, tbase, src_key, rsrc,
                            cparams, cstart, ccount, o.knots, o.knots_per_parent, o.sub, bid);
            else
                CAPS_LAUNCH((bucket_count_kernel<idx_t, BITS, SRC_ARRAYS>), pgrid, TILE_NT, be, psd, P, n_words, tbase, src_key, rsrc,
                            cparams, cstart, ccount, no_tab, 0u, 1u, (uint16_t*)nullptr);
            if (equalise) {
                CAPS_LAUNCH(bucket_group_kernel, s.G < 16384 ? (s.G ? s.G : 1) : 16384, 256, be, s.G, (const uint64_t*)bk.segB, (const uint64_t*)bk.bstart,
                            (const uint64_t*)bk.fsegB, (const uint64_t*)bk.fstart, (const uint64_t*)bk.fcount, bk.count, bk.gfirst);
                CAPS_LAUNCH(bucket_ranges_kernel, (bk.nb_cap + 255) / 256, 256, be, (const uint64_t*)bk.bstart, s.G,
                            (const BucketParams*)bk.params, o.pkey, o.range_mode, o.part_off, o.part_total ? o.part_total : s.G, o.sub,
                            bk.tile_map, (const BucketParams*)bk.fparams, (const uint32_t*)bk.gfirst, o.gshift, bk.kshift);
            }
            BackendEvent c1 = be.record();
            if (o.count_clock) { o.count_clock->spans.push_back({c0, c1}); o.count_clock->elems.push_back(n_elems); }
            adopt_buckets();
            ElemBuf<idx_t> dst = from_text ? cur : oth;
            scatter((const uint64_t*)bk.sub.seg_start, 0u, dst.key, dst.sa, equalise);
            if (!from_text) {
                std::swap(cur, oth);                     // the scattered copy is the working buffer now
                r.buf[0] = cur;
                r.buf[1] = oth;
            }
            bucket_tiles();
        } else if (!slot_cap) {
            slot_cap = TILE_E;                           // the tile sort reads the slots, writes `cur`
        }
        mark("bucket scatter");
        n_tiles = (uint32_t)out2[0];
        max_len = out2[1];
        if (dbg) std::fprintf(stderr, "[sort] buckets: tiles %u max_len %llu\n", n_tiles, (unsigned long long)max_len);
        from_text = false;
        skip = true;
        r.skip_finished = true;
        r.segs = segs;
        r.n_tiles = n_tiles;
    }
    const SegDesc sd = segs.desc();
    // the kernels at the head of the queue chain (tile_sort_kernel, the plain builds of tile_sort_eq_kernel) get a tile table in which
    // the tiles of outgrown buckets are empty (kernels.h hot_tiles_kernel): they pass those on, the rest of the chain has `sd`
    SegDesc sd_hot = sd;
    if (slot_cap && max_len > slot_cap && o.hot_tile_rec) {
        CAPS_LAUNCH(hot_tiles_kernel, (n_tiles + 255) / 256, 256, be, sd, slot_cap, o.hot_tile_rec);
        sd_hot.tile_rec = o.hot_tile_rec;
    } else if (slot_cap && max_len > slot_cap) {
        throw std::invalid_argument("slots with outgrown buckets need room for the second tile table");
    }
    const uint32_t lcp_mode = o.need_lcp ? 1u : 0u;
    FinalOut<idx_t> fin;
    if (o.need_lcp && o.final_sa) {
        fin.sa = static_cast<idx_t*>(o.final_sa);
        fin.lcp = static_cast<idx_t*>(o.final_lcp);
        fin.first_key = o.bnd.first_key;
        fin.last_key = o.bnd.last_key;
        fin.first_sa = static_cast<idx_t*>(o.bnd.first_sa);
        fin.last_sa = static_cast<idx_t*>(o.bnd.last_sa);
    }
    r.fin = fin;
    // queues of unfinished tiles ([0] = length) in the (idle) tile descriptors' memory: tile_sort_kernel -> redo ->
    // tile_sort_eq_kernel -> redo2 -> tile_sort_general_kernel
    // ... -> redo2 -> tile_sort_eq_kernel<VDEEP> (the build with the third tie stage: exact long duplicates) -> redo3 -> general
    uint32_t* redo = reinterpret_cast<uint32_t*>(desc);
    uint32_t* redo2 = redo + ((size_t)n_tiles + 2);
    uint32_t* redo3 = redo2 + ((size_t)n_tiles + 2);
    static_assert(sizeof(TileDesc) >= 3 * sizeof(uint32_t) + 1, "three queues fit the descriptor array");
    be.memset(redo, 0, sizeof(uint32_t));
    be.memset(redo2, 0, sizeof(uint32_t));
    be.memset(redo3, 0, sizeof(uint32_t));
    const uint32_t ggrid = n_tiles < 4 * be.persistent_blocks() ? n_tiles : 4 * be.persistent_blocks();
    const bool eq_tiles = !std::getenv("CAPS_SA_NO_EQ_TILES");        // measurement: skip tile_sort_eq_kernel
    // tests: tile_sort_eq_kernel loses the lcp notes of its tied pairs on purpose -- its emit phase must notice and pass the tile on
    const uint32_t drop = std::getenv("CAPS_SA_TEST_DROP_NOTE") ? 2u : 0u;
    if (!eq_tiles) redo3 = redo2 = redo;
    // deferred ties: only where the sort writes THE arrays (sentinels must not reach a consumer that reads LCPs as numbers)
    uint64_t* bflag = o.defer_flags && o.need_lcp && o.final_sa && !o.keys_only && !(o.k32 && slot_cap) ? o.defer_flags : nullptr;
    if (bflag) {
        be.memset(bflag, 0, (size_t)segs.G * sizeof(uint64_t));
        r.defer_flags = bflag;
        r.defer_desc = desc;
    }
    r.total = n_elems;
    r.cap = n_elems;
    // Deferring pays when the tiles the comparison sort gets are the exception (repeat families, tandem arrays in a genome).  A text
    // in which they are the rule -- byte alphabets, whose 8-char keys tie all the time -- would flood the refinement (its work
    // memory is the sort's own buffers): there every tie is compared, as before.  Decided before the comparison sort runs.
    auto defer_check = [&](uint32_t tiles_to_compare) {
        if (bflag && (uint64_t)tiles_to_compare * TILE_E > n_elems / 2) { bflag = nullptr; r.defer_flags = nullptr; }
    };
    BackendEvent t0 = be.record();
    if (o.seg_ends && segs.seg_start == s.seg_start)
        throw std::invalid_argument("segments in fixed-capacity regions must be bucketed (results are written compactly)");
    const uint64_t* in_key = slot_cap ? slot_key : (o.in_key && segs.seg_start == s.seg_start) ? o.in_key : cur.key;
    const idx_t* in_sa = slot_cap ? slot_sa : (o.in_key && segs.seg_start == s.seg_start) ? static_cast<const idx_t*>(o.in_sa) : cur.sa;
    const uint8_t* no_shift = nullptr;
    r.k32 = o.k32 && slot_cap != 0;
    if (from_text) {
        CAPS_LAUNCH((tile_sort_kernel<idx_t, BITS, true>), n_tiles, TILE_NT, be, sd, P, n, o.text_base, lcp_mode, 0u,
                    (const uint64_t*)nullptr, (const idx_t*)nullptr, cur.key, cur.sa, cur.lcp, fin, seg_map, redo, no_shift);
        if (eq_tiles) CAPS_LAUNCH((tile_sort_eq_kernel<idx_t, BITS, true>), ggrid, TILE_NT, be, sd, P, n, o.text_base, lcp_mode, 0u,
                    (const uint64_t*)nullptr, (const idx_t*)nullptr, cur.key, cur.sa, cur.lcp, fin, seg_map, (const uint32_t*)redo, redo3, 0u | drop, redo2);
        if (eq_tiles) CAPS_LAUNCH((tile_sort_eq_kernel<idx_t, BITS, true, true>), ggrid, TILE_NT, be, sd, P, n, o.text_base, lcp_mode, 0u,
                    (const uint64_t*)nullptr, (const idx_t*)nullptr, cur.key, cur.sa, cur.lcp, fin, seg_map, (const uint32_t*)redo2, redo3, 0u, (uint32_t*)nullptr);
        CAPS_LAUNCH_RUNS(tile_sort_general_kernel, (idx_t, BITS, true), ggrid, TILE_NT, be, sd, P, n, o.text_base, lcp_mode, 0u,
                    (const uint64_t*)nullptr, (const idx_t*)nullptr, cur.key, cur.sa, cur.lcp, fin, (const uint32_t*)redo3, 0u, bflag);
    } else if (r.k32) {
        // 32-bit keys in the slots; the tiles the first kernel cannot finish are re-sorted from 64-bit keys cut from the text
        CAPS_LAUNCH((tile_sort_kernel<idx_t, BITS, false, uint32_t>), n_tiles, TILE_NT, be, sd, P, n, (uint64_t)0, lcp_mode, slot_cap,
                    reinterpret_cast<const uint32_t*>(in_key), in_sa, cur.key, cur.sa, cur.lcp, fin, seg_map, redo, (const uint8_t*)o.bk->kshift);
        if (eq_tiles) CAPS_LAUNCH((tile_sort_eq_kernel<idx_t, BITS, false>), ggrid, TILE_NT, be, sd, P, n, (uint64_t)0, lcp_mode, slot_cap,
                    (const uint64_t*)nullptr, in_sa, cur.key, cur.sa, cur.lcp, fin, (const BucketParams*)nullptr, (const uint32_t*)redo, redo3, 1u | drop, redo2);
        if (eq_tiles) CAPS_LAUNCH((tile_sort_eq_kernel<idx_t, BITS, false, true>), ggrid, TILE_NT, be, sd, P, n, (uint64_t)0, lcp_mode, slot_cap,
                    (const uint64_t*)nullptr, in_sa, cur.key, cur.sa, cur.lcp, fin, (const BucketParams*)nullptr, (const uint32_t*)redo2, redo3, 1u, (uint32_t*)nullptr);
        CAPS_LAUNCH_RUNS(tile_sort_general_kernel, (idx_t, BITS, false), ggrid, TILE_NT, be, sd, P, n, (uint64_t)0, lcp_mode, slot_cap,
                    (const uint64_t*)nullptr, in_sa, cur.key, cur.sa, cur.lcp, fin, (const uint32_t*)redo3, 1u, bflag);
    } else {
        // every tile queued (skewed keys): one workgroup per tile -- the build of tile_sort_eq_kernel without the loop over
        // the queue (kernels.h PERSIST); otherwise the queue is short (often empty) and a fixed grid walks it
        const bool all_queued = o.skewed_keys && eq_tiles && !o.comparison_only;
        const bool per_tile = all_queued && (uint64_t)n_tiles * TILE_NT <= 0xFFFFFFFFull && !std::getenv("CAPS_SA_EQ_PERSISTENT");
        if (o.comparison_only) {
            CAPS_LAUNCH(queue_all_tiles_kernel, (n_tiles + 255) / 256, 256, be, sd, redo3);
            CAPS_LAUNCH_RUNS(tile_sort_general_kernel, (idx_t, BITS, false), ggrid, TILE_NT, be, sd, P, n, (uint64_t)0, lcp_mode, slot_cap,
                        in_key, in_sa, cur.key, cur.sa, cur.lcp, fin, (const uint32_t*)redo3, 0u, bflag);
        } else {
        if (all_queued && !per_tile) CAPS_LAUNCH(queue_all_tiles_kernel, (n_tiles + 255) / 256, 256, be, sd, redo);
        else if (!all_queued) CAPS_LAUNCH((tile_sort_kernel<idx_t, BITS, false>), n_tiles, TILE_NT, be, sd_hot, P, n, (uint64_t)0, lcp_mode, slot_cap,
                    in_key, in_sa, cur.key, cur.sa, cur.lcp, fin, seg_map, redo, no_shift);
        if (per_tile) {
            // (no queue: workgroup b takes tile b)
            CAPS_LAUNCH((tile_sort_eq_kernel<idx_t, BITS, false, false, false>), n_tiles, TILE_NT, be, sd_hot, P, n, (uint64_t)0, lcp_mode, slot_cap,
                        in_key, in_sa, cur.key, cur.sa, cur.lcp, fin, seg_map, (const uint32_t*)nullptr, redo3, 0u | drop, redo2);
            // what is left goes on with exact grids, one workgroup per entry: the lengths of the two queues come back to the
            // host (a round trip of ~20 us; a grid over all tiles for a queue that is mostly empty costs 0.8 ms at 3e9, and the
            // builds that walk a queue with a fixed grid hold 6 - 20 x more registers in scratch)
            uint32_t qn[2] = {0, 0};
            be.d2h(&qn[0], redo2, sizeof(uint32_t));
            be.d2h(&qn[1], redo3, sizeof(uint32_t));
            be.sync();
            if (qn[0]) CAPS_LAUNCH((tile_sort_eq_kernel<idx_t, BITS, false, true, false>), qn[0], TILE_NT, be, sd, P, n, (uint64_t)0, lcp_mode, slot_cap,
                        in_key, in_sa, cur.key, cur.sa, cur.lcp, fin, seg_map, (const uint32_t*)redo2, redo3, 0u, (uint32_t*)nullptr);
            const uint32_t ng = qn[0] + qn[1];           // (the third stage may pass entries on: a bound)
            if (bflag && qn[0]) {                        // ... the exact count for the decision to defer
                uint32_t q3 = 0;
                be.d2h(&q3, redo3, sizeof q3);
                be.sync();
                defer_check(q3);
            } else defer_check(qn[1]);
            if (ng && be.long_runs)
                CAPS_LAUNCH((tile_sort_general_kernel<idx_t, BITS, false, true, false>), ng, TILE_NT, be, sd, P, n, (uint64_t)0, lcp_mode, slot_cap,
                            in_key, in_sa, cur.key, cur.sa, cur.lcp, fin, (const uint32_t*)redo3, 0u, bflag);
            else if (ng)
                CAPS_LAUNCH((tile_sort_general_kernel<idx_t, BITS, false, false, false>), ng, TILE_NT, be, sd, P, n, (uint64_t)0, lcp_mode, slot_cap,
                            in_key, in_sa, cur.key, cur.sa, cur.lcp, fin, (const uint32_t*)redo3, 0u, bflag);
        } else {
            if (eq_tiles) CAPS_LAUNCH((tile_sort_eq_kernel<idx_t, BITS, false>), ggrid, TILE_NT, be, sd_hot, P, n, (uint64_t)0, lcp_mode, slot_cap,
                        in_key, in_sa, cur.key, cur.sa, cur.lcp, fin, seg_map, (const uint32_t*)redo, redo3, 0u | drop, redo2);
            if (eq_tiles) CAPS_LAUNCH((tile_sort_eq_kernel<idx_t, BITS, false, true>), ggrid, TILE_NT, be, sd, P, n, (uint64_t)0, lcp_mode, slot_cap,
                        in_key, in_sa, cur.key, cur.sa, cur.lcp, fin, seg_map, (const uint32_t*)redo2, redo3, 0u, (uint32_t*)nullptr);
            if (bflag) {                                  // (one round trip, only where ties may be deferred)
                uint32_t ng = 0;
                be.d2h(&ng, redo3, sizeof ng);
                be.sync();
                defer_check(ng);
            }
            CAPS_LAUNCH_RUNS(tile_sort_general_kernel, (idx_t, BITS, false), ggrid, TILE_NT, be, sd, P, n, (uint64_t)0, lcp_mode, slot_cap,
                        in_key, in_sa, cur.key, cur.sa, cur.lcp, fin, (const uint32_t*)redo3, 0u, bflag);
        }
        }
    }
    BackendEvent t1 = be.record();
    mark("tile sort");
    if (dbg) {
        uint32_t q[3] = {0, 0, 0};
        be.d2h(&q[0], redo, 4); be.d2h(&q[1], redo2, 4); be.d2h(&q[2], redo3, 4);
        be.sync();
        std::fprintf(stderr, "[sort] queues after the tile sorts: %u %u %u of %u tiles, deferring %d\n", q[0], q[1], q[2], n_tiles, bflag ? 1 : 0);
    }
    if (o.tile_clock) { o.tile_clock->spans.push_back({t0, t1}); o.tile_clock->elems.push_back(n_elems); }
    const uint32_t grid = n_tiles < be.persistent_blocks() ? n_tiles : be.persistent_blocks();
    ElemBuf<idx_t> a = cur, b = oth;
    uint32_t* n_active = reinterpret_cast<uint32_t*>(desc + n_tiles);     // (the descriptor array has an entry to spare)
    for (uint64_t R = TILE_E; R < max_len; R *= 2) {
        be.memset(n_active, 0, sizeof(uint32_t));
        CAPS_LAUNCH_RUNS(merge_partition_kernel, (idx_t, BITS), (n_tiles + 255) / 256, 256, be, sd, P, n, R, ~0ull, skip ? 1u : 0u,
                    lcp_mode, (const uint64_t*)a.key, (const idx_t*)a.sa, desc,
                    o.pass_counters ? o.pass_counters + r.passes : (uint64_t*)nullptr, bflag ? 1u : 0u, n_active);
        BackendEvent m0 = be.record();
        CAPS_LAUNCH_RUNS(merge_pass_kernel, (idx_t, BITS), grid, TILE_NT, be, (const TileDesc*)desc, (const uint32_t*)n_active, P, n,
                    (const uint64_t*)a.key, (const idx_t*)a.sa, b.key, b.sa, b.lcp, bflag);
        BackendEvent m1 = be.record();
        if (o.merge_clock) { o.merge_clock->spans.push_back({m0, m1}); o.merge_clock->elems.push_back(n_elems); }
        std::swap(a, b);
        ++r.passes;
        mark("merge pass");
    }
    if (o.unify && r.skip_finished && r.passes) {
        CAPS_LAUNCH((unify_kernel<idx_t>), n_tiles, 256, be, sd, (const uint64_t*)r.buf[1].key, (const idx_t*)r.buf[1].sa,
                    r.buf[0].key, r.buf[0].sa);
        r.unified = true;
    } else if (o.unify && r.skip_finished) {
        r.unified = true;                                 // no pass ran: everything is in buf[0]
    }
    return r;
}

// The letter-run buckets of a sort (SortResult::run_buckets; text.h "letter runs"): every element gets its run key -- (class of
// the run's terminator, what is left of the run, the text behind it) -- in place of the key all of them share; an ordinary
// sort by those keys (tile sort + merge passes whose comparisons are register compares: the text is read only where two
// runs of one length are followed by the same chars) orders the bucket, run_emit_kernel writes SA and derives the LCPs from
// the run keys.  Called from finalize(), after the rest of the result is in dSA / dLCP (the buckets' neighbours are read
// from there) and the bucket tables are idle.
template <typename idx_t, int BITS>
void sort_run_buckets(Backend& be, const uint32_t* P, uint64_t n, const SortResult<idx_t>& r, idx_t* dSA, idx_t* dLCP)
{
    for (const auto& rb : r.run_buckets) {
        const uint64_t len = rb.s1 - rb.s0;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((len + 255) / 256, 4096);
        SegBufs t = r.run_tables;
        t.G = 1;
        t.seg_end = nullptr;
        CAPS_LAUNCH(segment_range_kernel, 1, 64, be, t.seg_start, rb.s0, rb.s1);
        prepare_segments(be, t, tiles_of(len));
        CAPS_LAUNCH((run_rekey_kernel<idx_t, BITS>), grid, 256, be, P, n, rb.s0, rb.s1, rb.key, r.buf[0].key, (const idx_t*)r.buf[0].sa);
        SortOpts o;                                   // order only: the LCPs come from the run keys
        const SortResult<idx_t> rr = segmented_sort<idx_t, BITS>(be, P, n, r.run_desc, t, tiles_of(len), len, r.buf[0], r.buf[1], len, o);
        const ElemBuf<idx_t> res = rr.uniform();
        CAPS_LAUNCH((run_emit_kernel<idx_t, BITS>), grid, 256, be, P, n, rb.s0, rb.s1, r.total, (const uint64_t*)res.key, (const idx_t*)res.sa,
                    dSA, dLCP);
    }
}

// ---- deferred ties, resolved (kernels.h "Deferred ties, resolved") --------------------------------------------------
// Work memory: the two element buffers of the sort, idle once its result is in SA / LCP -- two chunks of 16 (24) bytes per element.
constexpr uint32_t MSD_MAX_LEVELS = 128;          // (GRCh38-shaped repeats: 7; a million-member group at 2 % divergence: ~15)
struct MsdArena {
    char* lo[2] = {nullptr, nullptr};
    char* hi[2] = {nullptr, nullptr};
    bool failed = false;
    static char* up(char* p) { return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255)); }
    template <typename T> T* head(size_t count, int pref = 0)
    {
        for (int k = 0; k < 2; ++k) {
            const int c = (pref + k) & 1;
            char* p = up(lo[c]);
            if (p && p + count * sizeof(T) <= hi[c]) { lo[c] = p + count * sizeof(T); return reinterpret_cast<T*>(p); }
        }
        failed = true;
        return nullptr;
    }
    template <typename T> T* tail(size_t count)
    {
        for (int c = 1; c >= 0; --c) {
            if (!hi[c]) continue;
            char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(hi[c]) - count * sizeof(T)) & ~uintptr_t(255));
            if (p >= lo[c] && hi[c] - lo[c] >= (ptrdiff_t)(count * sizeof(T))) { hi[c] = p; return reinterpret_cast<T*>(p); }
        }
        failed = true;
        return nullptr;
    }
};

// Orders the groups of equal keys a sort with deferred ties has left open (r.defer_flags) in place in SA / LCP (the slice the sort
// wrote).  False: the groups do not fit the work memory -- nothing of the result may be used, the caller sorts again without
// deferring (a text in which most suffixes sit in large groups of equal keys; never seen outside constructed inputs).
template <typename idx_t, int BITS>
bool msd_refine(Backend& be, const uint32_t* P, uint64_t n, const SortResult<idx_t>& r, idx_t* SA, idx_t* LCP)
{
    constexpr uint32_t KCH = TextTraits<BITS>::KCH;
    const SegDesc sd = r.segs.desc();
    const uint32_t n_tiles = r.n_tiles;
    const bool dbg = std::getenv("CAPS_SA_DEBUG") != nullptr;
    if (!r.buf[0].region_bytes || !r.buf[1].region_bytes || !r.defer_desc) return false;
    if (std::getenv("CAPS_SA_TEST_MSD_FAIL")) return false;             // tests: "the groups do not fit the work memory"
    // ---- level 0: which tiles hold sentinels (tables in the idle tile descriptors: 48 bytes per tile)
    char* dm = reinterpret_cast<char*>(r.defer_desc);
    uint64_t* tcnt = reinterpret_cast<uint64_t*>(dm);
    uint64_t* toff = tcnt + n_tiles;                                      // [n_tiles + 1]
    uint32_t* ftile = reinterpret_cast<uint32_t*>(toff + n_tiles + 1);    // [n_tiles]
    uint64_t* ttmp = reinterpret_cast<uint64_t*>(ftile + n_tiles + (n_tiles & 1));
    static_assert(sizeof(TileDesc) >= 8 + 8 + 4 + 8, "level-0 tables fit the tile descriptors");
    if ((size_t)n_tiles * 20 + 16 + 16 * ((size_t)n_tiles / SCAN_CHUNK + 3) > (size_t)(n_tiles + 1) * sizeof(TileDesc)) return false;   // (a handful of tiles)
    CAPS_LAUNCH(msd_tile_counts_kernel, (n_tiles + 255) / 256, 256, be, sd, (const uint64_t*)r.defer_flags, tcnt);
    device_exclusive_scan<uint64_t>(be, tcnt, n_tiles, toff, ttmp);
    CAPS_LAUNCH(msd_tile_list_kernel, (n_tiles + 255) / 256, 256, be, (const uint64_t*)tcnt, (const uint64_t*)toff, n_tiles, ftile);
    uint64_t tot0 = 0;
    be.d2h(&tot0, toff + n_tiles, sizeof tot0);
    be.sync();
    const uint64_t M0 = tot0 & MSD_ELEM_MASK, NF = tot0 >> 40;
    if (NF == 0) return true;
    if (M0 >= (1ull << 32) - 2 || NF > 0x7FFFFFFFull) return false;       // (member counts travel in 32 bits)
    MsdArena ar;
    const size_t per = sizeof(uint64_t) + 2 * sizeof(idx_t);
    for (int c = 0; c < 2; ++c) {
        ar.lo[c] = reinterpret_cast<char*>(r.buf[c].key);
        ar.hi[c] = ar.lo[c] + std::min<size_t>(r.buf[c].region_bytes, (size_t)r.cap * per);
    }
    uint64_t* flags0 = ar.head<uint64_t>(M0, 0);
    uint64_t* offs0 = ar.head<uint64_t>(M0 + 1, 1);
    uint64_t* stmp = ar.head<uint64_t>(2 * (M0 / SCAN_CHUNK + 3), 1);
    if (ar.failed) return false;
    CAPS_LAUNCH((msd_flags0_kernel<idx_t>), (uint32_t)NF, 256, be, sd, (const uint32_t*)ftile, (const uint64_t*)toff, (const idx_t*)LCP, flags0);
    device_exclusive_scan<uint64_t>(be, flags0, (uint32_t)M0, offs0, stmp);
    uint64_t tot1 = 0;
    be.d2h(&tot1, offs0 + M0, sizeof tot1);
    be.sync();                                         // members and groups: the rest of the work memory is sized by them, not by M0
    const uint64_t m0 = tot1 & 0xFFFFFFFFull, G0 = tot1 >> 32;
    if (m0 == 0) return true;
    idx_t* wsa0 = ar.tail<idx_t>(m0 + 2);
    uint64_t* segX = ar.tail<uint64_t>(m0 / 2 + 2);              // (later generations of the group table live here too: a group may split)
    uint64_t* gposX = ar.tail<uint64_t>(m0 / 2 + 2);
    uint64_t* gdepX = ar.tail<uint64_t>(m0 / 2 + 2);
    uint64_t* gplX = ar.tail<uint64_t>(m0 / 2 + 2);
    uint8_t* skip = ar.tail<uint8_t>(m0 / 2 + 2);
    uint8_t* skip_tiles = ar.tail<uint8_t>(m0 / 2 + 2);
    uint32_t* wgid = ar.tail<uint32_t>(m0 + 2);
    uint32_t* flist = ar.tail<uint32_t>(m0 + 2);
    uint32_t* qlist[3] = {ar.tail<uint32_t>(m0 / (MSD_FIN_MAX + 1) + 2), ar.tail<uint32_t>(m0 / (MSD_QK_CAP[0] + 1) + 2),
                          ar.tail<uint32_t>(m0 / (MSD_QK_CAP[1] + 1) + 2)};
    uint64_t* edges = ar.tail<uint64_t>(2 * G0 + 4);
    uint64_t* out3 = ar.tail<uint64_t>(12);
    if (ar.failed) return false;
    CAPS_LAUNCH((msd_compact0_kernel<idx_t>), (uint32_t)NF, 256, be, sd, (const uint32_t*)ftile, (const uint64_t*)toff, (const idx_t*)SA,
                (const uint64_t*)flags0, (const uint64_t*)offs0, wsa0, segX, gposX, wgid, gdepX, (uint64_t)KCH, gplX);
    CAPS_LAUNCH(msd_close_kernel, 1, 64, be, (const uint64_t*)(offs0 + M0), segX, out3);
    uint64_t D = KCH;                                  // (the depth of the groups that never jumped: for the log)
    auto finish = [&](const uint64_t* seg, const uint64_t* gpos, const uint64_t* gdep, const idx_t* wsa, const uint32_t* gid, uint64_t m_bound) {
        const uint32_t ggrid = (uint32_t)std::min<uint64_t>((m_bound / 2 + 255) / 256 + 1, 4ull * be.persistent_blocks());
        CAPS_LAUNCH(msd_groups_kernel, ggrid, 256, be, seg, skip, skip_tiles, flist, qlist[0], qlist[1], qlist[2], out3);
        const uint32_t grid = (uint32_t)std::min<uint64_t>((m_bound + MSD_FIN_MEMBERS - 1) / MSD_FIN_MEMBERS, 32ull * be.persistent_blocks());
        if (be.long_runs) CAPS_LAUNCH((msd_finish_kernel<idx_t, BITS, true>), grid ? grid : 1, 256, be, P, n, gdep, seg, gpos, wsa, gid, (const uint32_t*)flist, SA, LCP, (const uint64_t*)out3);
        else CAPS_LAUNCH((msd_finish_kernel<idx_t, BITS, false>), grid ? grid : 1, 256, be, P, n, gdep, seg, gpos, wsa, gid, (const uint32_t*)flist, SA, LCP, (const uint64_t*)out3);
    };
    CAPS_LAUNCH(msd_edges_kernel, (uint32_t)std::min<uint64_t>((G0 + 255) / 256 + 1, 4ull * be.persistent_blocks()), 256, be, (const uint64_t*)segX,
                (const uint64_t*)gposX, (const uint64_t*)out3, edges);
    finish(segX, gposX, gdepX, wsa0, wgid, m0);
    // (the groups of up to a tile, each finished by one workgroup in LDS; launched once their number is known, reads this generation's tables)
    auto quick = [&](const uint64_t* seg, const uint64_t* gpos, const uint64_t* gdep, const idx_t* wsa, const uint64_t* nq) {
        auto go = [&](auto cls) {
            constexpr uint32_t C = decltype(cls)::value;
            if (!nq[C]) return;
            if (be.long_runs) CAPS_LAUNCH((msd_quick_kernel<idx_t, BITS, true, MSD_QK_CAP[C], MSD_QK_NT[C]>), (uint32_t)nq[C], MSD_QK_NT[C], be, P, n, (const uint32_t*)qlist[C], seg, gpos, gdep, wsa, SA, LCP);
            else CAPS_LAUNCH((msd_quick_kernel<idx_t, BITS, false, MSD_QK_CAP[C], MSD_QK_NT[C]>), (uint32_t)nq[C], MSD_QK_NT[C], be, P, n, (const uint32_t*)qlist[C], seg, gpos, gdep, wsa, SA, LCP);
        };
        go(std::integral_constant<uint32_t, 2>());      // (the long ones first)
        go(std::integral_constant<uint32_t, 1>());
        go(std::integral_constant<uint32_t, 0>());
    };
    uint64_t h3[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    be.d2h(h3, out3, sizeof h3);
    be.sync();
    uint64_t G = h3[0], m = h3[1], nopen = h3[2], gmax = h3[3];
    quick(segX, gposX, gdepX, wsa0, h3 + 6);
    r.msd_groups = G;
    r.msd_elems = m;
    // (when every group is in its final order: the LCPs at its two ends, see msd_edges_kernel)
    const uint64_t n_edges = 2 * G;
    auto fix_edges = [&]() {
        if (n_edges) CAPS_LAUNCH((msd_fix_edges_kernel<idx_t, BITS>), (uint32_t)((n_edges + 255) / 256), 256, be, P, n, (const uint64_t*)edges, n_edges,
                                 r.total, (const idx_t*)SA, LCP);
    };
    if (dbg) std::fprintf(stderr, "[msd] level 0: %llu flagged tiles, %llu elements in them, %llu groups of %llu members, %llu above %u finished in LDS, %llu above %u (largest %llu)\n",
                          (unsigned long long)NF, (unsigned long long)M0, (unsigned long long)G, (unsigned long long)m, (unsigned long long)(h3[6] + h3[7] + h3[8]), MSD_FIN_MAX,
                          (unsigned long long)nopen, MSD_QK_MAX, (unsigned long long)gmax);
    if (nopen == 0) { fix_edges(); return true; }
    // ---- the levels: work arrays for m members in at most m / 2 groups; level 0's outputs stay where they are (chunk tails)
    ar.lo[0] = reinterpret_cast<char*>(r.buf[0].key);                   // (flags0 / offs0 / stmp are dead)
    ar.lo[1] = reinterpret_cast<char*>(r.buf[1].key);
    const uint64_t cap2 = m + 2, Gm = m / 2 + 2, tiles_max = m / TILE_E + m / (MSD_QK_MAX + 1) + 4;
    ElemBuf<idx_t> E0, E1;
    for (ElemBuf<idx_t>* e : {&E0, &E1}) {
        char* base = reinterpret_cast<char*>(ar.head<uint64_t>(cap2 * per / sizeof(uint64_t) + 1, e == &E0 ? 0 : 1));
        if (!base) return false;
        e->key = reinterpret_cast<uint64_t*>(base);
        e->sa = reinterpret_cast<idx_t*>(base + cap2 * sizeof(uint64_t));
        e->lcp = e->sa + cap2;                                          // (sa | lcp contiguous: together they hold the scan of the flags)
    }
    uint64_t* segY = ar.head<uint64_t>(Gm);
    uint64_t* gposY = ar.head<uint64_t>(Gm);
    uint64_t* gdepY = ar.head<uint64_t>(Gm);
    uint64_t* gmin = ar.head<uint64_t>(Gm);
    uint64_t* gplY = ar.head<uint64_t>(Gm);
    uint32_t* wgidY = ar.head<uint32_t>(cap2);
    SegBufs ls;
    ls.tile_off = ar.head<uint32_t>(Gm);
    ls.tile_rec = ar.head<TileInfo>(tiles_max);
    ls.out2 = ar.head<uint64_t>(2);
    TileDesc* ldesc = ar.head<TileDesc>(tiles_max + 1);
    uint64_t* big_cnt = ar.head<uint64_t>(Gm);
    uint64_t* big_tmp = ar.head<uint64_t>(2 * (Gm / SCAN_CHUNK + 3));
    uint64_t* ltmp = ar.head<uint64_t>(2 * (cap2 / SCAN_CHUNK + 3));
    if (ar.failed) return false;
    uint64_t *seg = segX, *gpos = gposX, *seg_n = segY, *gpos_n = gposY, *gdep = gdepX, *gdep_n = gdepY, *gpl = gplX, *gpl_n = gplY;
    uint32_t *gid = wgid, *gid_n = wgidY;
    const idx_t* in_sa = wsa0;
    uint64_t* kin = E1.key;                                             // the level's keys before the sort; the flags after it
    uint64_t* flags = E1.key;
    uint64_t* offs = reinterpret_cast<uint64_t*>(E1.sa);
    // A level takes 32 chars off a group.  Groups whose members agree for tens of thousands of chars WITHOUT ever coming out of a
    // level unchanged (an exact tandem array of thousands of copies, a single-letter block outside quantile mode: every level
    // peels the members that leave the array inside the window) would take a level per window: past MSD_MAX_LEVELS the groups
    // are the comparators' business after all (the caller builds again without deferring; they know the run table).
    uint32_t max_levels = MSD_MAX_LEVELS;
    if (const char* ml = std::getenv("CAPS_SA_TEST_MSD_MAX_LEVELS")) max_levels = (uint32_t)std::atoi(ml);
    while (nopen > 0) {
        if (r.msd_levels >= max_levels) return false;
        const uint32_t mg = (uint32_t)((m + 255) / 256);
        if (r.msd_levels) {                             // a group the last level left as it was jumps to what all its members share
            const uint32_t gg = (uint32_t)std::min<uint64_t>((G + 255) / 256 + 1, 4ull * be.persistent_blocks());
            be.memset(gmin, 0, (size_t)G * sizeof(uint64_t));
            if (be.long_runs) CAPS_LAUNCH((msd_jump_kernel<idx_t, BITS, true>), mg, 256, be, P, n, m, (const uint64_t*)seg, (const uint32_t*)gid, (const uint8_t*)skip, in_sa, (const uint64_t*)gdep, (const uint64_t*)gpl, gmin);
            else CAPS_LAUNCH((msd_jump_kernel<idx_t, BITS, false>), mg, 256, be, P, n, m, (const uint64_t*)seg, (const uint32_t*)gid, (const uint8_t*)skip, in_sa, (const uint64_t*)gdep, (const uint64_t*)gpl, gmin);
            CAPS_LAUNCH(msd_jump_apply_kernel, gg, 256, be, (const uint64_t*)out3, (const uint8_t*)skip, (const uint64_t*)gmin, KCH, gdep);
        }
        CAPS_LAUNCH((msd_rekey_kernel<idx_t, BITS>), mg, 256, be, P, n, (const uint64_t*)gdep, m, (const uint32_t*)gid, (const uint8_t*)skip, in_sa, kin);
        {                                               // the segmented sort, tile by tile + merge passes
            ls.seg_start = seg;
            ls.G = (uint32_t)G;
            // (no read-back of the tile count: such groups have at most this many tiles, the kernels return on the rest)
            const uint32_t lt = (uint32_t)std::min<uint64_t>(tiles_max, m / TILE_E + nopen + 1);
            prepare_segments(be, ls, lt, big_tmp, big_cnt, false, skip_tiles);
            SortOpts o;
            o.keys_only = true;                        // (key, position descending): no comparison reads the text
            o.comparison_only = true;                  // few distinct keys per group: the bin sorts would pass every tile on
            o.skip_finished = true;
            o.unify = true;                            // everything ends in E0
            o.in_key = kin;
            o.in_sa = in_sa;
            const SortResult<idx_t> rr = segmented_sort<idx_t, BITS>(be, P, n, ldesc, ls, lt, gmax, E0, E1, m, o);
            if (rr.uniform().key != E0.key) return false;
        }
        CAPS_LAUNCH((msd_classify_kernel<idx_t, BITS>), mg, 256, be, n, (const uint64_t*)gdep, m, (const uint64_t*)seg, (const uint32_t*)gid, (const uint8_t*)skip,
                    (const uint64_t*)E0.key, (const idx_t*)E0.sa, (const uint64_t*)gpos, SA, LCP, flags);
        device_exclusive_scan<uint64_t>(be, flags, (uint32_t)m, offs, ltmp);
        CAPS_LAUNCH((msd_compact_kernel<idx_t>), mg, 256, be, m, (const uint64_t*)seg, (const uint32_t*)gid, (const idx_t*)E0.sa, (const uint64_t*)flags,
                    (const uint64_t*)offs, (const uint64_t*)gpos, E0.lcp, seg_n, gpos_n, gid_n, (const uint64_t*)gdep, gdep_n, KCH, gpl_n);
        CAPS_LAUNCH(msd_close_kernel, 1, 64, be, (const uint64_t*)(offs + m), seg_n, out3);
        D += KCH;
        ++r.msd_levels;
        std::swap(seg, seg_n);
        std::swap(gpos, gpos_n);
        std::swap(gid, gid_n);
        std::swap(gdep, gdep_n);
        std::swap(gpl, gpl_n);
        in_sa = E0.lcp;
        finish(seg, gpos, gdep, in_sa, gid, m);
        be.d2h(h3, out3, sizeof h3);
        be.sync();
        G = h3[0];
        m = h3[1];
        nopen = h3[2];
        gmax = h3[3];
        quick(seg, gpos, gdep, in_sa, h3 + 6);
        if (dbg) std::fprintf(stderr, "[msd] depth %llu: %llu groups of %llu members, %llu above %u finished in LDS, %llu above %u (largest %llu)\n", (unsigned long long)D,
                              (unsigned long long)G, (unsigned long long)m, (unsigned long long)(h3[6] + h3[7] + h3[8]), MSD_FIN_MAX, (unsigned long long)nopen,
                              MSD_QK_MAX, (unsigned long long)gmax);
        if (G > Gm - 1 || m > cap2 - 2) return false;                   // (cannot happen: both only shrink)
    }
    fix_edges();
    return true;
}

// Gather SA/LCP of a sorted segment set into the caller's arrays + segment-head LCPs (a11).
// With boundary records (r.fin) only the segments that needed merge passes are still to be
// copied; the heads are then fixed from the records.
template <typename idx_t, int BITS>
void finalize(Backend& be, const uint32_t* P, uint64_t n, const SortResult<idx_t>& r, idx_t* dSA, idx_t* dLCP)
{
    if (r.n_tiles == 0) return;
    auto timed = [&](KernelClock* c, uint64_t elems, auto&& body) {
        if (!c) { body(); return; }
        const BackendEvent a = be.record();
        body();
        c->spans.push_back({a, be.record()});
        c->elems.push_back(elems);
    };
    timed(r.finish_clock, r.total, [&] {
        if (r.fin.sa == nullptr || r.passes > 0)
            CAPS_LAUNCH((finalize_kernel<idx_t, BITS>), r.n_tiles, 256, be, r.segs.desc(), P, n, r.pingpong(), r.skip_finished ? 1u : 0u,
                        r.passes & 1u, dSA, dLCP, r.fin);
        if (r.fin.sa != nullptr)
            CAPS_LAUNCH((head_lcp_kernel<idx_t, BITS>), (r.segs.G + 255) / 256, 256, be, P, n, (const uint64_t*)r.segs.seg_start, r.segs.G,
                        r.fin, r.k32 ? 1u : 0u);
    });
    if (!r.run_buckets.empty()) timed(r.runb_clock, 0, [&] { sort_run_buckets<idx_t, BITS>(be, P, n, r, dSA, dLCP); });
    // deferred ties last: the neighbours of a group never change their LCP with it (all members share the key, and a member
    // whose suffix ends inside the key is placed -- first -- by the sort itself), so the head LCPs above stand
    if (r.defer_flags && !std::getenv("CAPS_SA_DEBUG_NO_MSD"))
        timed(r.msd_clock, 0, [&] { if (!msd_refine<idx_t, BITS>(be, P, n, r, dSA, dLCP)) r.msd_failed = true; });
}

// Shape of the direct path's two-level distribution: PG consecutive partitions per group, K1 groups.  Level A
// (text -> groups) and level B (group -> buckets of one tile) should fan out about equally (the runs a tile
// contributes to a destination are then equally long at both levels), a group must fit the LDS histogram of its
// bucket split with room for imbalance, and the group table must fit LDS.
inline int direct_shape(uint64_t n, uint32_t p, uint64_t m, uint32_t* PG, uint32_t* K1)
{
    *PG = *K1 = 0;
    if (p < 2 || n < 32ull * TILE_E || m < 64 || m > (1ull << 31)) return CAPS_SA_FB_SHAPE;
    const double group_max = 0.9 * (double)BUCKET_LDS * (double)BUCKET_TARGET;   // groups are sample quantiles: +-1.3 % at C3
    // (measured at C3, K1 = 500 / 1000 / 1400 / 2000: 50.0 / 49.4 / 49.7 / 50.3 ms per build -- flat; the square root it is)
    double want = std::sqrt((double)n / (double)BUCKET_TARGET);
    if (const char* k1 = std::getenv("CAPS_SA_DIRECT_K1")) { if (std::atof(k1) >= 2.0) want = std::atof(k1); }   // measurement
    if (want > (double)BUCKET_LDS) want = (double)BUCKET_LDS;
    if ((double)n / group_max > want) want = (double)n / group_max;
    if (want < 2.0) want = 2.0;
    if (want > (double)BUCKET_LDS) return CAPS_SA_FB_SHAPE;
    uint32_t pg = (uint32_t)((double)p / want);
    if (pg < 1) pg = 1;
    const uint32_t k1 = (p + pg - 1) / pg;
    if (k1 < 2 || k1 > BUCKET_LDS || (double)n / k1 > group_max || m / k1 < 16) return CAPS_SA_FB_SHAPE;
    *PG = pg;
    *K1 = k1;
    return CAPS_SA_FB_NONE;
}


// Tests only: CAPS_SA_TEST_STREAM_CAP=<per cent> shrinks the regions of level A's streams to that share of the MEAN stream size,
// so that a stream outgrows its region on any text and the build must take the CAPS_SA_FB_GROUP_OVERFLOW way out (a real text
// gets there through one key that holds several per cent of it; the regions are 1.33 x the expected sizes otherwise).
inline uint64_t test_stream_cap(uint64_t cap, uint64_t n_elems, uint64_t n_streams)
{
    const char* e = std::getenv("CAPS_SA_TEST_STREAM_CAP");
    if (!e || std::atof(e) <= 0.0 || n_streams == 0) return cap;
    const uint64_t c = (uint64_t)((double)n_elems / (double)n_streams * std::atof(e) / 100.0) + 2;
    return c < cap ? c & ~1ull : cap;
}

// Consumer of finished slices of the result (capi_impl.h build_host: copies them to the caller's arrays on a second stream while
// the next groups are still being sorted).  Called on the host when everything that produces SA / LCP[base, base + cnt) has been
// enqueued on the build's stream.
struct WaveSink {
    virtual void wave_done(Backend& be, uint64_t base, uint64_t cnt) = 0;
    // the build starts over (a slot overflowed under 32-bit keys, the deferred ties did not fit): what was handed over so far is
    // void -- the sink finishes reading the arrays the new attempt is about to rewrite and forgets it
    virtual void reset(Backend& be) = 0;
    virtual ~WaveSink() {}
};

template <typename idx_t> class Builder {
public:
    Builder(Backend& be, const Plan<idx_t>& pl) : be_(be), pl_(pl) {}
    // The direct path's level B + tile sort in `waves` runs over consecutive groups (= consecutive slices of the suffix array),
    // each reported to `sink` as soon as it is enqueued.  One wave (the default) is the device-resident build.
    // scratch: element arrays for the largest wave (scratch_elems entries each).  While later waves' input still sits in buffer A
    // (level A's streams, a view over ALL of A's bytes), nothing may use A's arrays as a work buffer: the waves sort in scratch and B.
    void set_waves(uint32_t waves, WaveSink* sink, ElemBuf<idx_t> scratch = ElemBuf<idx_t>(), uint64_t scratch_elems = 0)
    {
        waves_ = waves ? waves : 1;
        sink_ = sink;
        wave_scratch_ = scratch;
        wave_scratch_elems_ = scratch_elems;
    }
    bool sink_served() const { return sink_served_; }          // every slice of the result went through the sink

    // dT: n raw bytes on the device.  dSA/dLCP: n idx_t each on the device.  max_context: 0 or >= n = the suffix array and
    // LCP array of T; else the reference's bounded-context result (bounded.h).
    void build(const uint8_t* dT, idx_t* dSA, idx_t* dLCP, caps_sa_stats* st, uint64_t max_context = 0)
    {
        const uint64_t n = pl_.n;
        if (st) { *st = caps_sa_stats(); st->n = n; st->idx_bytes = sizeof(idx_t); st->p_eff = pl_.p; }
        if (n == 0) return;
        BackendEvent e0 = be_.record();
        bits_ = prepare_text(be_, dT, n, pl_.P, pl_.present, pl_.lut, pl_.text_bits);
        BackendEvent e1 = be_.record();
        if (max_context != 0 && max_context < n) {
            if (pl_.p < 2 || pl_.ppp < 1) throw std::invalid_argument("bounded max_context needs n >= 32 and at least two subproblems (the reference is undefined below that)");
            if (bits_ == 2) run_bounded<2>(dSA, dLCP, max_context);
            else run_bounded<8>(dSA, dLCP, max_context);
            BackendEvent e2 = be_.record();
            be_.sync();
            if (st) {
                st->bits_per_char = (uint32_t)bits_;
                st->ppp = pl_.ppp;
                st->workspace_bytes = pl_.bytes;
                st->ms_pack = be_.elapsed_ms(e0, e1);
                st->ms_total = be_.elapsed_ms(e0, e2);
                st->ms_sort_subarrays = be_.elapsed_ms(e1, e2_);
                st->ms_select_pivots = be_.elapsed_ms(e2_, e3_);
                st->ms_locate_pivots = be_.elapsed_ms(e3_, e4_);
                st->ms_partition = be_.elapsed_ms(e4_, e5_);
                st->ms_merge_partitions = be_.elapsed_ms(e5_, e6_);
                st->ms_boundary_lcp = be_.elapsed_ms(e6_, e2);
                st->path_fallback = CAPS_SA_FB_BOUNDED;
            }
            be_.release_events();
            return;
        }
        if (bits_ == 2) run<2>(dSA, dLCP, st, e0, e1);
        else run<8>(dSA, dLCP, st, e0, e1);
    }

    int bits() const { return bits_; }

private:
    Backend& be_;
    const Plan<idx_t>& pl_;
    int bits_ = 0;
    uint32_t waves_ = 1;
    WaveSink* sink_ = nullptr;
    ElemBuf<idx_t> wave_scratch_;
    uint64_t wave_scratch_elems_ = 0;
    bool sink_served_ = false;
    uint32_t waves_used_ = 1;
    KernelClock merge_clock_;
    KernelClock tile_clock_;
    KernelClock scatter_clock_;
    KernelClock count_clock_;
    KernelClock collate_clock_;
    KernelClock finish_clock_, runb_clock_, msd_clock_;      // finalize(): gather + head LCPs, letter-run buckets, deferred ties
    uint32_t pass_base_ = 0;
    uint32_t slot_stats_[2] = {0, 0};
    uint64_t spill_stats_[3] = {0, 0, 0};
    // per-build results of the phase sequences below
    uint32_t passes1_ = 0, passes2_ = 0, passesS_ = 0;
    BackendEvent e2_, e3_, e4_, e5_, e6_, e7_, la0_, la1_;
    uint64_t max_part_ = 0;
    uint32_t path_direct_ = 0, path_fallback_ = 0, direct_groups_ = 0, direct_quantile_ = 0, direct_k32_ = 0, run_buckets_ = 0;
    uint64_t direct_max_group_ = 0;
    uint64_t tie_groups_ = 0, tie_elems_ = 0;
    uint32_t tie_levels_ = 0;

    // timed: the full-size sorts (phase 1, phase 2) feed the kernel clocks of caps_sa_stats;
    // their passes count the elements they really move in pl_.pass_elems[pass_base_ ...]
    template <int BITS>
    SortResult<idx_t> seg_sort(const SegBufs& s, uint32_t n_tiles, uint64_t max_len, ElemBuf<idx_t> cur, ElemBuf<idx_t> oth,
                               uint64_t n_elems, SortOpts o, bool timed)
    {
        if (timed) {
            o.tile_clock = &tile_clock_;
            o.merge_clock = &merge_clock_;
            o.scatter_clock = &scatter_clock_;
            o.count_clock = &count_clock_;
            o.pass_counters = pl_.pass_elems + pass_base_;
            o.slot_stats = slot_stats_;
            // no speculation once a split of this build had to be redone (phase 2 of a text whose phase 1 overflowed
            // its slots would overflow them too: the attempt costs a scatter pass).  CAPS_SA_NO_SLOTS: never (debugging)
            o.speculate = std::getenv("CAPS_SA_NO_SLOTS") == nullptr && slot_stats_[1] == 0;
        }
        SortResult<idx_t> r = segmented_sort<idx_t, BITS>(be_, pl_.P, pl_.n, pl_.desc, s, n_tiles, max_len, cur, oth, n_elems, o);
        if (timed) {
            pass_base_ += r.passes;
            r.finish_clock = &finish_clock_;
            r.runb_clock = &runb_clock_;
            r.msd_clock = &msd_clock_;
        }
        return r;
    }
    void prepare_segments(const SegBufs& s, uint64_t tile_bound) { ::caps::prepare_segments(be_, s, tile_bound); }
    // completed segments go straight to the caller's arrays (boundary records live in pl_.bk)
    void set_final(SortOpts& o, idx_t* dSA, idx_t* dLCP)
    {
        o.final_sa = dSA;
        o.final_lcp = dLCP;
        o.bnd.first_key = pl_.bk.first_key;
        o.bnd.last_key = pl_.bk.last_key;
        o.bnd.first_sa = pl_.bk.first_sa;
        o.bnd.last_sa = pl_.bk.last_sa;
    }

    // ---- bounded context (bounded.h): the reference's own sequence, one thread per merge node ----
    static uint32_t tree_depth(uint64_t cnt) { uint32_t d = 0; while ((1ull << d) < cnt) ++d; return d; }
    template <int BITS>
    void run_bounded(idx_t* dSA, idx_t* dLCP, uint64_t ctx)
    {
        const uint64_t n = pl_.n, s = n / pl_.p, last = s + n % pl_.p, m = pl_.ppp, nsamp = pl_.m;
        const uint32_t p = pl_.p;
        idx_t *SAw = pl_.A.sa, *LCPw = pl_.A.lcp;
        idx_t *pivot = pl_.SA_.sa, *pw = pl_.SB_.sa, *t1 = pl_.SA_.lcp, *t2 = pl_.SB_.lcp;
        idx_t *Pm = pl_.Pm, *ruler = pl_.PmT;
        uint64_t* scan = pl_.seg2.seg_start;
        const uint32_t* P = pl_.P;
        auto grid = [](uint64_t threads) { return capped_grid((threads + 255) / 256, 256); };
        // permute + sort_subarrays (:148-184)
        CAPS_LAUNCH((bounded_init_kernel<idx_t>), grid(n), 256, be_, n, (const idx_t*)nullptr, dSA, SAw, dLCP, LCPw);
        for (uint32_t d = tree_depth(last); d-- > 0;)
            CAPS_LAUNCH((bounded_sort_level_kernel<idx_t, BITS>), grid((uint64_t)p << d), 256, be_, P, n, ctx, n, p, s, d, dSA, dLCP, SAw, LCPw);
        e2_ = be_.record();
        // select_pivots (:197-222)
        CAPS_LAUNCH((bounded_sample_kernel<idx_t>), grid((uint64_t)p * m), 256, be_, (const idx_t*)dSA, n, p, s, m, pivot);
        CAPS_LAUNCH((bounded_init_kernel<idx_t>), grid(nsamp), 256, be_, nsamp, (const idx_t*)pivot, pw, pivot, t1, t2);
        for (uint32_t d = tree_depth(nsamp); d-- > 0;)
            CAPS_LAUNCH((bounded_sort_level_kernel<idx_t, BITS>), grid(1ull << d), 256, be_, P, n, ctx, nsamp, 1u, nsamp, d, pw, t1, pivot, t2);
        CAPS_LAUNCH((bounded_sample_kernel<idx_t>), grid(p - 1), 256, be_, (const idx_t*)pw, nsamp, 1u, nsamp, (uint64_t)(p - 1), pivot);
        e3_ = be_.record();
        // locate_pivots (:225-249)
        CAPS_LAUNCH((bounded_locate_kernel<idx_t, BITS>), grid((uint64_t)p * (p + 1)), 256, be_, P, n, ctx, p, s, (const idx_t*)dSA, (const idx_t*)pivot, Pm);
        e4_ = be_.record();
        // partition_sub_subarrays (:300-368) + the duplication of merge_sub_subarrays (:376-384)
        CAPS_LAUNCH((bounded_ruler_kernel<idx_t>), (p + 255) / 256, 256, be_, p, (const idx_t*)Pm, ruler, pl_.sizes);
        CAPS_LAUNCH(scan_sizes_kernel, 1, 1024, be_, (const uint64_t*)pl_.sizes, p, scan);
        CAPS_LAUNCH((bounded_collate_kernel<idx_t>), grid(n), 256, be_, n, p, s, (const uint64_t*)scan, (const idx_t*)ruler, (const idx_t*)Pm,
                    (const idx_t*)dSA, (const idx_t*)dLCP, SAw, LCPw);
        be_.d2d(dSA, SAw, n * sizeof(idx_t));
        be_.d2d(dLCP, LCPw, n * sizeof(idx_t));
        e5_ = be_.record();
        // merge_sub_subarrays (:386-409): sort_partition's merge tree of every partition
        for (uint32_t d = tree_depth(p); d-- > 0;)
            CAPS_LAUNCH((bounded_partition_level_kernel<idx_t, BITS>), grid((uint64_t)p << d), 256, be_, P, n, ctx, p, (const uint64_t*)scan,
                        (const idx_t*)ruler, d, dSA, dLCP, SAw, LCPw);
        e6_ = be_.record();
        // compute_partition_boundary_lcp (:431-447)
        CAPS_LAUNCH((bounded_boundary_kernel<idx_t, BITS>), (p + 255) / 256, 256, be_, P, n, p, (const uint64_t*)scan, (const idx_t*)dSA, dLCP);
    }

    // ---- the samplesort path: the reference's six phases (src/Suffix_Array.cpp:466-494) ----
    template <int BITS>
    void run_classic(idx_t* dSA, idx_t* dLCP)
    {
        const uint64_t n = pl_.n;
        const uint32_t p = pl_.p;
        const uint64_t s = n / p, last = s + n % p;
        // ---- phase 1 (a5): sort the p subarrays of contiguous text positions (no LCPs needed:
        //      the collate step moves only keys and indices)
        CAPS_LAUNCH(uniform_segments_kernel, (p + 256) / 256, 256, be_, pl_.seg1.seg_start, p, s, n);
        const uint32_t n_tiles1 = (p - 1) * tiles_of(s) + tiles_of(last);
        prepare_segments(pl_.seg1, n_tiles1);
        SortOpts o1;
        o1.from_text = true;
        o1.bk = &pl_.bk;                    // keys of a subarray span the whole key range
        o1.unify = true;                    // sample / locate / collate index subarrays as arrays
        SortResult<idx_t> r1 = seg_sort<BITS>(pl_.seg1, n_tiles1, last, pl_.A, pl_.B, n, o1, true);
        passes1_ = r1.passes;
        ElemBuf<idx_t> cur = r1.uniform();
        ElemBuf<idx_t> oth = cur.key == pl_.A.key ? pl_.B : pl_.A;
        e2_ = be_.record();

        // ---- pivots (a6)
        const uint64_t m = pl_.m;
        CAPS_LAUNCH((sample_kernel<idx_t>), (uint32_t)((m + 255) / 256), 256, be_, (const uint64_t*)pl_.seg1.seg_start, p,
                    pl_.ppp, (const uint64_t*)cur.key, (const idx_t*)cur.sa, pl_.SA_.key, pl_.SA_.sa);
        CAPS_LAUNCH(uniform_segments_kernel, 1, 256, be_, pl_.segS.seg_start, 1u, m, m);
        prepare_segments(pl_.segS, tiles_of(m));
        SortOpts os;
        os.bk = &pl_.bk;                    // phase 1's bucket tables are free again: the samples are bucket-sorted too
        os.unify = true;                    //   (one split + tile sort instead of ~11 merge passes over 5.6 M samples)
        SortResult<idx_t> rs = seg_sort<BITS>(pl_.segS, tiles_of(m), m, pl_.SA_, pl_.SB_, m, os, false);
        passesS_ = rs.passes;
        ElemBuf<idx_t> smp = rs.uniform();
        if (std::getenv("CAPS_SA_DEBUG_CHECK_SORTS")) {       // debugging: the sorted samples back on the host -- non-decreasing keys, positions in range
            std::vector<uint64_t> hk(m);
            std::vector<idx_t> hs(m);
            be_.d2h(hk.data(), smp.key, m * sizeof(uint64_t));
            be_.d2h(hs.data(), smp.sa, m * sizeof(idx_t));
            be_.sync();
            uint64_t bad_order = 0, bad_pos = 0, first_o = ~0ull, first_p = ~0ull;
            for (uint64_t i = 0; i < m; ++i) {
                if ((uint64_t)hs[i] >= n) { ++bad_pos; if (first_p == ~0ull) first_p = i; }
                if (i && hk[i] < hk[i - 1]) { ++bad_order; if (first_o == ~0ull) first_o = i; }
            }
            std::fprintf(stderr, "[check] sample sort: m %llu passes %u unified %d | keys out of order %llu (first at %llu) | positions >= n %llu (first at %llu",
                         (unsigned long long)m, rs.passes, (int)rs.unified, (unsigned long long)bad_order, (unsigned long long)first_o,
                         (unsigned long long)bad_pos, (unsigned long long)first_p);
            if (first_p != ~0ull) {
                std::fprintf(stderr, ":");
                for (uint64_t i = first_p > 2 ? first_p - 2 : 0; i < first_p + 6 && i < m; ++i) std::fprintf(stderr, " [%llu] %016llx/%llu", (unsigned long long)i, (unsigned long long)hk[i], (unsigned long long)hs[i]);
            }
            std::fprintf(stderr, ")\n[check] inversions at:");
            uint64_t shown = 0;
            for (uint64_t i = 1; i < m && shown < 200; ++i)
                if (hk[i] < hk[i - 1]) { std::fprintf(stderr, " %llu", (unsigned long long)i); ++shown; }
            std::fprintf(stderr, "\n[check] bucket starts:");
            std::vector<uint64_t> hb(64);
            be_.d2h(hb.data(), rs.segs.seg_start, 64 * sizeof(uint64_t));
            be_.sync();
            for (uint64_t i = 0; i < 64 && (i == 0 || hb[i] > 0) && hb[i] <= m; ++i) std::fprintf(stderr, " %llu", (unsigned long long)hb[i]);
            std::fprintf(stderr, "\n");
        }
        CAPS_LAUNCH((pick_pivots_kernel<idx_t>), (p + 255) / 256, 256, be_, (const uint64_t*)smp.key, (const idx_t*)smp.sa,
                    m, p, pl_.pkey, pl_.psa);
        e3_ = be_.record();

        // ---- locate (a7/a8)
        const uint32_t np = p - 1;
        const uint32_t bpr = (np + 255) / 256;
        if (n / p < 8ull * np)          // short subarrays: galloping searches (locate_kernel)
            CAPS_LAUNCH((locate_kernel<idx_t, BITS, true>), capped_grid((uint64_t)p * bpr, 256), 256, be_, (const uint32_t*)pl_.P, n,
                        (const uint64_t*)pl_.seg1.seg_start, p, (const uint64_t*)cur.key, (const idx_t*)cur.sa,
                        (const uint64_t*)pl_.pkey, (const idx_t*)pl_.psa, np, pl_.Pm);
        else
            CAPS_LAUNCH((locate_kernel<idx_t, BITS, false>), capped_grid((uint64_t)p * bpr, 256), 256, be_, (const uint32_t*)pl_.P, n,
                        (const uint64_t*)pl_.seg1.seg_start, p, (const uint64_t*)cur.key, (const idx_t*)cur.sa,
                        (const uint64_t*)pl_.pkey, (const idx_t*)pl_.psa, np, pl_.Pm);
        e4_ = be_.record();

        // ---- partition sizes, offsets, collate (a9)
        CAPS_LAUNCH((partition_partial_kernel<idx_t>), ((p + 255) / 256) * PART_CHUNKS, 256, be_, (const idx_t*)pl_.Pm, p, p, pl_.partial);
        CAPS_LAUNCH((partition_sizes_kernel<idx_t>), ((p + 255) / 256) * PART_CHUNKS, 256, be_, (const idx_t*)pl_.Pm, p, p,
                    (const uint64_t*)pl_.partial, pl_.ruler, pl_.sizes);
        CAPS_LAUNCH(scan_sizes_kernel, 1, 1024, be_, (const uint64_t*)pl_.sizes, p, pl_.seg2.seg_start);
        prepare_segments(pl_.seg2, n / TILE_E + p + 1);
        uint64_t out2[2];
        be_.d2h(out2, pl_.seg2.out2, sizeof out2);
        be_.sync();                                   // out2 = {#tiles, largest partition}
        const uint32_t n_tiles2 = (uint32_t)out2[0];
        max_part_ = out2[1];
        // A partition longer than a tile is bucketed by key range before its tile sort; that
        // bucket split can read the partition straight from the sorted subarrays (through the
        // transposed partition matrix): no separate collate pass.  Otherwise (tiny inputs):
        // collate as the reference does (cpp:343-358), then tile-sort the partitions.
        const bool fused = max_part_ > TILE_E;
        uint32_t* tile_plan = reinterpret_cast<uint32_t*>(pl_.desc);      // the tile descriptors are idle here
        RunSrc<idx_t> rsrc;
        ElemBuf<idx_t> in2 = cur, out2buf = oth;
        if (fused) {
            CAPS_LAUNCH((transpose_kernel<idx_t>), capped_grid((uint64_t)((p + 31) / 32) * ((p + 1 + 31) / 32), 256), 256, be_, (const idx_t*)pl_.Pm, p, p + 1,
                        pl_.PmT);
            CAPS_LAUNCH((transpose_kernel<idx_t>), capped_grid((uint64_t)((p + 31) / 32) * ((p + 31) / 32), 256), 256, be_, (const idx_t*)pl_.ruler, p, p,
                        pl_.rulerT);
            CAPS_LAUNCH((runs_plan_kernel<idx_t>), (n_tiles2 + 255) / 256, 256, be_, pl_.seg2.desc(), (const idx_t*)pl_.rulerT, p,
                        tile_plan);
            rsrc.PmT = pl_.PmT;
            rsrc.rulerT = pl_.rulerT;
            rsrc.sub_start = pl_.seg1.seg_start;
            rsrc.first_run = tile_plan;
            rsrc.G1 = p;
        } else {
            BackendEvent k0 = be_.record();
            CAPS_LAUNCH((collate_plan_kernel<idx_t>), (n_tiles1 + 255) / 256, 256, be_, pl_.seg1.desc(), p, (const idx_t*)pl_.Pm,
                        tile_plan);
            CAPS_LAUNCH((collate_kernel<idx_t>), n_tiles1, TILE_NT, be_, pl_.seg1.desc(), p, (const idx_t*)pl_.Pm,
                        (const idx_t*)pl_.ruler, (const uint64_t*)pl_.seg2.seg_start, (const uint32_t*)tile_plan,
                        (const uint64_t*)cur.key, (const idx_t*)cur.sa, oth.key, oth.sa);
            BackendEvent k1 = be_.record();
            collate_clock_.spans.push_back({k0, k1});
            collate_clock_.elems.push_back(n);
            in2 = oth;
            out2buf = cur;
        }
        e5_ = be_.record();

        // ---- phase 2 (a10): sort every partition; a partition's last step emits its LCPs
        SortOpts o2;
        o2.need_lcp = true;
        o2.skip_finished = true;
        o2.bk = &pl_.bk;                    // partition j holds keys in [pivot j-1, pivot j]
        o2.range_mode = 1;
        o2.pkey = pl_.pkey;
        if (fused) o2.runs = &rsrc;
        set_final(o2, dSA, dLCP);
        SortResult<idx_t> r2 = seg_sort<BITS>(pl_.seg2, n_tiles2, max_part_, in2, out2buf, n, o2, true);
        passes2_ = r2.passes;
        e6_ = be_.record();

        // ---- gather SA/LCP + partition-boundary LCPs (a11)
        finalize<idx_t, BITS>(be_, pl_.P, n, r2, dSA, dLCP);
        e7_ = be_.record();
    }

    // ---- the direct path ------------------------------------------------------------------------------------
    // Phase 1 of the samplesort (sort_subarrays) and locate_pivots only serve to find out which partition every
    // suffix belongs to; the order they establish inside the subarrays is discarded when the partitions are sorted.
    // With 64-bit keys the partition of a suffix can be read off its key: the pivots are sampled from the text
    // itself, and ONE scatter distributes the text into groups of PG consecutive partitions (level A: splitter table
    // in LDS, fixed-capacity region per group, no count pass).  Level B is the per-partition sort of the samplesort
    // path, run on the groups: bucket split by key range between the group's pivots -> tile sort -> SA, LCP.
    // Returns false (nothing of the result written) when the keys cannot balance the groups: two pivots with one key
    // or an overflowing group -- long repeats; the samplesort path then does the build.
    template <int BITS>
    bool run_direct(idx_t* dSA, idx_t* dLCP, uint32_t PG, uint32_t K1, BackendEvent e1, bool allow_k32 = true, bool defer = true)
    {
        const uint64_t n = pl_.n, m = pl_.m;
        const uint32_t p = pl_.p;
        // ---- pivots (a6) from samples of the text
        CAPS_LAUNCH((sample_text_kernel<idx_t, BITS>), (uint32_t)((m + 255) / 256), 256, be_, (const uint32_t*)pl_.P, (uint64_t)0, n, m,
                    pl_.SA_.key, pl_.SA_.sa);
        CAPS_LAUNCH(uniform_segments_kernel, 1, 256, be_, pl_.segS.seg_start, 1u, m, m);
        prepare_segments(pl_.segS, tiles_of(m));
        SortOpts os;
        os.bk = &pl_.bk;
        os.unify = true;
        os.keys_only = true;                          // (this path uses the pivots' keys only)
        SortResult<idx_t> rs = seg_sort<BITS>(pl_.segS, tiles_of(m), m, pl_.SA_, pl_.SB_, m, os, false);
        passesS_ = rs.passes;
        ElemBuf<idx_t> smp = rs.uniform();
        CAPS_LAUNCH((pick_pivots_kernel<idx_t>), (p + 255) / 256, 256, be_, (const uint64_t*)smp.key, (const idx_t*)smp.sa,
                    m, p, pl_.pkey, pl_.psa);
        be_.memset(pl_.dstat, 0, 4 * sizeof(uint64_t));
        uint32_t* dflag = reinterpret_cast<uint32_t*>(pl_.dstat + 2);
        CAPS_LAUNCH(group_keys_kernel, (p + 255) / 256, 256, be_, (const uint64_t*)pl_.pkey, p, PG, K1, pl_.gkey, dflag);
        CAPS_LAUNCH(skew_probe_kernel, (p + 255) / 256, 256, be_, (const uint64_t*)pl_.pkey, p, PG, K1, dflag + 2);
        uint32_t probe[4];
        be_.d2h(probe, dflag, sizeof probe);
        be_.sync();                                   // {pivot-key ties, -, skewed, longest run of equal pivot keys}

        // ---- how level B will cut the groups into buckets.  Linear: interpolation between the group's end keys, buckets
        //      in fixed-capacity slots, no count pass -- right when the keys are about uniform inside a group.  Quantile:
        //      bucket boundaries are quantiles of QUANTILE_SPB x more samples (count pass + exact scatter); the groups are
        //      runs of KPG buckets, and groups that end at one and the same key (a frequent key: a long repeat, an N-block)
        //      share the room of all of them.  Chosen when the pivots say so (ties or skew); CAPS_SA_DIRECT_MODE forces.
        const ElemBuf<idx_t>& A = pl_.A;
        const uint32_t n_tilesA = tiles_of(n);
        const char* sub_env = std::getenv("CAPS_SA_DIRECT_SUB");          // measurement: 1 = one stream per group
        // sub-streams only when each gets enough tiles to be of even size (a stream that outgrows its region voids the attempt)
        uint32_t SUB = (n + GA_E - 1) / GA_E >= 32ull * DIRECT_SUB ? DIRECT_SUB : 1u;
        if (sub_env && std::atoi(sub_env) >= 1 && (uint32_t)std::atoi(sub_env) <= DIRECT_SUB) SUB = (uint32_t)std::atoi(sub_env);
        const uint32_t n_streams = K1 * SUB;
        uint64_t capA = A.region_bytes / ((sizeof(uint64_t) + sizeof(idx_t)) * (uint64_t)n_streams);
        const uint64_t idx_max = (uint64_t)std::numeric_limits<idx_t>::max() - TILE_E;
        if (capA > idx_max / n_streams) capA = idx_max / n_streams;     // region offsets are idx_t in the scatter
        capA = test_stream_cap(capA, n, n_streams);
        uint64_t* a_key = A.key;

        const char* mode_env = std::getenv("CAPS_SA_DIRECT_MODE");
        bool quantile = probe[0] != 0 || probe[2] != 0;
        if (mode_env && std::string(mode_env) == "linear") quantile = false;
        if (mode_env && std::string(mode_env) == "quantile") quantile = true;
        const uint64_t NBt = (n + BUCKET_Q - 1) / BUCKET_Q;
        const uint32_t KPG = (uint32_t)((NBt + K1 - 1) / K1);
        const uint64_t NB = (uint64_t)K1 * KPG;
        uint64_t m2 = NB * QUANTILE_SPB;
        if (m2 > n / 4) m2 = n / 4;
        if (quantile && (KPG < 2 || KPG > BUCKET_LDS || NB > pl_.bk.nb_cap || m2 < 8 * NB || m2 > (1ull << 31))) quantile = false;
        // a single key that covers a large part of the text (a^n) is the samplesort path's business: its exact comparator
        // splits such a stretch over the partitions, keys cannot.  Up to a quarter of the text in one key is fine here: all
        // its suffixes land in one bucket, which the LCP-merge passes finish (20 passes over n / 4 elements at most)
        if (quantile && probe[3] > p / 4) { path_fallback_ = CAPS_SA_FB_PIVOT_TIES; return false; }
        if (!quantile && probe[0] != 0) { path_fallback_ = CAPS_SA_FB_PIVOT_TIES; return false; }
        uint64_t *rstart = nullptr, *rcap = nullptr;
        direct_quantile_ = quantile ? 1u : 0u;
        // 32-bit keys (text.h key32_of; CAPS_SA_KEYS=32): linear mode on 2-bit texts; the elements shrink from 8 + w to 4 + w
        // bytes through level A, level B and the tile sort.  A slot that overflows in level B sends the build back here with
        // 64-bit keys.  Measured at C3: level A 13.4 -> 11.1 ms, but level B 16.6 -> 18.1 and the tile sort 17.2 ->
        // 20.3 ms -- the group's shift + 32 bits are ~20 bases, about ten elements of a tile tie and every tile then waits for
        // text reads -- 52.3 against 49.8 ms per build: NOT the default.  (Its place is the exchange of a sharded build, where
        // a third fewer bytes cross xGMI: shard.h scatter() takes them whenever world > 1.)
        const char* keys_env = std::getenv("CAPS_SA_KEYS");
        const bool k32 = allow_k32 && !quantile && BITS == 2 && keys_env && std::string(keys_env) == "32";
        direct_k32_ = k32 ? 1u : 0u;
        idx_t* a_sa = reinterpret_cast<idx_t*>(reinterpret_cast<char*>(A.key) + (uint64_t)n_streams * capA * (k32 ? sizeof(uint32_t) : sizeof(uint64_t)));
        if (k32) CAPS_LAUNCH(group_shift_kernel, (K1 + 255) / 256, 256, be_, (const uint64_t*)pl_.gkey, K1, pl_.gshift);
        if (quantile) {
            // more samples, sorted in the (still idle) big buffers; their quantiles are the bucket boundaries and, every KPG-th, the group keys
            CAPS_LAUNCH((sample_text_kernel<idx_t, BITS>), (uint32_t)((m2 + 255) / 256), 256, be_, (const uint32_t*)pl_.P, (uint64_t)0, n, m2,
                        pl_.A.key, pl_.A.sa);
            SegBufs sseg = pl_.seg1;
            sseg.G = 1;
            CAPS_LAUNCH(uniform_segments_kernel, 1, 256, be_, sseg.seg_start, 1u, m2, m2);
            prepare_segments(sseg, tiles_of(m2));
            SortOpts oq;
            oq.bk = &pl_.bk;
            oq.unify = true;
            // split by the group keys of the FIRST sample (K1 buckets, then tile sort + merge passes inside them).  One segment
            // of 47 M skewed keys under a linear map has more buckets than an LDS histogram holds: count pass and scatter then
            // do one global atomic per sample, most of them on a few hot buckets -- 3.3 + 3.7 ms of the 10 ms this sort took
            // at 3e9 (the whole build: 144 ms)
            oq.knots = pl_.gkey;
            oq.knots_per_parent = K1;
            oq.skewed_keys = probe[2] != 0;
            oq.keys_only = true;
            SortResult<idx_t> rq = seg_sort<BITS>(sseg, tiles_of(m2), m2, pl_.A, pl_.B, m2, oq, false);
            ElemBuf<idx_t> smp2 = rq.uniform();
            CAPS_LAUNCH(knots_kernel, (uint32_t)((NB + 255) / 256), 256, be_, (const uint64_t*)smp2.key, m2, NB, KPG, K1, pl_.knots, pl_.gkey);
            rstart = pl_.rstart;
            rcap = pl_.rcap;
            // every stream gets `token` elements of room on top of its share (the stray elements of an expectedly empty stream);
            // the shares add up to (capA - token) * streams, so all regions together are exactly the capA * streams elements
            // of the buffer -- which only works when the budget per stream can carry the token (found by tools/stress_gpu.py:
            // sub-streams forced on a 9-tile text gave regions that overran the buffer)
            const uint64_t token = 2 * GA_E / K1 + 16;
            if (capA <= 2 * token) { path_fallback_ = CAPS_SA_FB_SHAPE; direct_quantile_ = 0; return false; }
            CAPS_LAUNCH(group_caps_kernel, (K1 + 255) / 256, 256, be_, (const uint64_t*)pl_.knots, NB, KPG, K1, SUB, capA - token, token, rcap);
            CAPS_LAUNCH(scan_sizes_kernel, 1, 1024, be_, (const uint64_t*)rcap, n_streams, rstart);
        }
        CAPS_LAUNCH(split_lut_kernel, (SPLIT_LUT_CELLS + 256) / 256, 256, be_, (const uint64_t*)pl_.gkey, K1 - 1, pl_.glut, dflag + 1);
        e3_ = be_.record();

        // ---- level A (a9 without a8): the text -> K1 groups, each as SUB sub-streams in their own regions of buffer A
        CAPS_LAUNCH(uniform_segments_kernel, 1, 256, be_, pl_.seg1.seg_start, 1u, n, n);
        SegBufs whole = pl_.seg1;
        whole.G = 1;
        prepare_segments(whole, n_tilesA);
        BucketParams one = make_bucket_params(0, ~0ull, K1);            // only B is used by MAP_SPLIT
        const uint64_t b01[2] = {0, K1};
        be_.h2d(pl_.bk.params, &one, sizeof one);
        be_.h2d(pl_.bk.bstart, b01, sizeof b01);
        be_.memset(pl_.dcur, 0, (size_t)n_streams * sizeof(idx_t));
        // CAPS_SA_LEVEL_A=tile: the same distribution by bucket_scatter_kernel<SRC_TEXT, MAP_SPLIT> on TILE_E positions (cross-check)
        const char* la = std::getenv("CAPS_SA_LEVEL_A");
        const bool big_tiles = k32 || !(la && std::string(la) == "tile");
        {
            BackendEvent s0 = be_.record();
            if (k32)
                CAPS_LAUNCH((group_scatter_kernel<idx_t, BITS, uint32_t>), (uint32_t)((n + GA_E - 1) / GA_E), TILE_NT, be_, (const uint32_t*)pl_.P,
                            packed_words(n, BITS), (uint64_t)0, n, (const uint64_t*)pl_.gkey, K1, (const uint16_t*)pl_.glut,
                            (const uint32_t*)(dflag + 1), SUB, capA, pl_.dcur, reinterpret_cast<uint32_t*>(a_key), a_sa, 0u, 1u,
                            (const uint64_t*)rstart, (const uint64_t*)rcap, (const uint8_t*)pl_.gshift, 0u, K1);
            else if (big_tiles)
                CAPS_LAUNCH((group_scatter_kernel<idx_t, BITS>), (uint32_t)((n + GA_E - 1) / GA_E), TILE_NT, be_, (const uint32_t*)pl_.P,
                            packed_words(n, BITS), (uint64_t)0, n, (const uint64_t*)pl_.gkey, K1, (const uint16_t*)pl_.glut,
                            (const uint32_t*)(dflag + 1), SUB, capA, pl_.dcur, a_key, a_sa, 0u, 1u, (const uint64_t*)rstart, (const uint64_t*)rcap,
                            (const uint8_t*)nullptr, 0u, K1);
            else
                CAPS_LAUNCH((bucket_scatter_kernel<idx_t, BITS, SRC_TEXT, MAP_SPLIT>), n_tilesA, TILE_NT, be_, whole.desc(), (const uint32_t*)pl_.P,
                            packed_words(n, BITS), (uint64_t)0, (const uint64_t*)nullptr, (const idx_t*)nullptr, RunSrc<idx_t>(),
                            (const BucketParams*)pl_.bk.params, (const uint64_t*)pl_.bk.bstart, (const uint64_t*)nullptr, capA,
                            pl_.dcur, a_key, a_sa, (const BucketParams*)nullptr, (const uint32_t*)nullptr,
                            (const uint64_t*)pl_.gkey, (const uint16_t*)pl_.glut, (const uint32_t*)(dflag + 1), SUB, 0u, 1u, (const uint16_t*)nullptr,
                            Spill<idx_t>());
            BackendEvent s1 = be_.record();
            scatter_clock_.spans.push_back({s0, s1});
            scatter_clock_.elems.push_back(n);
            la0_ = s0;
            la1_ = s1;
        }
        SegBufs groups = pl_.seg2;
        groups.G = n_streams;
        CAPS_LAUNCH((slot_segments_kernel<idx_t>), (n_streams + 256) / 256, 256, be_, (const idx_t*)pl_.dcur, K1, SUB, big_tiles ? 1u : 0u, capA,
                    (const uint64_t*)rstart, (const uint64_t*)rcap, groups.seg_start, groups.seg_end, pl_.dstat);
        ::caps::prepare_segments(be_, groups, n / TILE_E + n_streams + 1, nullptr, nullptr, true);
        uint64_t out2[2], dstat[4];
        // waves (set_waves): the groups' sizes on the host tell where every wave's slice of the suffix array starts
        uint32_t W = (k32 || waves_ < 2 || !wave_scratch_.key) ? 1u : std::min<uint32_t>(waves_, K1);
        std::vector<idx_t> h_cur;
        if (W > 1) {
            h_cur.resize(n_streams);
            be_.d2h(h_cur.data(), pl_.dcur, (size_t)n_streams * sizeof(idx_t));
        }
        be_.d2h(out2, groups.out2, sizeof out2);
        be_.d2h(dstat, pl_.dstat, sizeof dstat);
        be_.sync();                                   // out2 = {#tiles, largest sub-stream (clamped to its region)}
        direct_groups_ = K1;
        direct_max_group_ = dstat[1] ? dstat[1] : out2[1];
        if (dstat[1] != 0 || dstat[0] != n) { path_fallback_ = CAPS_SA_FB_GROUP_OVERFLOW; return false; }
        max_part_ = out2[1];
        e2_ = e1;                                     // no sort_subarrays, no locate_pivots
        e4_ = e3_;
        e5_ = be_.record();

        // ---- level B (a10): every group -> buckets between its pivots -> tile sort -> SA, LCP
        SortOpts o2;
        o2.need_lcp = true;
        o2.skip_finished = true;
        o2.bk = &pl_.bk;
        o2.range_mode = 1;                            // group g holds the keys in (gkey[g-1], gkey[g]]
        o2.pkey = pl_.gkey;
        o2.part_total = K1;
        o2.sub = SUB;
        o2.seg_ends = true;
        if (quantile) {
            o2.knots = pl_.knots;
            o2.knots_per_parent = KPG;
            o2.in_extent = (uint64_t)n_streams * capA;
            // no count pass (segmented_sort "speculative split by knots") -- unless a wave of this build already had to redo its split
            o2.spill_slots = spill_stats_[1] == 0;
            o2.spill_stats = spill_stats_;
            o2.hot_tile_rec = pl_.seg1.tile_rec;     // (level A's one-segment tables: idle until finalize() sorts the letter-run buckets)
        }
        o2.skewed_keys = quantile && probe[2] != 0 && !std::getenv("CAPS_SA_TRY_LINEAR_TILES");   // (the variable: measurement)
        if (k32) { o2.k32 = true; o2.range_mode = 2; o2.gshift = pl_.gshift; }
        o2.in_key = a_key;
        o2.in_sa = a_sa;
        // large groups of equal keys (tandem arrays, repeat families) are not compared through the text but re-keyed deeper after
        // the sort (kernels.h "Deferred ties"); the flags live in the fine-count table, idle outside the equalised split
        if (defer && !k32 && !std::getenv("CAPS_SA_NO_DEFER")) o2.defer_flags = pl_.bk.fcount;
        o2.run_tables = &pl_.seg1;                    // (level A's one-segment tables: idle by now; the bucket tables stay intact)
        passes2_ = 0;
        run_buckets_ = 0;
        tie_groups_ = tie_elems_ = 0;
        tie_levels_ = 0;
        // wave boundaries: consecutive groups, about n / W suffixes each, never more than the scratch arrays hold
        std::vector<uint32_t> wave_end;
        if (W > 1) {
            auto group_size = [&](uint32_t g) {
                uint64_t z = 0;
                for (uint32_t x = 0; x < SUB; ++x) { const uint32_t s = g * SUB + x; z += (uint64_t)h_cur[big_tiles ? (size_t)(s % SUB) * K1 + s / SUB : s]; }
                return z;
            };
            const uint64_t target = std::min<uint64_t>((n + W - 1) / W, wave_scratch_elems_);
            uint64_t acc = 0;
            for (uint32_t g = 0; g < K1; ++g) {
                const uint64_t z = group_size(g);
                if (z > wave_scratch_elems_) { wave_end.clear(); break; }            // a group the scratch cannot take: one wave
                if (acc && acc + z > target) { wave_end.push_back(g); acc = 0; }
                acc += z;
                if (g + 1 == K1) wave_end.push_back(K1);
            }
            W = wave_end.empty() ? 1u : (uint32_t)wave_end.size();
        }
        uint64_t base = 0;
        for (uint32_t w = 0; w < W; ++w) {
            const uint32_t g0 = W > 1 && w ? wave_end[w - 1] : 0u, g1 = W > 1 ? wave_end[w] : K1;
            SegBufs gw = groups;
            uint32_t n_tiles_w = (uint32_t)out2[0];
            uint64_t max_len_w = out2[1], elems_w = n;
            SortOpts ow = o2;
            if (W > 1) {
                // the wave's streams as a segment list of their own (the tile tables are rebuilt for it)
                gw.seg_start = groups.seg_start + (size_t)g0 * SUB;
                gw.seg_end = groups.seg_end + (size_t)g0 * SUB;
                gw.G = (g1 - g0) * SUB;
                elems_w = 0;
                max_len_w = 0;
                uint64_t tiles = 0;
                for (uint32_t s = g0 * SUB; s < g1 * SUB; ++s) {
                    const uint64_t z = (uint64_t)h_cur[big_tiles ? (size_t)(s % SUB) * K1 + s / SUB : s];     // (slot_segments_kernel's layout)
                    elems_w += z;
                    max_len_w = std::max(max_len_w, z);
                    tiles += (z + TILE_E - 1) / TILE_E;
                }
                n_tiles_w = (uint32_t)tiles;
                ::caps::prepare_segments(be_, gw, tiles + 1, nullptr, nullptr, true);
                ow.part_off = g0;
                if (quantile) { ow.knots = pl_.knots + (size_t)g0 * KPG; ow.knots_have_prev = g0 > 0; }
            }
            set_final(ow, dSA + base, dLCP + base);
            if (elems_w) {
                SortResult<idx_t> r2 = seg_sort<BITS>(gw, n_tiles_w, max_len_w, W > 1 ? wave_scratch_ : pl_.A, pl_.B, elems_w, ow, true);
                if (r2.failed) {                                                                // a slot overflowed under 32-bit keys: again with 64
                    if (sink_) sink_->reset(be_);
                    return run_direct<BITS>(dSA, dLCP, PG, K1, e1, false, defer);
                }
                passes2_ = std::max(passes2_, r2.passes);
                run_buckets_ += (uint32_t)r2.run_buckets.size();
                finalize<idx_t, BITS>(be_, pl_.P, n, r2, dSA + base, dLCP + base);
                // the deferred groups did not fit the work memory (most of the text in large groups of equal keys): the whole
                // build again, every tie settled by comparison as before
                if (r2.msd_failed) {
                    if (sink_) sink_->reset(be_);
                    return run_direct<BITS>(dSA, dLCP, PG, K1, e1, allow_k32, false);
                }
                tie_groups_ += r2.msd_groups;
                tie_elems_ += r2.msd_elems;
                tie_levels_ = std::max(tie_levels_, r2.msd_levels);
                if (base) CAPS_LAUNCH((wave_head_lcp_kernel<idx_t, BITS>), 1, 64, be_, (const uint32_t*)pl_.P, n, (const idx_t*)dSA, dLCP, base);
            }
            if (sink_ && elems_w) sink_->wave_done(be_, base, elems_w);
            base += elems_w;
        }
        sink_served_ = sink_ != nullptr;
        waves_used_ = W;
        e6_ = be_.record();
        e7_ = be_.record();
        return true;
    }

    template <int BITS>
    void run(idx_t* dSA, idx_t* dLCP, caps_sa_stats* st, BackendEvent e0, BackendEvent e1)
    {
        const uint64_t n = pl_.n;
        const uint32_t p = pl_.p;
        be_.memset(pl_.pass_elems, 0, kMaxPasses * sizeof(uint64_t));

        if (p < 2) {
            // Below the reference's valid domain (n < 32 or p_eff < 2) the
            // samplesort degenerates to ONE segment: tile sort + merge passes over [0, n).
            CAPS_LAUNCH(uniform_segments_kernel, 1, 256, be_, pl_.seg1.seg_start, 1u, n, n);
            prepare_segments(pl_.seg1, tiles_of(n));
            SortOpts o;
            o.from_text = true;
            o.need_lcp = true;
            o.bk = &pl_.bk;                     // one segment: bucket it by key range too
            set_final(o, dSA, dLCP);
            SortResult<idx_t> r = seg_sort<BITS>(pl_.seg1, tiles_of(n), n, pl_.A, pl_.B, n, o, true);
            passes1_ = r.passes;
            e2_ = e3_ = e4_ = e5_ = e6_ = be_.record();
            finalize<idx_t, BITS>(be_, pl_.P, n, r, dSA, dLCP);
            e7_ = be_.record();
        } else {
            // direct path unless the text (long periodic stretches: keys alone cannot split them), the shape
            // (tiny inputs) or CAPS_SA_PATH=classic says otherwise
            uint32_t PG = 0, K1 = 0;
            const char* force = std::getenv("CAPS_SA_PATH");
            if (force && std::string(force) == "classic") path_fallback_ = CAPS_SA_FB_FORCED;
            else path_fallback_ = (uint32_t)direct_shape(n, p, pl_.m, &PG, &K1);
            if (path_fallback_ == CAPS_SA_FB_NONE && run_direct<BITS>(dSA, dLCP, PG, K1, e1)) path_direct_ = 1;
            else {
                // a direct attempt that gave up has written nothing of the result; its kernel-clock entries stay (it cost them)
                run_classic<BITS>(dSA, dLCP);
            }
        }
        uint64_t pass_elems[kMaxPasses];
        be_.d2h(pass_elems, pl_.pass_elems, sizeof pass_elems);
        BackendEvent e8 = be_.record();
        be_.sync();

        if (st) {
            st->bits_per_char = BITS;
            st->long_runs = be_.long_runs ? 1u : 0u;
            st->ppp = pl_.ppp;
            st->max_partition = max_part_;
            st->merge_passes_phase1 = passes1_;
            st->merge_passes_phase2 = passes2_;
            st->merge_passes_samples = passesS_;
            st->workspace_bytes = pl_.bytes;
            st->ms_pack = be_.elapsed_ms(e0, e1);
            st->ms_sort_subarrays = be_.elapsed_ms(e1, e2_);
            st->ms_select_pivots = be_.elapsed_ms(e2_, e3_);
            st->ms_locate_pivots = be_.elapsed_ms(e3_, e4_);
            st->ms_partition = be_.elapsed_ms(e4_, e5_);
            st->ms_merge_partitions = be_.elapsed_ms(e5_, e6_);
            st->ms_boundary_lcp = be_.elapsed_ms(e6_, e7_);
            st->ms_output = be_.elapsed_ms(e7_, e8);
            st->ms_total = be_.elapsed_ms(e0, e8);
            auto sum = [&](KernelClock& c, double* ms, uint64_t* launches, uint64_t* elems) {
                for (size_t i = 0; i < c.spans.size(); ++i) {
                    *ms += be_.elapsed_ms(c.spans[i].first, c.spans[i].second);
                    *elems += c.elems[i];
                }
                *launches = c.spans.size();
            };
            sum(merge_clock_, &st->merge_pass_ms, &st->merge_pass_launches, &st->merge_pass_elems);
            sum(tile_clock_, &st->tile_sort_ms, &st->tile_sort_launches, &st->tile_sort_elems);
            uint64_t dummy_l = 0, dummy_e = 0;
            sum(scatter_clock_, &st->bucket_scatter_ms, &st->bucket_scatter_launches, &st->bucket_scatter_elems);
            sum(count_clock_, &st->bucket_count_ms, &dummy_l, &dummy_e);
            sum(collate_clock_, &st->collate_ms, &dummy_l, &dummy_e);
            sum(finish_clock_, &st->finish_ms, &dummy_l, &dummy_e);
            sum(runb_clock_, &st->run_bucket_ms, &dummy_l, &dummy_e);
            sum(msd_clock_, &st->msd_ms, &dummy_l, &dummy_e);
            st->path_direct = path_direct_;
            st->path_fallback = path_fallback_;
            st->direct_groups = direct_groups_;
            st->direct_quantile = direct_quantile_;
            st->run_buckets = run_buckets_;
            st->tie_groups_deferred = tie_groups_;
            st->tie_elems_deferred = tie_elems_;
            st->tie_levels = tie_levels_;
            st->result_waves = sink_served_ ? waves_used_ : 1u;
            st->n_devices = 1;
            st->direct_key_bits = path_direct_ && direct_k32_ ? 32u : 64u;
            st->direct_max_group = direct_max_group_;
            st->level_a_ms = direct_groups_ ? be_.elapsed_ms(la0_, la1_) : 0.0;
            st->slot_splits = slot_stats_[0];
            st->slot_splits_redone = slot_stats_[1];
            st->knot_slot_splits = (uint32_t)spill_stats_[0];
            st->knot_slot_splits_redone = (uint32_t)spill_stats_[1];
            st->spill_entries = spill_stats_[2];
            st->merge_pass_elems = 0;                      // elements the timed passes really merged
            for (uint32_t i = 0; i < pass_base_ && i < kMaxPasses; ++i) st->merge_pass_elems += pass_elems[i];
        }
        be_.release_events();
    }
};

}  // namespace caps
