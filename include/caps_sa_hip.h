/*
 * include/caps_sa_hip.h -- C ABI of the MI355X suffix-array / LCP-array construction path.
 *
 * This is the drop-in boundary: a plain-C shared library (libcaps_sa_hip.so) that sits
 * UNDER the reference's class surface CaPS_SA::Suffix_Array<idx> (reference
 * include/Suffix_Array.hpp:148-181).  The reference has no FFI of its own -- its only
 * caller is src/main.cpp:78-80,84-86 -- so each entry point cites the C++ member it
 * replaces.  caps-sa_amd/csrc/Suffix_Array.hpp is the host-side mirror of that class
 * whose construct() calls caps_sa_hip_build_u32/_u64; INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions: no exceptions, no exit(): every function returns 0 on success or a
 * negative CAPS_SA_E* code (the reference calls std::exit, src/Suffix_Array.cpp:33-37);
 * caps_sa_hip_last_error() returns a thread-local message.  Index width is chosen by
 * the caller exactly like src/main.cpp:76-87 (n <= UINT32_MAX -> _u32, else _u64).
 * Output is THE suffix array and LCP array of T[0..n): shorter suffix first when one is
 * a prefix of the other, bytes ordered as signed char (src/Suffix_Array.cpp:75-77),
 * LCP[0] = 0.  It does not depend on subproblem_count.
 */
#ifndef CAPS_SA_HIP_H
#define CAPS_SA_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAPS_SA_OK 0
#define CAPS_SA_EINVAL (-1)       /* bad argument (null pointer, n too large for the index type) */
#define CAPS_SA_EUNSUPPORTED (-2) /* bounded max_context on several devices / in a shard, or with fewer than two subproblems */
#define CAPS_SA_EHIP (-3)         /* HIP runtime error; see caps_sa_hip_last_error() */
#define CAPS_SA_ENOMEM (-4)       /* device or host allocation failed */
#define CAPS_SA_ENODEVICE (-5)    /* no usable GPU */
#define CAPS_SA_EALPHABET (-6)    /* the workspace was sized for a 2-bit text (caps_sa_hip_workspace_bytes_ex), the text has more than 4 bytes */

/* Per-build record; replaces the per-phase stderr lines of construct()
 * (src/Suffix_Array.cpp:469-493).  Times are HIP-event milliseconds on the build's stream. */
typedef struct caps_sa_stats {
    uint64_t n;
    uint32_t idx_bytes;            /* 4 or 8 */
    uint32_t p_eff;                /* effective subproblem count, src/Suffix_Array.cpp:24 */
    uint32_t ppp;                  /* samples per subarray, src/Suffix_Array.cpp:27 */
    uint32_t bits_per_char;        /* 2 (alphabet <= 4 symbols) or 8 */
    uint32_t merge_passes_phase1, merge_passes_phase2, merge_passes_samples;
    uint32_t long_runs;            /* 1: the text holds a periodic stretch (period <= 16 chars) of >= 1024 chars and the
                                      comparators skipped such stretches through the run table (csrc/text.h) */
    uint64_t max_partition;        /* largest partition (elements) */
    uint64_t workspace_bytes;
    double ms_total;               /* whole build, device-resident interval */
    double ms_pack;                /* alphabet scan + text packing */
    double ms_sort_subarrays;      /* a5  sort_subarrays        (cpp:161-184) */
    double ms_select_pivots;       /* a6  select_pivots         (cpp:197-222) */
    double ms_locate_pivots;       /* a8  locate_pivots         (cpp:225-249) */
    double ms_partition;           /* a9  partition_sub_subarrays (cpp:300-368) */
    double ms_merge_partitions;    /* a10 merge_sub_subarrays   (cpp:371-409) */
    double ms_boundary_lcp;        /* a11 compute_partition_boundary_lcp (cpp:431-447) */
    double ms_output;              /* copy of SA/LCP into the caller's device buffers */
    double ms_h2d, ms_d2h;         /* host-buffer entry points only */
    /* the two tile-granular kernels, summed over their launches in this build */
    double merge_pass_ms;  uint64_t merge_pass_launches;  uint64_t merge_pass_elems;
    double tile_sort_ms;   uint64_t tile_sort_launches;   uint64_t tile_sort_elems;
    /* bucketing (count + scatter kernels) and collate, same convention */
    double bucket_scatter_ms; uint64_t bucket_scatter_launches; uint64_t bucket_scatter_elems;
    double bucket_count_ms;
    double collate_ms;
    /* bucket splits done without a count pass (fixed-capacity slots) / splits that had to be redone with one */
    uint32_t slot_splits, slot_splits_redone;
    /* Which construction ran.  1 = direct path: pivots sampled from the text, ONE scatter of the text into groups of
     * partitions, then the per-partition sort (select_pivots -> partition -> merge_partitions; sort_subarrays and
     * locate_pivots are skipped: their order would be discarded).  0 = samplesort path, all six phases of construct().
     * path_fallback: why a build that tried the direct path took the samplesort path after all (CAPS_SA_FB_*). */
    uint32_t path_direct, path_fallback;
    uint32_t direct_groups;        /* groups of consecutive partitions the text was scattered into */
    uint32_t direct_quantile;      /* 1: level B's buckets were sample quantiles (skewed keys / frequent keys), 0: linear maps */
    uint64_t direct_max_group;     /* largest stream of a group (elements) */
    double level_a_ms;             /* direct path: the text -> groups scatter (also counted in bucket_scatter_ms) */
    uint32_t direct_key_bits;      /* key bits that travelled with every suffix through the direct path's passes: 64, or 32 */
    uint32_t run_buckets;          /* buckets of ONE single-letter key (the suffixes deep inside N-blocks) that were ordered by
                                      (terminator class, rest of the run, text behind it) instead of being compared: csrc/text.h */
    uint32_t result_waves;         /* host-buffer entry points: slices in which SA / LCP left the device while later groups were still
                                      being sorted (1: one copy after the build) */
    uint32_t n_devices;            /* caps_sa_hip_build_multi_*: devices that built (1 elsewhere) */
    /* caps_sa_hip_build_multi_*: host wall clock of the three stages, the slowest and the fastest device of each (ms) */
    double ms_upload_max, ms_upload_min;       /* text to the device: ONE chunked upload to the first device, fanned out to the others
                                                  over xGMI chunk by chunk; a device's figure ends when its last chunk has arrived, so
                                                  it includes the wait for the devices served before it */
    double ms_device_build_max, ms_device_build_min;   /* level A .. boundary LCPs of the device's slice */
    double ms_download_max, ms_download_min;   /* the device's slice of SA / LCP to the caller's arrays */
    /* Groups of suffixes with one and the same key (64 / bits_per_char chars) that the sort did not settle by comparison but by
     * re-keying them deeper, level by level, afterwards (csrc/kernels.h "Deferred ties": tandem arrays, repeat families), their
     * members, and the deepest level (each 64 / bits_per_char chars). */
    uint64_t tie_groups_deferred, tie_elems_deferred;
    uint32_t tie_levels;
    uint32_t lcp_bytes_on_link;    /* host-buffer entry points: bytes per LCP value on the PCIe link -- 1 when the values left the device as
                                      bytes (+ a list of the values of 255 and more) and were widened on the host, else the index width */
    /* the three parts of the last stage, summed over their launches (device time, as the kernel clocks above): gathering SA / LCP
     * with the segment-head LCPs (a11), the letter-run buckets (run_buckets), the deferred ties (tie_groups_deferred) */
    double finish_ms, run_bucket_ms, msd_ms;
    /* direct path, quantile mode: level-B splits by knots done WITHOUT the count pass (slots + a stream for what outgrows them) /
     * redone with it (the stream ran full), and the elements that took the stream */
    uint32_t knot_slot_splits, knot_slot_splits_redone;
    uint64_t spill_entries;
} caps_sa_stats;

/* sizeof(caps_sa_stats) / sizeof(caps_sa_shard_info) of THIS library.  Both structs grow at their end from release to release and
 * the entry points write all of them: a caller compiled against an older header must check these against its own sizeof before
 * passing a buffer (caps-sa_amd/_binding.py does at load time). */
uint32_t caps_sa_hip_stats_bytes(void);
uint32_t caps_sa_hip_shard_info_bytes(void);

#define CAPS_SA_FB_NONE 0
#define CAPS_SA_FB_FORCED 1        /* CAPS_SA_PATH=classic */
#define CAPS_SA_FB_SHAPE 2         /* too small / too few subproblems for a two-level distribution */
#define CAPS_SA_FB_LONG_RUNS 3     /* the text holds a long periodic stretch: keys alone cannot split it */
#define CAPS_SA_FB_PIVOT_TIES 4    /* two sampled pivots share their key */
#define CAPS_SA_FB_GROUP_OVERFLOW 5 /* a group outgrew its region (a very frequent key) */
#define CAPS_SA_FB_KEY32 6         /* sharded build: a bucket slot overflowed under 32-bit keys; repeat with caps_sa_hip_shard_set_key_bits(s, 64) */
#define CAPS_SA_FB_BOUNDED 7       /* 0 < max_context < n: the reference's own sequence, one thread per merge node (csrc/bounded.h) */

int caps_sa_hip_device_count(void);
const char* caps_sa_hip_last_error(void);
const char* caps_sa_hip_version(void);

/*
 * The host-buffer entry points (caps_sa_hip_build_*) keep their device memory (text, SA, LCP,
 * workspace) between calls -- allocating and freeing tens of GB costs far more than a build.
 * One grow-only block per process; this releases it.  (The reference frees everything in
 * clean_up() / the destructor, src/Suffix_Array.cpp:42-45,455-459; a caller that wants that
 * behaviour calls this after construct().)
 */
void caps_sa_hip_release_cache(void);

/*
 * Page-locked host memory for SA / LCP buffers (the reference mallocs them in its constructor,
 * src/Suffix_Array.cpp:20-21).  Results land in such buffers at the PCIe link rate; pageable
 * buffers work too, slower.  Returns NULL on failure (caps_sa_hip_last_error()).
 */
void* caps_sa_hip_host_alloc(uint64_t bytes);
void caps_sa_hip_host_free(void* p);

/*
 * The reference's input generator, utils/gen_rand_seq.py:9-13 -- random.seed(seed); n x random.choice(['A','C','G','T']) --
 * bit for bit: MT19937 seeded like CPython's random.seed(int) (init_by_array([seed])), every choice = the top 3 bits of one
 * 32-bit output, drawn again while >= 4 (random.choice -> _randbelow(4) -> getrandbits(3)).  Writes n letters to out (host
 * memory; the script's print() adds a newline, which the CLI remaps to 'C': callers append it themselves).  Host code, no
 * GPU: ~5 ns per letter.  BASELINE's workloads are quoted on this stream, so bench.py builds C2 / C3 from it.
 */
int caps_sa_hip_gen_rand_seq(uint32_t seed, uint64_t n, char* out);

/* Device workspace a build of n suffixes needs (bytes), for any text. */
int caps_sa_hip_workspace_bytes(uint64_t n, uint64_t subproblem_count, int idx_bytes, uint64_t* bytes);
/* The same for a text of at most 4 distinct bytes (bits_per_char = 2: everything behind the reference CLI, src/main.cpp:61-70):
 * the packed text and its run table then take n / 2 bytes instead of 2 n.  caps_sa_hip_build_device_* reads the alphabet it
 * may assume off the size of the workspace it is given; a text with more letters is refused with CAPS_SA_EALPHABET.
 * bits_per_char = 8 is caps_sa_hip_workspace_bytes. */
int caps_sa_hip_workspace_bytes_ex(uint64_t n, uint64_t subproblem_count, int idx_bytes, int bits_per_char, uint64_t* bytes);

/*
 * Replaces the body of Suffix_Array<uint32_t>::construct() / <uint64_t>
 * (src/Suffix_Array.cpp:466-494; constructor arguments of include/Suffix_Array.hpp:155).
 * T: n bytes, host memory, borrowed.  SA, LCP: n entries each, host memory owned by the
 * caller (the class allocates them in its constructor, src/Suffix_Array.cpp:20-21).
 * subproblem_count 0 -> 8192 (include/Suffix_Array.hpp:42), clamped to n/16 (cpp:24).
 * max_context 0 or >= n: THE suffix array and LCP array of T.  0 < max_context < n: the reference's bounded-context result
 * (comparisons stop after max_context chars, ties keep the reference's merge history: csrc/bounded.h; parity unpinned by any
 * reference-held vector, a compatibility mode that is seconds, not milliseconds).  device: HIP device ordinal.
 */
int caps_sa_hip_build_u32(const char* T, uint64_t n, uint64_t subproblem_count, uint64_t max_context,
                          uint32_t* SA, uint32_t* LCP, int device, caps_sa_stats* stats);
int caps_sa_hip_build_u64(const char* T, uint64_t n, uint64_t subproblem_count, uint64_t max_context,
                          uint64_t* SA, uint64_t* LCP, int device, caps_sa_stats* stats);

/*
 * construct() on several GPUs of one node from ONE process (what Suffix_Array(T, n, p, ctx, devices)
 * calls): devices[0 .. n_devices) are HIP device ordinals, one rank of the sharded direct path each (a device may be listed
 * more than once).  The text is copied to every device.  Default: NO element crosses a link -- every device classifies the
 * whole text (level A), keeps the suffixes of the groups of partitions it owns (1 / n_devices of them), sorts them and
 * copies its slice of SA / LCP into the caller's arrays; per device: the packed text + about 2.7 x 16 B x n / n_devices of
 * work arrays.  CAPS_SA_SHARD_EXCHANGE=1 selects the variant in which every device classifies every n_devices-th tile only
 * and the blocks of (key, sa) are then copied device to device over xGMI (8.8 B per suffix).  n_devices = 1 is
 * caps_sa_hip_build_*.  Texts the direct path does not
 * take (stats->path_fallback says why) are built on devices[0] alone.  stats: host wall-clock per stage, all devices.
 */
int caps_sa_hip_build_multi_u32(const char* T, uint64_t n, uint64_t subproblem_count, uint64_t max_context,
                                uint32_t* SA, uint32_t* LCP, const int* devices, int n_devices, caps_sa_stats* stats);
int caps_sa_hip_build_multi_u64(const char* T, uint64_t n, uint64_t subproblem_count, uint64_t max_context,
                                uint64_t* SA, uint64_t* LCP, const int* devices, int n_devices, caps_sa_stats* stats);

/*
 * Same construction with everything resident in HBM: dT (n bytes), dSA, dLCP (n entries)
 * are device pointers on the current device; hip_stream is a hipStream_t (NULL = default
 * stream).  workspace: device memory of at least caps_sa_hip_workspace_bytes(), or NULL
 * to let the call allocate and free it.  Returns after the stream work has completed.
 */
int caps_sa_hip_build_device_u32(const void* dT, uint64_t n, uint64_t subproblem_count, uint64_t max_context,
                                 void* dSA, void* dLCP, void* workspace, uint64_t workspace_bytes,
                                 void* hip_stream, caps_sa_stats* stats);
int caps_sa_hip_build_device_u64(const void* dT, uint64_t n, uint64_t subproblem_count, uint64_t max_context,
                                 void* dSA, void* dLCP, void* workspace, uint64_t workspace_bytes,
                                 void* hip_stream, caps_sa_stats* stats);

/*
 * Device verifier in the spirit of the reference's (never called) is_sorted
 * (src/Suffix_Array.cpp:512-536) plus a permutation check; byte loops on the raw text,
 * independent of the build kernels' packed text.  *n_errors = 0 iff (dSA, dLCP) is the
 * suffix array and LCP array of dT.
 */
int caps_sa_hip_verify_device_u32(const void* dT, uint64_t n, const void* dSA, const void* dLCP,
                                  void* hip_stream, uint64_t* n_errors);
int caps_sa_hip_verify_device_u64(const void* dT, uint64_t n, const void* dSA, const void* dLCP,
                                  void* hip_stream, uint64_t* n_errors);

/* The same checks on a SLICE of the arrays (cnt entries starting anywhere in the suffix array: one rank's share of a
 * sharded build): values in range and none twice within the slice, adjacent order, exact LCP; is_head != 0: entry 0 is
 * the head of the suffix array (its LCP must be 0), else entry 0's LCP is not checked (its predecessor is not in the slice). */
int caps_sa_hip_verify_slice_device_u32(const void* dT, uint64_t n, const void* dSA, const void* dLCP, uint64_t cnt,
                                        int is_head, void* hip_stream, uint64_t* n_errors);
int caps_sa_hip_verify_slice_device_u64(const void* dT, uint64_t n, const void* dSA, const void* dLCP, uint64_t cnt,
                                        int is_head, void* hip_stream, uint64_t* n_errors);

/* ---- kernel-level entry points (host buffers) for differential tests -------------- */

/* merge_sort (src/Suffix_Array.cpp:112-129) of an arbitrary list of cnt distinct suffix
 * positions: out_sa = sorted list, out_lcp = its LCP array (out_lcp[0] = 0). */
int caps_sa_hip_sort_suffixes_u32(const char* T, uint64_t n, const uint32_t* idx, uint64_t cnt,
                                  uint32_t* out_sa, uint32_t* out_lcp, int device);
int caps_sa_hip_sort_suffixes_u64(const char* T, uint64_t n, const uint64_t* idx, uint64_t cnt,
                                  uint64_t* out_sa, uint64_t* out_lcp, int device);

/* sort_partition over every partition (src/Suffix_Array.cpp:388-404): independent sorts of
 * the G consecutive segments [seg_start[g], seg_start[g+1]) of the suffix list idx (seg_start:
 * host u64[G+1], seg_start[0] = 0, seg_start[G] = cnt).  out_lcp at a segment head is the lcp
 * with the last suffix of the previous non-empty segment (cpp:431-447); out_lcp[0] = 0. */
int caps_sa_hip_sort_segments_u32(const char* T, uint64_t n, const uint32_t* idx, uint64_t cnt,
                                  const uint64_t* seg_start, uint64_t G, uint32_t* out_sa, uint32_t* out_lcp, int device);
int caps_sa_hip_sort_segments_u64(const char* T, uint64_t n, const uint64_t* idx, uint64_t cnt,
                                  const uint64_t* seg_start, uint64_t G, uint64_t* out_sa, uint64_t* out_lcp, int device);

/* merge (src/Suffix_Array.cpp:48-109): X, Y sorted suffix runs with their LCP arrays ->
 * Z (len_x + len_y) and LCP_z. */
int caps_sa_hip_merge_u32(const char* T, uint64_t n, const uint32_t* X, uint64_t len_x, const uint32_t* Y,
                          uint64_t len_y, const uint32_t* LCP_x, const uint32_t* LCP_y, uint32_t* Z,
                          uint32_t* LCP_z, int device);   /* LCP_x/LCP_y: accepted, not needed (keys) */
int caps_sa_hip_merge_u64(const char* T, uint64_t n, const uint64_t* X, uint64_t len_x, const uint64_t* Y,
                          uint64_t len_y, const uint64_t* LCP_x, const uint64_t* LCP_y, uint64_t* Z,
                          uint64_t* LCP_z, int device);

/* upper_bound (src/Suffix_Array.cpp:252-297) without the 65,536-char cutoff: for each
 * pivot suffix position, the number of elements of the sorted list X that are <= it. */
int caps_sa_hip_upper_bound_u32(const char* T, uint64_t n, const uint32_t* X, uint64_t cnt,
                                const uint32_t* pivots, uint64_t n_pivots, uint32_t* out, int device);
int caps_sa_hip_upper_bound_u64(const char* T, uint64_t n, const uint64_t* X, uint64_t cnt,
                                const uint64_t* pivots, uint64_t n_pivots, uint64_t* out, int device);

/* LCP<8> (include/Suffix_Array.hpp:195-241): out[i] = lcp(T[a[i]..), T[b[i]..)). */
int caps_sa_hip_lcp_u32(const char* T, uint64_t n, const uint32_t* a, const uint32_t* b, uint64_t cnt,
                        uint32_t* out, int device);
int caps_sa_hip_lcp_u64(const char* T, uint64_t n, const uint64_t* a, const uint64_t* b, uint64_t cnt,
                        uint64_t* out, int device);


/* ---- multi-GPU: one shard per process/GPU ----------------------------
 * The reference has no distributed mode; these entry points cut construct() at its one
 * exchange step (partition_sub_subarrays, src/Suffix_Array.cpp:300-368) so that a host
 * driver can run the collectives between them (caps_sa_dist.py: torch.distributed / RCCL):
 *
 *   shard_create -> shard_phase1 -> [all_gather samples] -> shard_pivots
 *   -> [all_gather partition sizes] -> shard_collate -> [all_to_all_v of (key, sa)]
 *   -> shard_phase2 -> [all_gather last SA] -> shard_fix_first_lcp
 *
 * The text is replicated (dT: n bytes on this rank's device).  Rank r sorts subarrays
 * [r*p/world, (r+1)*p/world) and ends up owning a contiguous slice of the global SA/LCP.
 * All d_* pointers are device pointers on the rank's device; idx_bytes is 4 or 8. */
typedef struct caps_sa_shard caps_sa_shard;
typedef struct caps_sa_shard_info {
    uint64_t n;
    uint32_t p, ppp, rank, world;
    uint32_t g0, g1;               /* subarrays sorted by this rank */
    uint32_t bits_per_char, idx_bytes;
    uint32_t part_lo, part_hi;     /* partitions owned after shard_collate */
    uint64_t local_elems;          /* elements of this rank's subarrays = capacity of the send buffers */
    uint64_t m_local, m_total;     /* samples of this rank / of all ranks */
    uint64_t recv_total;           /* elements of the owned partitions (after shard_collate) */
    uint64_t slice_off;            /* position of the rank's slice in the global SA/LCP */
    uint64_t capacity;             /* most elements this shard can receive (size of recv / SA / LCP buffers) */
    double ms_phase1, ms_pivots, ms_collate, ms_phase2;
    /* direct path (shard_scatter / shard_plan / shard_sort) */
    uint32_t direct_fallback;      /* CAPS_SA_FB_NONE: the shape allows the direct path; else why not */
    uint32_t direct_groups, direct_sub, n_streams;   /* groups, sub-streams per group, streams = groups x sub-streams */
    uint64_t stream_cap;           /* elements a stream region holds */
    uint64_t send_capacity;        /* elements the send buffers must hold (either path) */
    double ms_scatter, ms_sort;
    /* kernel families of the last direct build on this rank (HIP events on its stream) */
    double ms_level_a, ms_level_b, ms_tile_sort, ms_merge_passes;
    uint64_t level_a_elems;        /* text positions this rank distributed */
    uint32_t slot_splits, slot_splits_redone;
    uint32_t key_bytes;            /* bytes per key in the send / receive buffers of the last shard_scatter: 8, or 4 (32-bit keys,
                                      csrc/text.h key32_of: what travels when world > 1 on 2-bit texts) */
    uint32_t exchange;             /* 1: the streams shard_scatter wrote must be exchanged (all-to-all by the counts of shard_plan)
                                      before shard_sort; 0 (the default for the direct path): nothing travels -- every rank
                                      scattered the whole text and kept its own groups; shard_sort reads the send buffers */
    uint32_t direct_quantile;      /* 1: the last shard_scatter chose quantile buckets for level B (skewed keys, frequent keys, long
                                      runs; csrc/pipeline.h Builder::run_direct) -- no-exchange mode only */
    uint32_t run_buckets;          /* letter-run buckets of the last shard_sort (see caps_sa_stats.run_buckets) */
    uint64_t tie_groups_deferred;  /* groups of equal keys the last shard_sort re-keyed instead of comparing (see caps_sa_stats; no-exchange
                                      mode, 64-bit keys) */
    uint32_t tie_levels, reserved_;
} caps_sa_shard_info;

int caps_sa_hip_shard_create(const void* dT, uint64_t n, uint64_t subproblem_count, int idx_bytes, int rank, int world,
                             void* hip_stream, caps_sa_shard** out);
void caps_sa_hip_shard_destroy(caps_sa_shard* s);
int caps_sa_hip_shard_info(const caps_sa_shard* s, caps_sa_shard_info* info);
/* d_sample_keys: u64[m_local], d_sample_sa: idx[m_local] */
int caps_sa_hip_shard_phase1(caps_sa_shard* s, void* d_sample_keys, void* d_sample_sa);
/* d_all_keys: u64[m_total], d_all_sa: idx[m_total] (any order); d_local_sizes: u64[p] out */
int caps_sa_hip_shard_pivots(caps_sa_shard* s, const void* d_all_keys, const void* d_all_sa, void* d_local_sizes);
/* all_sizes: HOST u64[world][p]; d_send_*: [local_elems]; send_counts/recv_counts: HOST u64[world] out */
int caps_sa_hip_shard_collate(caps_sa_shard* s, const uint64_t* all_sizes, void* d_send_keys, void* d_send_sa,
                              uint64_t* send_counts, uint64_t* recv_counts);
/* d_recv_*: [recv_total], source-rank-major; dSA/dLCP: idx[recv_total] out */
int caps_sa_hip_shard_phase2(caps_sa_shard* s, const void* d_recv_keys, const void* d_recv_sa, void* dSA, void* dLCP);
/*
 * Direct path of a sharded build (what Builder::run_direct does on one GPU; no sort_subarrays, no locate_pivots):
 *
 *   shard_create -> shard_scatter -> [all_gather of the reports] -> shard_plan
 *   -> [only if shard_info.exchange: all-to-all of the (key, sa) blocks] -> shard_sort -> [all_gather last SA] -> shard_fix_first_lcp
 *
 * shard_scatter: packs the text, derives the pivots (every rank the same ones, from the same samples of the text) and
 * distributes suffixes into stream regions of d_send_keys (u64[send_capacity]) / d_send_sa (idx[send_capacity]).
 *   Default (shard_info.exchange = 0): the suffixes of the WHOLE text that belong to the groups this rank owns -- the text is
 *   replicated, so no element has to cross a link; shard_plan then returns all-zero counts and shard_sort reads the send
 *   buffers (d_recv_* = d_send_*).
 *   CAPS_SA_SHARD_EXCHANGE=1 (exchange = 1): the suffixes of every world-th tile of the text, all groups; the block for rank d
 *   is contiguous, shard_plan returns the elements to send to / receive from every rank (gaps of the regions included) and
 *   shard_sort takes the received blocks in rank order.
 * d_report: u64[n_streams + 2], this rank's stream sizes and flags.  shard_plan: all_reports = HOST u64[world][n_streams + 2];
 * returns 0, or a positive CAPS_SA_FB_* code -- the same on every rank -- when the text cannot be split by keys alone: the
 * ranks then run the samplesort sequence above (shard_phase1 ...).  shard_sort: dSA / dLCP: idx[capacity] out, recv_total
 * entries valid.  When d_recv_* ARE the send buffers of the last shard_scatter (exchange = 0), shard_sort may rewrite them: large
 * groups of equal keys are then re-keyed instead of compared, and if that refinement does not fit its work memory the shard
 * scatters into the same buffers once more and sorts with every tie compared.  Buffers of the caller's own are read only.
 */
int caps_sa_hip_shard_scatter(caps_sa_shard* s, void* d_send_keys, void* d_send_sa, void* d_report);
int caps_sa_hip_shard_plan(caps_sa_shard* s, const uint64_t* all_reports, uint64_t* send_counts, uint64_t* recv_counts);
int caps_sa_hip_shard_sort(caps_sa_shard* s, const void* d_recv_keys, const void* d_recv_sa, void* dSA, void* dLCP);
/* Key width (exchange = 1 only): shard_scatter writes 32-bit keys (shard_info.key_bytes = 4: d_send_keys / d_recv_keys are then u32
 * arrays) when the world has more than one rank and the text packs to 2 bits -- a third fewer bytes cross xGMI.  shard_sort then returns
 * CAPS_SA_FB_KEY32 (> 0) when a bucket slot overflowed (skewed keys the pivots did not reveal); the ranks take the maximum
 * of their codes, and if it is not 0 every rank calls shard_set_key_bits(s, 64) and repeats scatter / exchange / sort. */
int caps_sa_hip_shard_set_key_bits(caps_sa_shard* s, int bits);
/* Differential tests: copies this rank's sorted subarrays after shard_phase1 into d_keys_out (u64[count]) / d_sa_out
 * (idx[count]) (either may be NULL: sizes only); subarray g of the rank = entries [g * subarray_len, (g + 1) * subarray_len),
 * the last one to the end. */
int caps_sa_hip_shard_phase1_arrays(caps_sa_shard* s, void* d_keys_out, void* d_sa_out, uint64_t* count, uint64_t* subarray_len);
/* last SA value of the slice (UINT64_MAX when the slice is empty) */
int caps_sa_hip_shard_last_sa(caps_sa_shard* s, uint64_t* last_sa);
/* prev_sa: last SA value of the nearest non-empty lower rank (UINT64_MAX: none) */
int caps_sa_hip_shard_fix_first_lcp(caps_sa_shard* s, uint64_t prev_sa, void* dLCP);

#ifdef __cplusplus
}
#endif
#endif /* CAPS_SA_HIP_H */
