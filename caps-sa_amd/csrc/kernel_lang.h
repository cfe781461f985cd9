// caps-sa_amd/csrc/kernel_lang.h
//
// Thin kernel-language layer.  Every kernel in kernels.h is written once, as a
// sequence of block-uniform code and thread-parallel *phases*:
//
//     PAR(tid) { ...per-thread work, no barrier inside... }
//     SYNC();
//
// * Product build (hipcc, gfx950): PAR runs its body once with tid = threadIdx.x,
//   SYNC is __syncthreads(), SHARED_ARRAY is LDS, TL_* are registers.
// * CAPS_EMUL build (g++, tests/emul only): the same source becomes a host function;
//   PAR is a loop over the block's threads, SYNC is a no-op (phases are sequential).
//   This exists because the development container has no GPU: kernel *logic* (index
//   arithmetic, merge-path splits, ragged tiles) is debugged there.  It is test
//   infrastructure: the shipped library never contains or dispatches to it, and the
//   `-m gpu` parity tests never use it.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef CAPS_EMUL
// ------------------------------------------------------------------ host emulation
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#define HD inline
#define DEV_INLINE inline
#define HD_NOINLINE
#define GLOBAL_FN static void
#define LAUNCH_BOUNDS(n)
#define LAUNCH_BOUNDS2(n, w)
namespace caps { struct EmulCtx { uint32_t block_idx, grid_dim, block_dim; }; }
#define KCTX const caps::EmulCtx& kctx_,
#define KCTX_PASS kctx_,
#define K_BLOCK_IDX (kctx_.block_idx)
#define STREAM_LOAD(p) (*(p))                   /* a load of data read once (GPU: nontemporal, see below) */
#define STREAM_STORE(p, v) (*(p) = (v))
#define STREAM_STORE2(p, v) (*(p) = (v))
#define K_GRID_DIM (kctx_.grid_dim)
#define K_BLOCK_DIM (kctx_.block_dim)
// CAPS_EMUL_RACE (tests/emul/race_rt.h): the barrier-race detector -- every PAR iteration announces its thread, every barrier
// starts a new epoch, LDS arrays register their storage, atomics mark themselves.  Without the switch all of it expands to nothing.
#ifdef CAPS_EMUL_RACE
#include "race_rt.h"
#define CAPS_RACE_IN(t) caps_race::enter_thread((t), __LINE__)
#define CAPS_RACE_OUT() caps_race::leave_region()
#define CAPS_RACE_ARRAY(name, bytes) ; caps_race::ArrayGuard name##_rg_(name, (bytes), #name)
#define CAPS_RACE_ATOMIC caps_race::AtomicScope caps_as_
#define CAPS_RACE_BARRIER() caps_race::barrier()
#else
#define CAPS_RACE_IN(t) true
#define CAPS_RACE_OUT() false
#define CAPS_RACE_ARRAY(name, bytes)
#define CAPS_RACE_ATOMIC ((void)0)
#define CAPS_RACE_BARRIER() ((void)0)
#endif
#define CAPS_PAR_COND(tid, more) ((more) ? CAPS_RACE_IN(tid) : CAPS_RACE_OUT())
#if defined(CAPS_EMUL_SCATTER)   /* ... or in a scattered order: start anywhere, step by an odd stride (block sizes are powers of two) */
static inline uint32_t caps_emul_phase_start() { static uint32_t c = 12345u; c = c * 1664525u + 1013904223u; return c >> 8; }
#define PAR(tid) for (uint32_t tid##_i_ = 0, tid##_s_ = (kctx_.block_dim / 2u + 1u) | 1u, tid = caps_emul_phase_start() % kctx_.block_dim;                       CAPS_PAR_COND(tid, tid##_i_ < kctx_.block_dim); ++tid##_i_, tid = (tid + tid##_s_) % kctx_.block_dim)
#elif defined(CAPS_EMUL_REVERSE)   /* the threads of every phase in descending order: a phase that depends on the order of its threads
                              (a missing barrier between a write and a read of two threads) gives a different result than with
                              the ascending build, or a wrong one (tests/test_emul_pipeline.py, reversed-order cases) */
#define PAR(tid) for (uint32_t tid##_i_ = 0, tid = kctx_.block_dim - 1; CAPS_PAR_COND(tid, tid##_i_ < kctx_.block_dim); ++tid##_i_, --tid)
#else
#define PAR(tid) for (uint32_t tid = 0; CAPS_PAR_COND(tid, tid < kctx_.block_dim); ++tid)
#endif
// PAR_FRESH_SET / PAR_SAME (see the product side below): a region that shares the thread index of a group must come behind the
// region that took it -- here a flag says so, and a PAR_SAME that runs before any PAR_FRESH_SET aborts (on the GPU it would run all
// threads as thread 0).  PAR_TID_RESET: at the head of every tile of a kernel that loops over tiles (the flag is per tile).
static inline bool caps_par_same_ok_(bool set, const char* file, int line)
{
    if (!set) { std::fprintf(stderr, "%s:%d: PAR_SAME before its PAR_FRESH_SET\n", file, line); std::abort(); }
    return true;
}
#define PAR_TID_DECL bool caps_par_set_ = false
#define PAR_TID_RESET (caps_par_set_ = false)
// (if (!cond) {} else for (...): an `else` behind the use site cannot pair with this `if`)
#define PAR_FRESH_SET(tid) if (!(caps_par_set_ = true)) {} else PAR(tid)
#define PAR_SAME_G(g, tid) if (!caps_par_same_ok_(caps_par_set_, __FILE__, __LINE__)) {} else PAR(tid)
#define SYNC() CAPS_RACE_BARRIER()
#define SYNC_LDS() CAPS_RACE_BARRIER()
// CAPS_EMUL_POISON: LDS arrays and per-thread registers start as 0xA5 bytes instead of zeros (on the GPU they start as whatever
// the last workgroup left: a kernel that reads what it has not written must not pass because the emulation hands it zeros)
#ifdef CAPS_EMUL_POISON
#define CAPS_EMUL_FILL(vec) std::memset((void*)(vec).data(), 0xA5, (vec).size() * sizeof((vec)[0]))
#else
#define CAPS_EMUL_FILL(vec) ((void)0)
#endif
#define SHARED_ARRAY(type, name, count) std::vector<type> name##_vec_(count); CAPS_EMUL_FILL(name##_vec_); type* name = name##_vec_.data() CAPS_RACE_ARRAY(name, (size_t)(count) * sizeof(type))
#define TL_DECL(type, name, cnt) std::vector<type> name##_tl_((size_t)kctx_.block_dim * (cnt)); CAPS_EMUL_FILL(name##_tl_); \
    type* const name##_tlp_ = name##_tl_.data(); const uint32_t name##_tlc_ = (cnt)
#define TL(name, tid, k) name##_tlp_[(size_t)(tid) * name##_tlc_ + (k)]
#define UNROLL
template <typename T> static inline void caps_emul_or(T* p, T v) { CAPS_RACE_ATOMIC; *p |= v; }
template <typename T> static inline void caps_emul_min(T* p, T v) { CAPS_RACE_ATOMIC; if (v < *p) *p = v; }
template <typename T> static inline void caps_emul_max(T* p, T v) { CAPS_RACE_ATOMIC; if (v > *p) *p = v; }
template <typename T> static inline void caps_emul_add(T* p, T v) { CAPS_RACE_ATOMIC; *p += v; }
// a load / store of a word that other threads update atomically in the same phase, BY DESIGN (any interleaving gives a valid
// state: the use sites say why): plain accesses on the GPU, marked for the race detector here
template <typename T> static inline T caps_emul_load(const T* p) { CAPS_RACE_ATOMIC; return *p; }
template <typename T> static inline void caps_emul_store(T* p, T v) { CAPS_RACE_ATOMIC; *p = v; }
#define RACY_LOAD_U32(ptr) caps_emul_load<uint32_t>((ptr))
#define RACY_STORE_U32(ptr, v) caps_emul_store<uint32_t>((ptr), (v))
#define RACY_STORE_U16(ptr, v) caps_emul_store<uint16_t>((ptr), (v))
#define ATOMIC_OR_U32(ptr, v) caps_emul_or<uint32_t>((ptr), (v))
#define ATOMIC_MIN_U32(ptr, v) caps_emul_min<uint32_t>((ptr), (v))
#define ATOMIC_MAX_U32(ptr, v) caps_emul_max<uint32_t>((ptr), (v))
#define ATOMIC_ADD_U64(ptr, v) caps_emul_add<uint64_t>((ptr), (v))
#define ATOMIC_ADD_LDS_U64(ptr, v) caps_emul_add<uint64_t>((ptr), (v))
#define ATOMIC_MAX_U64(ptr, v) caps_emul_max<uint64_t>((ptr), (v))
#define ATOMIC_MAX_LDS_U64(ptr, v) caps_emul_max<uint64_t>((ptr), (v))
static inline uint32_t caps_fetch_add_u32(uint32_t* p, uint32_t v) { CAPS_RACE_ATOMIC; const uint32_t o = *p; *p = o + v; return o; }
static inline uint64_t caps_fetch_add_u64(uint64_t* p, uint64_t v) { CAPS_RACE_ATOMIC; const uint64_t o = *p; *p = o + v; return o; }
#define FETCH_ADD_U32(ptr, v) caps_fetch_add_u32((ptr), (v))      /* returns the old value */
static inline void caps_emul_minmax(uint64_t* pmin, uint64_t* pmax, uint64_t mn, uint64_t mx) { caps_emul_min<uint64_t>(pmin, mn); caps_emul_max<uint64_t>(pmax, mx); }
#define BLOCK_MINMAX_U64(pmin, pmax, mn, mx) caps_emul_minmax((pmin), (pmax), (mn), (mx))
#define FETCH_ADD_U64(ptr, v) caps_fetch_add_u64((ptr), (v))
static inline uint32_t caps_fetch_add(uint32_t* p, uint32_t v) { return caps_fetch_add_u32(p, v); }
static inline uint64_t caps_fetch_add(uint64_t* p, uint64_t v) { return caps_fetch_add_u64(p, v); }
static inline uint64_t caps_umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
static inline uint32_t caps_umul32hi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline int caps_clz64(uint64_t x) { return __builtin_clzll(x); }
static inline uint32_t caps_bswap32(uint32_t x) { return __builtin_bswap32(x); }
#else
// ------------------------------------------------------------------ gfx950 (product)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#define DEV_INLINE __device__ __forceinline__
#define HD_NOINLINE __host__ __device__ __attribute__((noinline))
#define GLOBAL_FN __global__ void
#define LAUNCH_BOUNDS(n) __launch_bounds__(n)
#define LAUNCH_BOUNDS2(n, w) __launch_bounds__(n, w)   /* w = min waves per SIMD = blocks/CU * n / 256 */
#define KCTX
#define KCTX_PASS
#define K_BLOCK_IDX (blockIdx.x)
// Cache hints for data that passes once (round 5; measured at C3 on one box, 6 builds each, twice):
//   STREAM_LOAD   the streams between the passes, read by the scatters and the tile sorts: nontemporal -- level B 16.85 -> 16.62 ms;
//   STREAM_STORE  the results (SA, LCP), written once and never read again by the build: nontemporal -- tile sort 14.63 -> 14.37 ms;
//   STREAM_STORE2 the scatters' own stores must NOT be: they are short runs that complete each other's cache lines in L2, and with
//                 nontemporal stores level A took 35 ms instead of 13.6 and level B 54 instead of 16.6 (-DCAPS_STREAM_STORES2 repeats it).
// -DCAPS_NO_STREAM_HINTS: plain loads and stores everywhere (the builds before this).
#ifndef CAPS_NO_STREAM_HINTS
#define STREAM_LOAD(p) __builtin_nontemporal_load(p)
#define STREAM_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define STREAM_LOAD(p) (*(p))
#define STREAM_STORE(p, v) (*(p) = (v))
#endif
#ifdef CAPS_STREAM_STORES2
#define STREAM_STORE2(p, v) __builtin_nontemporal_store((v), (p))
#else
#define STREAM_STORE2(p, v) (*(p) = (v))
#endif
#define K_GRID_DIM (gridDim.x)
#define K_BLOCK_DIM (blockDim.x)
#define PAR(tid) for (uint32_t tid = threadIdx.x, par_once_ = 1; par_once_; par_once_ = 0)
// PAR_FRESH: the same, but the compiler is kept from knowing that this region's thread index is the one of the regions
// before it.  What a kernel with many phases derives from the index (addresses, element numbers) is otherwise computed
// once, held in registers across all phases (or hoisted out of a loop over tiles) and, at the 64-register budget of two
// 1024-thread workgroups per CU, spilled to scratch memory; recomputing it per region is a few instructions.  Pays in the
// long kernels (tile_sort_eq_kernel, tile_sort_general_kernel: kernels.h switches PAR over for them); the short ones
// (scatters, tile_sort_kernel) lose 2-7 % to it.
static __device__ __forceinline__ uint32_t caps_tid_fresh() { uint32_t t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
#define PAR_FRESH(tid) for (uint32_t tid = caps_tid_fresh(), par_once_ = 1; par_once_; par_once_ = 0)
// ... and a middle way (-DCAPS_PAR_GROUPS; off in the product build, see DESIGN "the reverted fault"): PAR_FRESH_SET starts a group
// of regions, PAR_SAME regions behind it share its index (what they derive from it may stay in registers across the group, not
// across the kernel).  PAR_TID_DECL: once at the head of the kernel.  Without the switch both are PAR.
// CAPS_PAR_GROUPS is a mask of the groups that share (1: equalisation, 2: tie rounds, 4: pick-up + placement); PAR_SAME_G(g, tid).
#ifdef CAPS_PAR_GROUPS
#define PAR_TID_DECL uint32_t caps_par_tid_ = 0
#define PAR_TID_RESET ((void)0)
#define PAR_FRESH_SET(tid) for (uint32_t tid = (caps_par_tid_ = caps_tid_fresh()), par_once_ = 1; par_once_; par_once_ = 0)
#define PAR_SAME_ON_(tid) for (uint32_t tid = caps_par_tid_, par_once_ = 1; par_once_; par_once_ = 0)
#if (CAPS_PAR_GROUPS) & 1
#define PAR_SAME_1(tid) PAR_SAME_ON_(tid)
#else
#define PAR_SAME_1(tid) PAR(tid)
#endif
#if (CAPS_PAR_GROUPS) & (1 | 8)
#define PAR_SAME_1a(tid) PAR_SAME_ON_(tid)
#else
#define PAR_SAME_1a(tid) PAR(tid)
#endif
#if (CAPS_PAR_GROUPS) & (1 | 16)
#define PAR_SAME_1b(tid) PAR_SAME_ON_(tid)
#else
#define PAR_SAME_1b(tid) PAR(tid)
#endif
#if (CAPS_PAR_GROUPS) & (1 | 32)
#define PAR_SAME_1c(tid) PAR_SAME_ON_(tid)
#else
#define PAR_SAME_1c(tid) PAR(tid)
#endif
#if (CAPS_PAR_GROUPS) & (1 | 64)
#define PAR_SAME_1d(tid) PAR_SAME_ON_(tid)
#else
#define PAR_SAME_1d(tid) PAR(tid)
#endif
#if (CAPS_PAR_GROUPS) & (1 | 128)
#define PAR_SAME_1e(tid) PAR_SAME_ON_(tid)
#else
#define PAR_SAME_1e(tid) PAR(tid)
#endif
#if (CAPS_PAR_GROUPS) & (1 | 256)
#define PAR_SAME_1f(tid) PAR_SAME_ON_(tid)
#else
#define PAR_SAME_1f(tid) PAR(tid)
#endif
#if (CAPS_PAR_GROUPS) & 2
#define PAR_SAME_2(tid) PAR_SAME_ON_(tid)
#else
#define PAR_SAME_2(tid) PAR(tid)
#endif
#if (CAPS_PAR_GROUPS) & 4
#define PAR_SAME_4(tid) PAR_SAME_ON_(tid)
#else
#define PAR_SAME_4(tid) PAR(tid)
#endif
#else
#define PAR_TID_DECL ((void)0)
#define PAR_TID_RESET ((void)0)
#define PAR_FRESH_SET(tid) PAR(tid)
#define PAR_SAME_1(tid) PAR(tid)
#define PAR_SAME_1a(tid) PAR(tid)
#define PAR_SAME_1b(tid) PAR(tid)
#define PAR_SAME_1c(tid) PAR(tid)
#define PAR_SAME_1d(tid) PAR(tid)
#define PAR_SAME_1e(tid) PAR(tid)
#define PAR_SAME_1f(tid) PAR(tid)
#define PAR_SAME_2(tid) PAR(tid)
#define PAR_SAME_4(tid) PAR(tid)
#endif
#define PAR_SAME_G(g, tid) PAR_SAME_##g(tid)
// (bits 8 .. 256 of CAPS_PAR_GROUPS switch the six regions of group 1 one by one: PAR_SAME_G(1a .. 1f, tid))
#define SYNC() __syncthreads()
// Barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also drains the wave's
// outstanding global stores and returning atomics (vmcnt); where nothing that went to global memory is handed to another
// thread of the workgroup, this one lets them stay in flight across the barrier (the cursor bumps of the scatter kernels
// return while the tile is scanned and re-ordered in LDS).
static __device__ __forceinline__ void caps_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#ifdef CAPS_FULL_BARRIERS      /* measurement / debugging: every barrier is __syncthreads() */
#define SYNC_LDS() __syncthreads()
#else
#define SYNC_LDS() caps_lds_barrier()
#endif
#define SHARED_ARRAY(type, name, count) __shared__ type name[count]
#define TL_DECL(type, name, cnt) type name##_reg_[cnt]
#define TL(name, tid, k) name##_reg_[k]
#define UNROLL _Pragma("unroll")
#define RACY_LOAD_U32(ptr) (*(ptr))                 /* see the emulation side: racing with atomics by design */
#define RACY_STORE_U32(ptr, v) (*(ptr) = (v))
#define RACY_STORE_U16(ptr, v) (*(ptr) = (v))
#define ATOMIC_OR_U32(ptr, v) atomicOr((ptr), (v))
#define ATOMIC_MIN_U32(ptr, v) atomicMin((ptr), (v))
#define ATOMIC_MAX_U32(ptr, v) atomicMax((ptr), (v))
#define ATOMIC_ADD_U64(ptr, v) atomicAdd((unsigned long long*)(ptr), (unsigned long long)(v))
#define ATOMIC_ADD_LDS_U64(ptr, v) atomicAdd((unsigned long long*)(ptr), (unsigned long long)(v))
#define ATOMIC_MAX_U64(ptr, v) atomicMax((unsigned long long*)(ptr), (unsigned long long)(v))
#define ATOMIC_MAX_LDS_U64(ptr, v) atomicMax((unsigned long long*)(ptr), (unsigned long long)(v))
#define FETCH_ADD_U32(ptr, v) atomicAdd((ptr), (v))                 /* returns the old value (LDS or global) */
// min/max of a per-thread value over the workgroup into two LDS words: wave64 butterfly with
// shuffles, then one LDS atomic per wave.
static __device__ __forceinline__ void caps_block_minmax_u64(uint64_t* pmin, uint64_t* pmax, uint64_t mn, uint64_t mx)
{
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin((unsigned long long*)pmin, (unsigned long long)mn);
        atomicMax((unsigned long long*)pmax, (unsigned long long)mx);
    }
}
#define BLOCK_MINMAX_U64(pmin, pmax, mn, mx) caps_block_minmax_u64((pmin), (pmax), (mn), (mx))
#define FETCH_ADD_U64(ptr, v) ((uint64_t)atomicAdd((unsigned long long*)(ptr), (unsigned long long)(v)))
static __device__ __forceinline__ uint32_t caps_fetch_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
static __device__ __forceinline__ uint64_t caps_fetch_add(uint64_t* p, uint64_t v) { return (uint64_t)atomicAdd((unsigned long long*)p, (unsigned long long)v); }
static __device__ __forceinline__ uint64_t caps_umul64hi(uint64_t a, uint64_t b) { return __umul64hi(a, b); }
static __device__ __forceinline__ uint32_t caps_umul32hi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
static __host__ __device__ __forceinline__ int caps_clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}
static __host__ __device__ __forceinline__ uint32_t caps_bswap32(uint32_t x) { return __builtin_bswap32(x); }
#endif
