// tests/emul/race_rt.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Barrier-race detector for the host emulation of the kernels (kernel_lang.h, -DCAPS_EMUL_RACE).
//
// The plain emulation runs the phases of a kernel one after the other and the threads of a phase in a loop, so a MISSING
// BARRIER between two phases is invisible to it: on the GPU the threads of a workgroup only meet at s_barrier, and what one
// thread leaves in LDS for another is safe only across one.  This build finds such hand-offs.  The emulation library is
// compiled with g++ -fsanitize=thread, which makes the compiler call __tsan_read<N> / __tsan_write<N> before every memory
// access; instead of libtsan this file supplies those hooks.  They look only at the addresses of the kernels' LDS arrays
// (SHARED_ARRAY registers its storage) and keep, per byte, who wrote and who read it last and in which EPOCH -- the stretch
// between two barriers (SYNC / SYNC_LDS bump the epoch).  Inside one epoch of one workgroup:
//   * a thread reads a byte another thread has written            -> read-after-write race
//   * a thread writes a byte another thread has read              -> write-after-read race
//   * two threads write one byte with DIFFERENT values            -> write-write race (equal values: the "every writer stores
//                                                                     1" flags, benign on the GPU too)
// Atomic read-modify-writes (the FETCH_ADD / ATOMIC_* macros) do not race with each other, but do with plain accesses of
// other threads.  Code outside a PAR region is executed by every thread: its reads count as reads by "all threads".
// A report names the array, the element, both threads and the source lines of the two PAR regions.
#pragma once
#include <cstdint>
#include <cstddef>

namespace caps_race {
constexpr uint32_t TID_ALL = 0xFFFFFFFFu;
// Everything the instrumented code tells the runtime goes through CALLS of functions defined in race_rt.cpp: g++ treats the
// __tsan_* hooks as builtins that touch no global, and plain stores to a "current thread" variable between two hooks were
// optimised away (found by tests/test_emul_race.py's self-test).
bool enter_thread(uint32_t tid, uint32_t line);   // a PAR iteration starts: this thread's code runs now; returns true
bool leave_region();                              // the PAR loop is over (code outside regions = all threads); returns false
void atomic_begin();                              // the accesses up to atomic_end() are one atomic read-modify-write
void atomic_end();
void barrier();                                   // a new epoch (SYNC / SYNC_LDS; every kernel block starts with one)
void register_array(const void* base, size_t bytes, const char* name);
void unregister_array(const void* base);
uint64_t races_found();
void reset_count();
struct ArrayGuard {
    const void* base;
    ArrayGuard(const void* b, size_t bytes, const char* name) : base(b) { register_array(b, bytes, name); }
    ~ArrayGuard() { unregister_array(base); }
};
struct AtomicScope {
    AtomicScope() { atomic_begin(); }
    ~AtomicScope() { atomic_end(); }
};
}  // namespace caps_race
