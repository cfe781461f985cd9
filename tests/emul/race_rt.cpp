// tests/emul/race_rt.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// Runtime of the emulation's barrier-race detector (race_rt.h): the __tsan_* hooks g++ -fsanitize=thread calls before every
// memory access of the instrumented emulation library.  THIS file is compiled WITHOUT the sanitizer flag.
#include "race_rt.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <tuple>
#include <vector>

namespace caps_race {
static uint32_t cur_tid = TID_ALL;       // thread whose code runs now (TID_ALL outside PAR regions)
static uint32_t cur_line = 0;            // source line of the PAR region
static uint32_t atomic_depth = 0;        // > 0: inside an atomic read-modify-write
static uint64_t epoch = 1;               // the stretch between two barriers
void flush_pending();

namespace {
constexpr uint32_t TID_MULTI = 0xFFFFFFFEu;     // several threads have read the byte in this epoch
struct Cell {
    uint64_t w_epoch = 0, r_epoch = 0;
    uint32_t w_tid = 0, r_tid = 0, w_line = 0, r_line = 0;
    uint8_t w_atomic = 0, r_atomic = 0;
};
struct Array {
    uintptr_t lo, hi;
    const char* name;
    Cell* cells;
};
std::vector<Array> arrays;
uintptr_t all_lo = ~(uintptr_t)0, all_hi = 0;
uint64_t n_races = 0;
std::set<std::tuple<std::string, int, uint32_t, uint32_t>> seen;

struct Pending {
    bool on = false;
    const Array* a = nullptr;
    uintptr_t addr = 0;
    uint32_t size = 0, prev_tid = 0, prev_line = 0, tid = 0, line = 0;
    unsigned char old[16];
} pending;

void bounds()
{
    all_lo = ~(uintptr_t)0;
    all_hi = 0;
    for (const Array& a : arrays) {
        if (a.lo < all_lo) all_lo = a.lo;
        if (a.hi > all_hi) all_hi = a.hi;
    }
}

const Array* find(uintptr_t p)
{
    if (p < all_lo || p >= all_hi) return nullptr;
    for (const Array& a : arrays)
        if (p >= a.lo && p < a.hi) return &a;
    return nullptr;
}

const char* tid_str(uint32_t t, char* buf)
{
    if (t == TID_ALL) return "all-threads";
    if (t == TID_MULTI) return "several-threads";
    std::snprintf(buf, 16, "%u", t);
    return buf;
}

void report(int kind, const Array& a, uintptr_t addr, uint32_t size, uint32_t tid1, uint32_t line1, uint32_t tid2, uint32_t line2)
{
    ++n_races;
    static const char* const kinds[] = {"read-after-write", "write-after-read", "write-write (different values)"};
    if (!seen.insert(std::make_tuple(std::string(a.name), kind, line1, line2)).second) return;
    if (seen.size() > 40) return;
    char b1[16], b2[16];
    std::fprintf(stderr,
                 "caps_race: %s on LDS array '%s' byte offset %zu (access of %u bytes): thread %s in the region at kernels line %u, "
                 "then thread %s in the region at line %u, no barrier in between (epoch %llu)\n",
                 kinds[kind], a.name, (size_t)(addr - a.lo), size, tid_str(tid1, b1), line1, tid_str(tid2, b2), line2,
                 (unsigned long long)epoch);
}
}  // namespace

void register_array(const void* base, size_t bytes, const char* name)
{
    flush_pending();
    Array a;
    a.lo = (uintptr_t)base;
    a.hi = a.lo + bytes;
    a.name = name;
    a.cells = new Cell[bytes ? bytes : 1];
    arrays.push_back(a);
    bounds();
}

void unregister_array(const void* base)
{
    flush_pending();
    for (size_t i = arrays.size(); i-- > 0;)
        if (arrays[i].lo == (uintptr_t)base) {
            delete[] arrays[i].cells;
            arrays.erase(arrays.begin() + (long)i);
            break;
        }
    bounds();
}

void flush_pending()
{
    if (!pending.on) return;
    pending.on = false;
    if (std::memcmp(pending.old, (const void*)pending.addr, pending.size) != 0)
        report(2, *pending.a, pending.addr, pending.size, pending.prev_tid, pending.prev_line, pending.tid, pending.line);
}

bool enter_thread(uint32_t tid, uint32_t line) { cur_tid = tid; cur_line = line; return true; }
bool leave_region() { cur_tid = TID_ALL; return false; }
void atomic_begin() { ++atomic_depth; }
void atomic_end() { flush_pending(); --atomic_depth; }
void barrier() { flush_pending(); ++epoch; cur_tid = TID_ALL; }
uint64_t races_found() { flush_pending(); return n_races; }
void reset_count() { flush_pending(); n_races = 0; seen.clear(); }

static inline void on_read(uintptr_t p, uint32_t size)
{
    const Array* a = find(p);
    if (!a) return;
    flush_pending();
    const bool at = atomic_depth != 0;
    bool told = false;
    for (uint32_t i = 0; i < size && p + i < a->hi; ++i) {
        Cell& c = a->cells[p + i - a->lo];
        if (!told && c.w_epoch == epoch && c.w_tid != cur_tid && c.w_tid != TID_ALL && !(at && c.w_atomic)) {
            report(0, *a, p, size, c.w_tid, c.w_line, cur_tid, cur_line);
            told = true;
        }
        if (c.r_epoch != epoch) {
            c.r_epoch = epoch;
            c.r_tid = cur_tid;
            c.r_line = cur_line;
            c.r_atomic = at;
        } else if (c.r_tid != cur_tid) {
            if (c.r_tid != TID_ALL) c.r_tid = TID_MULTI;         // (all-threads stays: it includes everybody)
            c.r_atomic = c.r_atomic && at;
            c.r_line = cur_line;
        }
    }
}

static inline void on_write(uintptr_t p, uint32_t size)
{
    const Array* a = find(p);
    if (!a) return;
    flush_pending();
    const bool at = atomic_depth != 0;
    bool told = false, waw = false;
    uint32_t ptid = 0, pline = 0;
    for (uint32_t i = 0; i < size && p + i < a->hi; ++i) {
        Cell& c = a->cells[p + i - a->lo];
        if (!told && c.r_epoch == epoch && c.r_tid != cur_tid && !(at && c.r_atomic)) {
            report(1, *a, p, size, c.r_tid, c.r_line, cur_tid, cur_line);
            told = true;
        }
        if (c.w_epoch == epoch && c.w_tid != cur_tid && !(at && c.w_atomic) && !waw) {
            waw = true;
            ptid = c.w_tid;
            pline = c.w_line;
        }
        c.w_epoch = epoch;
        c.w_tid = cur_tid;
        c.w_line = cur_line;
        c.w_atomic = at;
    }
    if (waw && size <= 16) {                       // different threads, one epoch: a race unless they store the same value
        pending.on = true;
        pending.a = a;
        pending.addr = p;
        pending.size = size;
        pending.prev_tid = ptid;
        pending.prev_line = pline;
        pending.tid = cur_tid;
        pending.line = cur_line;
        std::memcpy(pending.old, (const void*)p, size);
    }
}
}  // namespace caps_race

extern "C" {
void __tsan_init(void) {}
void __tsan_func_entry(void*) {}
void __tsan_func_exit(void) {}
void __tsan_vptr_update(void**, void*) {}
void __tsan_vptr_read(void**) {}
void __tsan_read1(void* p) { caps_race::on_read((uintptr_t)p, 1); }
void __tsan_read2(void* p) { caps_race::on_read((uintptr_t)p, 2); }
void __tsan_read4(void* p) { caps_race::on_read((uintptr_t)p, 4); }
void __tsan_read8(void* p) { caps_race::on_read((uintptr_t)p, 8); }
void __tsan_read16(void* p) { caps_race::on_read((uintptr_t)p, 16); }
void __tsan_write1(void* p) { caps_race::on_write((uintptr_t)p, 1); }
void __tsan_write2(void* p) { caps_race::on_write((uintptr_t)p, 2); }
void __tsan_write4(void* p) { caps_race::on_write((uintptr_t)p, 4); }
void __tsan_write8(void* p) { caps_race::on_write((uintptr_t)p, 8); }
void __tsan_write16(void* p) { caps_race::on_write((uintptr_t)p, 16); }
void __tsan_unaligned_read2(void* p) { caps_race::on_read((uintptr_t)p, 2); }
void __tsan_unaligned_read4(void* p) { caps_race::on_read((uintptr_t)p, 4); }
void __tsan_unaligned_read8(void* p) { caps_race::on_read((uintptr_t)p, 8); }
void __tsan_unaligned_read16(void* p) { caps_race::on_read((uintptr_t)p, 16); }
void __tsan_unaligned_write2(void* p) { caps_race::on_write((uintptr_t)p, 2); }
void __tsan_unaligned_write4(void* p) { caps_race::on_write((uintptr_t)p, 4); }
void __tsan_unaligned_write8(void* p) { caps_race::on_write((uintptr_t)p, 8); }
void __tsan_unaligned_write16(void* p) { caps_race::on_write((uintptr_t)p, 16); }
void __tsan_read_range(void* p, unsigned long n) { for (unsigned long i = 0; i < n; ++i) caps_race::on_read((uintptr_t)p + i, 1); }
void __tsan_write_range(void* p, unsigned long n) { for (unsigned long i = 0; i < n; ++i) caps_race::on_write((uintptr_t)p + i, 1); }
// atomics of the instrumented code itself (function-local static guards, std::atomic in host-side helpers): plain builtins -- they
// never touch LDS arrays
#define CAPS_TSAN_ATOMICS(BITS, T)                                                                                              \
    T __tsan_atomic##BITS##_load(const volatile T* a, int) { return __atomic_load_n(a, __ATOMIC_SEQ_CST); }                    \
    void __tsan_atomic##BITS##_store(volatile T* a, T v, int) { __atomic_store_n(a, v, __ATOMIC_SEQ_CST); }                    \
    T __tsan_atomic##BITS##_exchange(volatile T* a, T v, int) { return __atomic_exchange_n(a, v, __ATOMIC_SEQ_CST); }          \
    T __tsan_atomic##BITS##_fetch_add(volatile T* a, T v, int) { return __atomic_fetch_add(a, v, __ATOMIC_SEQ_CST); }          \
    T __tsan_atomic##BITS##_fetch_sub(volatile T* a, T v, int) { return __atomic_fetch_sub(a, v, __ATOMIC_SEQ_CST); }          \
    T __tsan_atomic##BITS##_fetch_and(volatile T* a, T v, int) { return __atomic_fetch_and(a, v, __ATOMIC_SEQ_CST); }          \
    T __tsan_atomic##BITS##_fetch_or(volatile T* a, T v, int) { return __atomic_fetch_or(a, v, __ATOMIC_SEQ_CST); }            \
    T __tsan_atomic##BITS##_fetch_xor(volatile T* a, T v, int) { return __atomic_fetch_xor(a, v, __ATOMIC_SEQ_CST); }          \
    int __tsan_atomic##BITS##_compare_exchange_strong(volatile T* a, T* c, T v, int, int)                                        \
    { return __atomic_compare_exchange_n(a, c, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); }                                 \
    int __tsan_atomic##BITS##_compare_exchange_weak(volatile T* a, T* c, T v, int, int)                                          \
    { return __atomic_compare_exchange_n(a, c, v, true, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); }
CAPS_TSAN_ATOMICS(8, unsigned char)
CAPS_TSAN_ATOMICS(16, unsigned short)
CAPS_TSAN_ATOMICS(32, unsigned int)
CAPS_TSAN_ATOMICS(64, unsigned long long)
void __tsan_atomic_thread_fence(int) { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
void __tsan_atomic_signal_fence(int) { __atomic_signal_fence(__ATOMIC_SEQ_CST); }
// read by the tests
unsigned long long caps_sa_emul_races_found(void) { return caps_race::races_found(); }
void caps_sa_emul_races_reset(void) { caps_race::reset_count(); }
}
