// caps-sa_amd/csrc/capi_impl.h
//
// Bodies of the C ABI declared in include/caps_sa_hip.h.  Included exactly once by
// caps_sa_hip.hip (product: symbols caps_sa_hip_*) and, for the host emulation used by
// CPU-side logic tests, by tests/emul/emul_lib.cpp (symbols caps_sa_emul_*).
#pragma once
#include <chrono>
#include <cstring>
#include <exception>
#include <functional>
#include <limits>
#include <memory>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pipeline.h"
#include "shard.h"

#ifndef CAPS_API
#error "define CAPS_API(name) before including capi_impl.h"
#endif

namespace caps {

inline std::string& last_error_ref()
{
    static thread_local std::string s;
    return s;
}

// Runs f(); maps exceptions to the ABI's error codes.
template <typename F> int guarded(F&& f)
{
    try {
        last_error_ref().clear();
        return f();
    } catch (const OomError& e) {
        last_error_ref() = e.what();
        return CAPS_SA_ENOMEM;
    } catch (const HipError& e) {
        last_error_ref() = e.what();
        return CAPS_SA_EHIP;
    } catch (const AlphabetError& e) {
        last_error_ref() = e.what();
        return CAPS_SA_EALPHABET;
    } catch (const std::bad_alloc&) {
        last_error_ref() = "host allocation failed";
        return CAPS_SA_ENOMEM;
    } catch (const std::exception& e) {
        last_error_ref() = e.what();
        return CAPS_SA_EINVAL;
    }
}

inline int fail(int code, const char* msg)
{
    last_error_ref() = msg;
    return code;
}

// RAII device allocations of one call.
struct DevAllocs {
    Backend& be;
    std::vector<void*> ptrs;
    explicit DevAllocs(Backend& b) : be(b) {}
    ~DevAllocs() { for (void* p : ptrs) be.free(p); }
    template <typename T> T* get(size_t count)
    {
        T* p = static_cast<T*>(be.alloc(count * sizeof(T)));
        ptrs.push_back(p);
        return p;
    }
};

template <typename idx_t> int check_common(const void* T, uint64_t n, uint64_t max_context)
{
    if (n && !T) return fail(CAPS_SA_EINVAL, "null text");
    if (n > (uint64_t)std::numeric_limits<idx_t>::max())
        return fail(CAPS_SA_EINVAL, "n does not fit the index type (use the _u64 entry point, src/main.cpp:76)");
    // bounded context (csrc/bounded.h): defined where the reference is -- n >= 32 and at least two subproblems
    if (max_context != 0 && max_context < n && n < 32)
        return fail(CAPS_SA_EUNSUPPORTED, "bounded max_context needs n >= 32 (the reference is undefined below that)");
    return CAPS_SA_OK;
}

int set_device(int device);   // defined by the including translation unit

// waves / sink: Builder::set_waves (the host-buffer entry point streams finished slices of the result out while the rest is sorted)
inline uint64_t wave_scratch_elems(uint64_t n, uint32_t waves) { return std::max<uint64_t>(4 * TILE_E, (n / waves) * 3 / 2 + 2 * TILE_E); }
template <typename idx_t> inline size_t wave_scratch_bytes(uint64_t elems)
{
    return ((elems * sizeof(uint64_t) + 255) & ~size_t(255)) + 2 * ((elems * sizeof(idx_t) + 255) & ~size_t(255)) + 256;
}
template <typename idx_t>
int build_device(const void* dT, uint64_t n, uint64_t p_arg, uint64_t max_context, void* dSA, void* dLCP, void* workspace,
                 uint64_t workspace_bytes, void* stream, caps_sa_stats* stats, uint32_t waves = 1, WaveSink* sink = nullptr,
                 bool* sink_served = nullptr, int arena_bits = 0)
{
    // arena_bits: the code width the workspace's text arena was sized for (0: told by the size of the workspace)
    if (int rc = check_common<idx_t>(dT, n, max_context)) return rc;
    if (n && (!dSA || !dLCP)) return fail(CAPS_SA_EINVAL, "null output");
    return guarded([&]() -> int {
        Backend be(static_cast<decltype(Backend::stream)>(stream));
        DevAllocs da(be);
        Plan<idx_t> need = make_plan<idx_t>(n, p_arg, nullptr);
        char* base = static_cast<char*>(workspace);
        int text_bits = arena_bits == 2 ? 2 : 8;
        if (!base) base = da.get<char>(need.bytes);
        else if (arena_bits == 2) {
            if (workspace_bytes < make_plan<idx_t>(n, p_arg, nullptr, 2).bytes) return fail(CAPS_SA_EINVAL, "workspace too small");
        } else if (workspace_bytes < need.bytes) {
            // a workspace of caps_sa_hip_workspace_bytes_ex(.., 2, ..): the text arena holds 2-bit codes only
            if (workspace_bytes < make_plan<idx_t>(n, p_arg, nullptr, 2).bytes) return fail(CAPS_SA_EINVAL, "workspace too small");
            text_bits = 2;
        }
        // carve from a 256-byte aligned base
        char* aligned = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(base) + 255) & ~uintptr_t(255));
        Plan<idx_t> pl = make_plan<idx_t>(n, p_arg, aligned, text_bits);
        Builder<idx_t> b(be, pl);
        ElemBuf<idx_t> scratch;
        uint64_t scratch_elems = 0;
        if (waves > 1 && sink) {                         // the waves' work arrays, behind the plan in the caller's workspace
            scratch_elems = wave_scratch_elems(n, waves);
            char* sb = aligned + ((pl.bytes + 255) & ~size_t(255));
            if (workspace && static_cast<char*>(workspace) + workspace_bytes >= sb + wave_scratch_bytes<idx_t>(scratch_elems)) {
                scratch.key = reinterpret_cast<uint64_t*>(sb);
                scratch.sa = reinterpret_cast<idx_t*>(sb + ((scratch_elems * sizeof(uint64_t) + 255) & ~size_t(255)));
                scratch.lcp = reinterpret_cast<idx_t*>(reinterpret_cast<char*>(scratch.sa) + ((scratch_elems * sizeof(idx_t) + 255) & ~size_t(255)));
                scratch.region_bytes = wave_scratch_bytes<idx_t>(scratch_elems);
            } else scratch_elems = 0;
        }
        b.set_waves(waves, sink, scratch, scratch_elems);
        if (max_context != 0 && max_context < n && pl.p < 2)
            return fail(CAPS_SA_EUNSUPPORTED, "bounded max_context needs at least two subproblems (the reference divides by zero there)");
        b.build(static_cast<const uint8_t*>(dT), static_cast<idx_t*>(dSA), static_cast<idx_t*>(dLCP), stats, max_context);
        if (sink_served) *sink_served = b.sink_served();
        return CAPS_SA_OK;
    });
}

// Results of the host-buffer entry point leave the device slice by slice: when a wave of groups has been sorted (its slice of
// SA / LCP is final), a second stream waits for that point of the build's stream and copies the slice to the caller's arrays
// while the next wave is sorted -- the D2H of 2 n indices (24 GB at C3: 420 ms at the link rate) then hides the rest of the
// build instead of following it.  (The upload of T cannot hide anything: sampling and level A need the whole text.)
//
// LCP as bytes (32-bit indices; lcp_narrow_kernel): on the link the LCP array is a quarter of its size -- 15 GB instead of 24 at C3.
// Per slice, on the copy stream: narrow on the device, copy the bytes into a page-locked staging array, copy the SA slice; a host
// thread waits for the bytes and widens them into the caller's array (a few worker threads, streaming stores) while the SA slice
// and the next slices are on the link.  Values of 255 and more travel as (position, value) pairs at the end; when there are more of
// them than the list holds (a^n: every value), the whole array is copied at full width as before.
inline uint32_t host_worker_threads()
{
    if (const char* e = std::getenv("CAPS_SA_HOST_THREADS")) return (uint32_t)std::max(1, std::atoi(e));
    uint32_t hw = std::thread::hardware_concurrency();
    // (a container's CPU quota is usually far below the node's thread count: the GPU boxes show 256 and grant 16)
    FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        long long q = 0, per = 0;
        char qs[32] = {0};
        if (std::fscanf(f, "%31s %lld", qs, &per) == 2 && std::strcmp(qs, "max") != 0 && per > 0) {
            q = std::atoll(qs);
            if (q > 0) hw = std::min<uint32_t>(hw ? hw : 1u, (uint32_t)std::max<long long>(1, (q + per - 1) / per));
        }
        std::fclose(f);
    }
    return std::max(1u, std::min(hw ? hw : 1u, 32u));
}

// out[i] = in[i] for i in [0, cnt): bytes to 32-bit values, with streaming stores where the compiler has them (the 12 GB written at
// C3 would otherwise be read first, line by line)
inline void widen_bytes(const uint8_t* in, uint32_t* out, uint64_t cnt)
{
    uint64_t i = 0;
#if defined(__clang__) && defined(__SSE2__)
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    for (; i < cnt && (reinterpret_cast<uintptr_t>(out + i) & 15u); ++i) out[i] = in[i];
    for (; i + 4 <= cnt; i += 4) {
        const v4u v = {in[i], in[i + 1], in[i + 2], in[i + 3]};
        __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(out + i));
    }
#endif
    for (; i < cnt; ++i) out[i] = in[i];
}

template <typename idx_t> struct HostCopySink : WaveSink {
    decltype(Backend::stream) copy_stream;
    idx_t *SA, *LCP;
    const idx_t *dSA, *dLCP;
    uint64_t copied = 0;
    // LCP as bytes (null: plain copies)
    uint8_t* d8 = nullptr;             // device: the bytes
    uint64_t* dexc = nullptr;          // device: exceptions, (position << 32 | value)
    uint64_t* dexc_count = nullptr;
    uint64_t exc_cap = 0;
    // host, page-locked: the bytes land in CHUNKS of chunk_bytes (chunk c = positions [c * chunk_bytes, ...)), allocated one after
    // the other by a thread of their own while the text is uploaded and the first slices are sorted: page-locking memory costs ~4 ms
    // per 64 MB -- 240 ms for a C3-sized array, which a first call would otherwise pay up front -- and a slice only needs the chunks
    // under it (chunk_ready blocks until chunk c exists).  Kept between calls with the device block.
    std::vector<char*>* chunks = nullptr;
    uint64_t chunk_bytes = 0;
    std::function<void(uint64_t)> chunk_ready;
    std::vector<uint64_t> hexc_store;  // host: the exceptions (pageable: a few MB at the end)
    uint64_t* hexc = nullptr;
    uint32_t workers = 1;
    struct Task { BackendEvent ev; uint64_t base, cnt; };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Task> tasks;
    bool closing = false, busy = false;
    std::thread dispatcher;
    std::exception_ptr failure;

    bool narrow() const { return d8 != nullptr; }
    // f(chunk pointer + offset, position, length) for the pieces of [base, base + cnt) chunk by chunk
    template <typename F> void for_pieces(uint64_t base, uint64_t cnt, F&& f)
    {
        for (uint64_t pos = base; pos < base + cnt;) {
            const uint64_t c = pos / chunk_bytes, off = pos - c * chunk_bytes;
            const uint64_t len = std::min<uint64_t>(chunk_bytes - off, base + cnt - pos);
            f(c, off, pos, len);
            pos += len;
        }
    }
    void widen_slice(uint64_t base, uint64_t cnt)
    {
        if (sizeof(idx_t) != 4) return;
        for_pieces(base, cnt, [this](uint64_t c, uint64_t off, uint64_t pos, uint64_t len) {
            uint32_t* out = reinterpret_cast<uint32_t*>(LCP) + pos;
            const uint8_t* in = reinterpret_cast<const uint8_t*>((*chunks)[c]) + off;
#ifdef CAPS_EMUL
            widen_bytes(in, out, len);
#else
            const uint32_t K = (uint32_t)std::min<uint64_t>(workers, std::max<uint64_t>(1, len >> 20));
            if (K <= 1) { widen_bytes(in, out, len); return; }
            std::vector<std::thread> th;
            const uint64_t per = ((len + K - 1) / K + 63) & ~uint64_t(63);
            for (uint32_t k = 0; k < K; ++k) {
                const uint64_t a = std::min<uint64_t>(len, (uint64_t)k * per), b = std::min<uint64_t>(len, a + per);
                if (a < b) th.emplace_back([=] { widen_bytes(in + a, out + a, b - a); });
            }
            for (auto& t : th) t.join();
#endif
        });
    }
    void start()
    {
#ifndef CAPS_EMUL
        if (!narrow()) return;
        dispatcher = std::thread([this] {
            for (;;) {
                Task t;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [this] { return closing || !tasks.empty(); });
                    if (tasks.empty()) return;
                    t = tasks.front();
                    tasks.pop_front();
                    busy = true;
                }
                try {
                    Backend::wait_event(t.ev);
                    Backend::destroy_event(t.ev);
                    widen_slice(t.base, t.cnt);
                } catch (...) { std::lock_guard<std::mutex> lk(mu); failure = std::current_exception(); }
                { std::lock_guard<std::mutex> lk(mu); busy = false; }
                cv.notify_all();
            }
        });
#endif
    }
    void drain()                       // every slice handed over so far has been widened
    {
#ifndef CAPS_EMUL
        if (!dispatcher.joinable()) return;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return tasks.empty() && !busy; });
#endif
    }
    void stop()
    {
#ifndef CAPS_EMUL
        if (!dispatcher.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); closing = true; }
        cv.notify_all();
        dispatcher.join();
#endif
    }
    ~HostCopySink() override { stop(); }

    void wave_done(Backend& be, uint64_t base, uint64_t cnt) override
    {
        Backend::stream_wait(copy_stream, be.record());
        if (narrow()) {
            Backend cb(copy_stream);
            const uint64_t rounds = (cnt + 256ull * NARROW_PER - 1) / (256ull * NARROW_PER);
            CAPS_LAUNCH(lcp_narrow_kernel, capped_grid(std::min<uint64_t>(rounds, 4096), 256), 256, cb,
                        reinterpret_cast<const uint32_t*>(dLCP) + base, cnt, base, d8 + base, dexc_count, exc_cap, dexc);
            for_pieces(base, cnt, [this](uint64_t c, uint64_t off, uint64_t pos, uint64_t len) {
                chunk_ready(c);
                Backend::d2h_on(copy_stream, (*chunks)[c] + off, d8 + pos, len);
            });
            const BackendEvent ev = Backend::record_on(copy_stream);
            Backend::d2h_on(copy_stream, SA + base, dSA + base, cnt * sizeof(idx_t));
#ifdef CAPS_EMUL
            widen_slice(base, cnt);
            (void)ev;
#else
            { std::lock_guard<std::mutex> lk(mu); tasks.push_back(Task{ev, base, cnt}); }
            cv.notify_all();
#endif
        } else {
            Backend::d2h_on(copy_stream, SA + base, dSA + base, cnt * sizeof(idx_t));
            Backend::d2h_on(copy_stream, LCP + base, dLCP + base, cnt * sizeof(idx_t));
        }
        copied += cnt;
    }
    void reset(Backend&) override
    {
        Backend::sync_stream(copy_stream);                  // nothing reads the device arrays any more
        drain();
        if (narrow()) { Backend::memset_on(copy_stream, dexc_count, 0, sizeof(uint64_t)); Backend::sync_stream(copy_stream); }
        copied = 0;
    }
    // after the last slice: the values of 255 and more.  false: more of them than the list holds -- the caller copies LCP at full width
    bool finish_lcp(uint64_t n)
    {
        if (!narrow()) { Backend::sync_stream(copy_stream); return true; }        // (the last copies have landed)
        uint64_t ne = 0;
        Backend::d2h_on(copy_stream, &ne, dexc_count, sizeof(uint64_t));
        Backend::sync_stream(copy_stream);
        drain();
        if (failure) std::rethrow_exception(failure);
        if (ne > exc_cap) return false;
        if (ne) {
            hexc_store.resize(ne);
            hexc = hexc_store.data();
            Backend::d2h_on(copy_stream, hexc, dexc, ne * sizeof(uint64_t));
            Backend::sync_stream(copy_stream);
            auto patch = [this, n](uint64_t a, uint64_t b) {
                for (uint64_t i = a; i < b; ++i) {
                    const uint64_t pos = hexc[i] >> 32;
                    if (pos < n) LCP[pos] = (idx_t)(hexc[i] & 0xFFFFFFFFull);
                }
            };
#ifdef CAPS_EMUL
            patch(0, ne);
#else
            const uint32_t K = (uint32_t)std::min<uint64_t>(workers, std::max<uint64_t>(1, ne >> 16));
            if (K <= 1) patch(0, ne);
            else {                                           // (scattered stores: a few million of them on a repeat-rich genome)
                std::vector<std::thread> th;
                for (uint32_t k = 0; k < K; ++k) th.emplace_back(patch, ne * k / K, ne * (k + 1) / K);
                for (auto& t : th) t.join();
            }
#endif
        }
        return true;
    }
};
#ifndef CAPS_HOST_WAVES
#define CAPS_HOST_WAVES 12
#endif

// Device memory of the host-buffer entry points, kept between calls: hipMalloc + hipFree of the
// text, the result arrays and the workspace cost far more than the build they serve (n = 1e9:
// 1.2 s against 28 ms).  One grow-only block per process, re-allocated when a larger build or
// another device asks; released by caps_sa_hip_release_cache() (or at process exit).  Calls of the
// host-buffer entry points are serialised on it.
struct HostPathCache {
    std::mutex mu;
    int device = -1;
    char* base = nullptr;
    size_t bytes = 0;
    std::vector<char*> host_chunks;    // page-locked staging of the LCP bytes (HostCopySink), host_chunk_bytes each
    size_t host_chunk_bytes = 0;
    uint64_t calls = 0;                // host-buffer builds of this process so far
};
inline HostPathCache& host_cache()
{
    static HostPathCache c;
    return c;
}

inline void release_host_cache_locked(HostPathCache& c)
{
    if (c.base) {
        Backend be(nullptr);
        be.free(c.base);
    }
    for (char* q : c.host_chunks) Backend::host_free(q);
    c.host_chunks.clear();
    c.host_chunk_bytes = 0;
    c.base = nullptr;
    c.bytes = 0;
    c.device = -1;
}

template <typename idx_t>
int build_host(const char* T, uint64_t n, uint64_t p_arg, uint64_t max_context, idx_t* SA, idx_t* LCP, int device,
               caps_sa_stats* stats)
{
    if (int rc = check_common<idx_t>(T, n, max_context)) return rc;
    if (n && (!SA || !LCP)) return fail(CAPS_SA_EINVAL, "null output");
    DeviceScope restore_device_;
    if (int rc = set_device(device)) return rc;
    return guarded([&]() -> int {
        HostPathCache& hc = host_cache();
        std::lock_guard<std::mutex> lock(hc.mu);
        Backend be(nullptr);
        auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
        // the text arena is sized for 2-bit codes when the first MiB of the text shows at most 4 distinct bytes (a text that
        // turns out to have more is built again below with the arena of 8-bit codes)
        int text_bits = 8;
        {
            bool seen[256] = {false};
            int sigma = 0;
            const uint64_t look = n < (1u << 20) ? n : (1u << 20);
            for (uint64_t i = 0; i < look && sigma <= 4; ++i)
                if (!seen[(uint8_t)T[i]]) { seen[(uint8_t)T[i]] = true; ++sigma; }
            if (sigma <= 4 && !std::getenv("CAPS_SA_FULL_ALPHABET")) text_bits = 2;
        }
        const uint64_t earlier_calls = hc.calls++;
        for (;; text_bits = 8) {
        const Plan<idx_t> need = make_plan<idx_t>(n, p_arg, nullptr, text_bits);
        const size_t off_sa = up(n ? n : 1), off_lcp = off_sa + up((n ? n : 1) * sizeof(idx_t));
        const size_t off_ws = off_lcp + up((n ? n : 1) * sizeof(idx_t));
        // waves pay when the build is long enough to hide: measured at C2 (256 Mi: 5 ms of build against 38 ms of copies) twelve
        // waves cost 14 ms (per-wave host synchronisations, 24 copies instead of 2), at C3 (3e9) they save 31 of 524 ms
        uint32_t waves = n >= (400ull << 20) ? CAPS_HOST_WAVES : 1u;
        if (const char* e = std::getenv("CAPS_SA_HOST_WAVES")) waves = (uint32_t)std::max(1, std::atoi(e));
        // LCP as bytes: with waves (the builds long enough for the link to matter), 32-bit indices -- and from the SECOND host build of
        // a process on: the page-locked staging costs ~240 ms to allocate at C3 (measured: not hidden by allocating it on a thread
        // beside the upload -- the runtime serialises the two), which a process that builds once (the CLI) would pay for a saving
        // of 150 ms; one that builds again and again pays it once.  CAPS_SA_HOST_NARROW_LCP=0 / 1: never / from the first build.
        bool narrow = sizeof(idx_t) == 4 && waves > 1 && earlier_calls > 0;
        if (const char* e = std::getenv("CAPS_SA_HOST_NARROW_LCP")) narrow = sizeof(idx_t) == 4 && std::atoi(e) != 0;
        // (the device block has room for the bytes from the first build on: growing it by a second build would free and re-allocate
        // ~130 GB, which takes seconds)
        const bool narrow_later = sizeof(idx_t) == 4 && waves > 1 && !(std::getenv("CAPS_SA_HOST_NARROW_LCP") && !narrow);
        const uint64_t exc_cap = narrow || narrow_later ? std::max<uint64_t>(1024, n / 32) : 0;
        const size_t narrow_dev = narrow || narrow_later ? up(n) + up((exc_cap + 1) * sizeof(uint64_t)) + 512 : 0;
        const size_t ws_room = need.bytes + 1024 + wave_scratch_bytes<idx_t>(wave_scratch_elems(n, waves));
        const size_t total = off_ws + ws_room + narrow_dev;
        if (hc.device != device || hc.bytes < total) {
            std::vector<char*> keep;
            keep.swap(hc.host_chunks);                       // (the staging chunks survive a larger device block)
            const size_t keep_bytes = hc.host_chunk_bytes;
            release_host_cache_locked(hc);
            hc.host_chunks.swap(keep);
            hc.host_chunk_bytes = keep_bytes;
            const auto a0 = std::chrono::steady_clock::now();
            hc.base = static_cast<char*>(be.alloc(total));
            hc.bytes = total;
            hc.device = device;
            if (std::getenv("CAPS_SA_DEBUG_ALLOC"))
                std::fprintf(stderr, "[alloc] device block of %zu bytes: %.1f ms\n", total,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count());
        }
        // the page-locked chunks for the LCP bytes, allocated beside the upload (HostCopySink::chunks)
        constexpr uint32_t N_CHUNKS = 16;
        const uint64_t chunk_bytes = narrow ? up((n + N_CHUNKS - 1) / N_CHUNKS) : 0;
        const size_t chunks_need = narrow ? (size_t)((n + chunk_bytes - 1) / chunk_bytes) : 0;
        std::thread ring_alloc;
        std::exception_ptr ring_error;
        std::mutex chunk_mu;
        std::condition_variable chunk_cv;
        size_t chunks_done = hc.host_chunks.size();
        if (narrow && (hc.host_chunk_bytes != chunk_bytes || hc.host_chunks.size() < chunks_need)) {
            for (char* q : hc.host_chunks) Backend::host_free(q);
            hc.host_chunks.assign(chunks_need, nullptr);
            hc.host_chunk_bytes = chunk_bytes;
            chunks_done = 0;
            auto alloc = [&, chunks_need, chunk_bytes, device]() {
                try {
                    const auto a0 = std::chrono::steady_clock::now();
                    if (set_device(device) != CAPS_SA_OK) throw HipError("hipSetDevice failed");
                    for (size_t c = 0; c < chunks_need; ++c) {
                        char* q = static_cast<char*>(Backend::host_alloc(chunk_bytes));
                        { std::lock_guard<std::mutex> lk(chunk_mu); hc.host_chunks[c] = q; chunks_done = c + 1; }
                        chunk_cv.notify_all();
                    }
                    if (std::getenv("CAPS_SA_DEBUG_ALLOC"))
                        std::fprintf(stderr, "[alloc] %zu page-locked chunks of %llu bytes: %.1f ms (beside the upload and the first slices)\n", chunks_need,
                                     (unsigned long long)chunk_bytes, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count());
                } catch (...) {
                    { std::lock_guard<std::mutex> lk(chunk_mu); ring_error = std::current_exception(); }
                    chunk_cv.notify_all();
                }
            };
#ifdef CAPS_EMUL
            alloc();
#else
            ring_alloc = std::thread(alloc);
#endif
        }
        struct JoinGuard { std::thread& t; ~JoinGuard() { if (t.joinable()) t.join(); } } ring_join{ring_alloc};
        char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(hc.base) + 255) & ~uintptr_t(255));
        uint8_t* dT = reinterpret_cast<uint8_t*>(base);
        idx_t* dSA = reinterpret_cast<idx_t*>(base + off_sa);
        idx_t* dLCP = reinterpret_cast<idx_t*>(base + off_lcp);
        if (std::getenv("CAPS_SA_DEBUG_ALLOC")) {       // where the build's arrays sit (a faulting address can then be placed)
            const Plan<idx_t> pl = make_plan<idx_t>(n, p_arg, base + off_ws, text_bits);
            std::fprintf(stderr, "[alloc] block %p + %zu | T %p SA %p LCP %p ws %p | P %p A.key %p A.sa %p A.lcp %p B.key %p B.sa %p B.lcp %p seg1.rec %p SA_.key %p "
                         "Pm %p desc %p bk.params %p bk.count %p bk.sub.rec %p end %p\n", (void*)hc.base, hc.bytes, (void*)dT, (void*)dSA, (void*)dLCP,
                         (void*)(base + off_ws), (void*)pl.P, (void*)pl.A.key, (void*)pl.A.sa, (void*)pl.A.lcp, (void*)pl.B.key, (void*)pl.B.sa, (void*)pl.B.lcp,
                         (void*)pl.seg1.tile_rec, (void*)pl.SA_.key, (void*)pl.Pm, (void*)pl.desc, (void*)pl.bk.params, (void*)pl.bk.count,
                         (void*)pl.bk.sub.tile_rec, (void*)(base + off_ws + pl.bytes));
        }
        BackendEvent h0 = be.record();
        be.h2d(dT, T, n);
        BackendEvent h1 = be.record();
        be.sync();
        caps_sa_stats local;
        HostCopySink<idx_t> sink;
        sink.copy_stream = Backend::create_stream();
        sink.SA = SA;
        sink.LCP = LCP;
        sink.dSA = dSA;
        sink.dLCP = dLCP;
        if (narrow) {
            char* nb = base + off_ws + up(ws_room);
            sink.d8 = reinterpret_cast<uint8_t*>(nb);
            sink.dexc = reinterpret_cast<uint64_t*>(nb + up(n));
            sink.dexc_count = sink.dexc + exc_cap;
            sink.exc_cap = exc_cap;
            sink.chunks = &hc.host_chunks;
            sink.chunk_bytes = chunk_bytes;
            sink.chunk_ready = [&](uint64_t c) {
                std::unique_lock<std::mutex> lk(chunk_mu);
                chunk_cv.wait(lk, [&] { return chunks_done > c || ring_error; });
                if (ring_error) std::rethrow_exception(ring_error);
            };
            sink.workers = host_worker_threads();
            be.memset(sink.dexc_count, 0, sizeof(uint64_t));
            be.sync();
        }
        sink.start();
        struct StreamGuard { decltype(Backend::stream) s; ~StreamGuard() { Backend::destroy_stream(s); } } guard{sink.copy_stream};
        bool served = false;
        const auto t0 = std::chrono::steady_clock::now();
        int rc = build_device<idx_t>(dT, n, p_arg, max_context, dSA, dLCP, base + off_ws, ws_room - 256, nullptr, &local, waves,
                                     &sink, &served, text_bits);
        if (rc == CAPS_SA_EALPHABET && text_bits == 2) { Backend::sync_stream(sink.copy_stream); sink.drain(); sink.stop(); continue; }
        if (rc) { Backend::sync_stream(sink.copy_stream); sink.drain(); return rc; }
        bool lcp_as_bytes = sink.narrow();
        if (!served || sink.copied != n) {           // another construction (samplesort path, tiny input): nothing was streamed out
            Backend::sync_stream(sink.copy_stream);
            sink.drain();
            be.d2h(SA, dSA, n * sizeof(idx_t));
            be.d2h(LCP, dLCP, n * sizeof(idx_t));
            be.sync();
            local.result_waves = 1;
            lcp_as_bytes = false;
        } else if (!sink.finish_lcp(n)) {            // more values of 255 and more than the list holds: LCP at full width
            be.d2h(LCP, dLCP, n * sizeof(idx_t));
            be.sync();
            lcp_as_bytes = false;
        }
        sink.stop();
        local.lcp_bytes_on_link = lcp_as_bytes ? 1u : (uint32_t)sizeof(idx_t);
        local.ms_h2d = be.elapsed_ms(h0, h1);
        // build + result copies, overlapped: host wall clock from the launch of the build to the last byte on the host, minus the build
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        local.ms_d2h = wall > local.ms_total ? wall - local.ms_total : 0.0;
        if (stats) *stats = local;
        return CAPS_SA_OK;
        }
    });
}

// ---- one process, several GPUs (`devices, n_devices` under construct(); 8e) -----------------------------
// The sharded direct path of shard.h with one Shard per device, driven from this process: text replicated; by default every
// device classifies the WHOLE text (level A) and keeps the groups it owns -- nothing crosses a link -- sorts them, and the
// slices of SA / LCP are copied to the caller's arrays by all devices at once.  With CAPS_SA_SHARD_EXCHANGE=1 every device
// classifies every n_devices-th tile and the blocks of (key, sa) go straight from device to device (peer copies over xGMI:
// inside one process nothing else is needed -- the multi-PROCESS driver, caps_sa_dist.py, does that exchange with RCCL).  A device may be listed more than once (several ranks share it: how a one-GPU box tests this).
// Texts the direct path does not take (long repeats) are built on devices[0] alone.
template <typename idx_t> struct MultiRank {
    int dev = 0;
    decltype(Backend::stream) stream = nullptr;
    std::unique_ptr<Backend> be;
    std::unique_ptr<Shard<idx_t>> sh;
    std::vector<void*> owned;
    uint8_t* dT = nullptr;
    uint64_t *send_k = nullptr, *recv_k = nullptr, *report = nullptr;
    idx_t *send_s = nullptr, *recv_s = nullptr, *dSA = nullptr, *dLCP = nullptr;
    caps_sa_shard_info info;
    std::vector<uint64_t> sc, rc;
    int sort_code = 0;
    double ms_upload = 0, ms_build = 0, ms_download = 0;      // host wall clock of this device's stages (caps_sa_stats)
    template <typename T> T* get(size_t count)
    {
        T* q = static_cast<T*>(be->alloc((count ? count : 1) * sizeof(T)));
        owned.push_back(q);
        return q;
    }
    ~MultiRank()
    {
        if (set_device(dev) != CAPS_SA_OK) return;
        sh.reset();
        if (be) for (void* q : owned) be->free(q);
        be.reset();
        Backend::destroy_stream(stream);
    }
};

// fn(rank) on every rank, each on a thread of its own with its device current (emulation: one after the other)
template <typename R, typename F> void for_each_rank(std::vector<std::unique_ptr<R>>& ranks, F&& fn)
{
#ifdef CAPS_EMUL
    for (auto& r : ranks) fn(*r);
#else
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(ranks.size());
    for (size_t i = 0; i < ranks.size(); ++i)
        th.emplace_back([&, i]() {
            try {
                if (set_device(ranks[i]->dev) != CAPS_SA_OK) throw HipError("hipSetDevice failed");
                fn(*ranks[i]);
            } catch (...) { err[i] = std::current_exception(); }
        });
    for (auto& t : th) t.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
#endif
}

template <typename idx_t>
int build_host(const char* T, uint64_t n, uint64_t p_arg, uint64_t max_context, idx_t* SA, idx_t* LCP, int device,
               caps_sa_stats* stats);

template <typename idx_t>
int build_multi(const char* T, uint64_t n, uint64_t p_arg, uint64_t max_context, idx_t* SA, idx_t* LCP, const int* devices,
                int n_devices, caps_sa_stats* stats)
{
    if (!devices || n_devices < 1) return fail(CAPS_SA_EINVAL, "no devices");
    if (int rc = check_common<idx_t>(T, n, max_context)) return rc;
    if (n && (!SA || !LCP)) return fail(CAPS_SA_EINVAL, "null output");
    if (n_devices == 1) return build_host<idx_t>(T, n, p_arg, max_context, SA, LCP, devices[0], stats);
    // bounded context: the reference's merge trees are one sequence (csrc/bounded.h); built on devices[0] alone
    if (max_context != 0 && max_context < n) return build_host<idx_t>(T, n, p_arg, max_context, SA, LCP, devices[0], stats);
    DeviceScope restore_device_;
    for (int i = 0; i < n_devices; ++i)
        if (int rc = set_device(devices[i])) return rc;
    int fallback = CAPS_SA_FB_NONE;
    const int rc = guarded([&]() -> int {
        using clock = std::chrono::steady_clock;
        const auto t0 = clock::now();
        uint32_t p_eff = 0, ppp = 0;
        effective_params(n, p_arg, &p_eff, &ppp);
        if (p_eff < 2) { fallback = CAPS_SA_FB_SHAPE; return CAPS_SA_OK; }
        const int world = n_devices;
        auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::vector<std::unique_ptr<MultiRank<idx_t>>> ranks;
        for (int r = 0; r < world; ++r) {
            std::unique_ptr<MultiRank<idx_t>> q(new MultiRank<idx_t>);
            q->dev = devices[r];
            if (int e = set_device(q->dev)) return e;
            q->stream = Backend::create_stream();
            q->be.reset(new Backend(q->stream));
            q->dT = q->template get<uint8_t>(n);
            ranks.push_back(std::move(q));
        }
        // ---- the text to every device.  ONE upload over PCIe, to ranks[0], in chunks; every chunk goes on from there to the other
        // devices by peer copies (xGMI: one stream per destination, so the seven links of devices[0] work side by side) while the
        // next chunk is still coming up -- H2D + one chunk instead of `world` uploads one after the other from pageable memory
        // (8 x 53 ms at C3-size: more than the sharded build itself).
        {
            MultiRank<idx_t>& root = *ranks[0];
            if (int e = set_device(root.dev)) return e;
            for (int r = 1; r < world; ++r) Backend::enable_peer(root.dev, ranks[r]->dev);
            std::vector<decltype(Backend::stream)> fan((size_t)world, nullptr);
            struct FanGuard { std::vector<decltype(Backend::stream)>& f; ~FanGuard() { for (auto s : f) Backend::destroy_stream(s); } } fan_guard{fan};
            for (int r = 1; r < world; ++r) fan[r] = Backend::create_stream();
#ifdef CAPS_EMUL
            const uint64_t chunk = 1u << 12;                           // small: the CPU tests walk the chunk loop
#else
            const uint64_t chunk = 64ull << 20;
#endif
            for (uint64_t off = 0; off < n; off += chunk) {
                const uint64_t len = n - off < chunk ? n - off : chunk;
                root.be->h2d(root.dT + off, T + off, len);
                const BackendEvent ev = root.be->record();
                for (int r = 1; r < world; ++r) {
                    Backend::stream_wait(fan[r], ev);
                    Backend::peer_copy_on(fan[r], ranks[r]->dT + off, ranks[r]->dev, root.dT + off, root.dev, len);
                }
            }
            root.be->sync();
            root.ms_upload = ms(t0, clock::now());
            for (int r = 1; r < world; ++r) { Backend::sync_stream(fan[r]); ranks[r]->ms_upload = ms(t0, clock::now()); }
            root.be->release_events();
        }
        for (int r = 0; r < world; ++r) {
            MultiRank<idx_t>& q = *ranks[r];
            if (int e = set_device(q.dev)) return e;
            q.sh.reset(new Shard<idx_t>(q.dT, n, p_arg, r, world, q.stream));
            q.sh->info(&q.info);
        }
        if (ranks[0]->info.direct_fallback != CAPS_SA_FB_NONE) { fallback = (int)ranks[0]->info.direct_fallback; return CAPS_SA_OK; }
        const size_t W = (size_t)ranks[0]->info.n_streams + 2;
        for (auto& q : ranks) {
            if (int e = set_device(q->dev)) return e;
            q->send_k = q->template get<uint64_t>(q->info.send_capacity);
            q->send_s = q->template get<idx_t>(q->info.send_capacity);
            if (q->info.exchange) {
                q->recv_k = q->template get<uint64_t>(q->info.capacity);
                q->recv_s = q->template get<idx_t>(q->info.capacity);
            }
            q->dSA = q->template get<idx_t>(q->info.capacity);
            q->dLCP = q->template get<idx_t>(q->info.capacity);
            q->report = q->template get<uint64_t>(W);
        }
        const auto t1 = clock::now();
        auto t2 = t1;
        for (int attempt = 0;; ++attempt) {                       // second attempt: 64-bit keys after a slot overflow under 32
        // ---- level A on every device; the reports; the plan (identical on all ranks)
        for_each_rank(ranks, [&](MultiRank<idx_t>& q) {
            const auto a = clock::now();
            q.sh->scatter(q.send_k, q.send_s, q.report);
            q.be->sync();
            q.ms_build += ms(a, clock::now());
        });
        std::vector<uint64_t> all(W * world);
        for (int r = 0; r < world; ++r) {
            if (int e = set_device(ranks[r]->dev)) return e;
            ranks[r]->be->d2h(all.data() + W * r, ranks[r]->report, W * sizeof(uint64_t));
            ranks[r]->be->sync();
        }
        for (auto& q : ranks) {
            q->sc.assign(world, 0);
            q->rc.assign(world, 0);
            if (int e = set_device(q->dev)) return e;
            const int code = q->sh->plan(all.data(), q->sc.data(), q->rc.data());
            if (code != CAPS_SA_FB_NONE) { fallback = code; return CAPS_SA_OK; }
        }
        // ---- the exchange: block d of rank r's send buffers -> slot r of rank d's receive buffers
        for (auto& q : ranks) q->sh->info(&q->info);              // key_bytes of this attempt
        const bool exchange = ranks[0]->info.exchange != 0;      // (0: every device scattered the whole text and kept its groups)
        for (int r = 0; exchange && r < world; ++r) {
            MultiRank<idx_t>& src = *ranks[r];
            if (int e = set_device(src.dev)) return e;
            uint64_t so = 0;
            for (int d = 0; d < world; ++d) {
                MultiRank<idx_t>& dst = *ranks[d];
                const uint64_t cnt = src.sc[d];
                if (cnt != dst.rc[r]) throw std::runtime_error("send / receive counts disagree");
                uint64_t ro = 0;
                for (int q = 0; q < r; ++q) ro += dst.rc[q];
                const size_t kb = src.info.key_bytes;              // 4 under 32-bit keys: the key arrays are u32 then
                src.be->peer_copy(reinterpret_cast<char*>(dst.recv_k) + ro * kb, dst.dev, reinterpret_cast<const char*>(src.send_k) + so * kb,
                                  src.dev, cnt * kb);
                src.be->peer_copy(dst.recv_s + ro, dst.dev, src.send_s + so, src.dev, cnt * sizeof(idx_t));
                so += cnt;
            }
        }
        for (auto& q : ranks) { if (int e = set_device(q->dev)) return e; q->be->sync(); }
        t2 = clock::now();
        // ---- level B + tile sort of the owned groups; boundary LCPs between the slices
        for_each_rank(ranks, [&](MultiRank<idx_t>& q) {
            const auto a = clock::now();
            q.sort_code = exchange ? q.sh->sort_owned(q.recv_k, q.recv_s, q.dSA, q.dLCP) : q.sh->sort_owned(q.send_k, q.send_s, q.dSA, q.dLCP);
            q.be->sync();
            q.ms_build += ms(a, clock::now());
        });
        int worst = 0;
        for (auto& q : ranks) worst = q->sort_code > worst ? q->sort_code : worst;
        if (worst == CAPS_SA_FB_NONE) break;
        if (attempt > 0) throw std::runtime_error("the sharded direct path failed with 64-bit keys");
        for (auto& q : ranks) q->sh->set_key_bits(64);            // a slot overflowed under 32-bit keys on some rank: all again with 64
        }
        uint64_t prev = ~0ull;
        for (auto& q : ranks) {
            if (int e = set_device(q->dev)) return e;
            q->sh->fix_first_lcp(prev, q->dLCP);
            const uint64_t last = q->sh->last_sa();
            if (last != ~0ull) prev = last;
            q->sh->info(&q->info);
        }
        const auto t3 = clock::now();
        // ---- every device copies its slice to the caller's arrays
        uint64_t covered = 0;
        for (auto& q : ranks) {
            if (q->info.slice_off != covered) throw std::runtime_error("the ranks' slices do not tile the suffix array");
            covered += q->info.recv_total;
        }
        if (covered != n) throw std::runtime_error("the ranks' slices do not cover the suffix array");
        // (every device on a host thread and a stream of its own: the slices leave over all the devices' PCIe links at once)
        for_each_rank(ranks, [&](MultiRank<idx_t>& q) {
            const auto a = clock::now();
            q.be->d2h(SA + q.info.slice_off, q.dSA, q.info.recv_total * sizeof(idx_t));
            q.be->d2h(LCP + q.info.slice_off, q.dLCP, q.info.recv_total * sizeof(idx_t));
            q.be->sync();
            q.ms_download = ms(a, clock::now());
        });
        const auto t4 = clock::now();
        if (stats) {
            *stats = caps_sa_stats();
            stats->n = n;
            stats->idx_bytes = sizeof(idx_t);
            stats->p_eff = ranks[0]->info.p;
            stats->ppp = ranks[0]->info.ppp;
            stats->bits_per_char = ranks[0]->info.bits_per_char;
            stats->path_direct = 1;
            stats->direct_groups = ranks[0]->info.direct_groups;
            stats->direct_quantile = ranks[0]->info.direct_quantile;
            stats->ms_h2d = ms(t0, t1);                       // device setup: allocations + the text to every device
            stats->ms_partition = ms(t1, t2);                 // level A + exchange (host wall clock, all devices)
            stats->ms_merge_partitions = ms(t2, t3);          // level B + tile sort + boundary LCPs
            stats->ms_d2h = ms(t3, t4);
            stats->ms_total = ms(t1, t3);
            for (auto& q : ranks) { stats->slot_splits += q->info.slot_splits; stats->slot_splits_redone += q->info.slot_splits_redone; }
            for (auto& q : ranks) {
                stats->tie_groups_deferred += q->info.tie_groups_deferred;
                stats->tie_levels = std::max(stats->tie_levels, q->info.tie_levels);
            }
            stats->n_devices = (uint32_t)world;
            stats->result_waves = 1;
            stats->ms_upload_min = stats->ms_device_build_min = stats->ms_download_min = 1e300;
            for (auto& q : ranks) {
                stats->ms_upload_max = std::max(stats->ms_upload_max, q->ms_upload);
                stats->ms_upload_min = std::min(stats->ms_upload_min, q->ms_upload);
                stats->ms_device_build_max = std::max(stats->ms_device_build_max, q->ms_build);
                stats->ms_device_build_min = std::min(stats->ms_device_build_min, q->ms_build);
                stats->ms_download_max = std::max(stats->ms_download_max, q->ms_download);
                stats->ms_download_min = std::min(stats->ms_download_min, q->ms_download);
            }
        }
        return CAPS_SA_OK;
    });
    if (rc != CAPS_SA_OK) return rc;
    if (fallback != CAPS_SA_FB_NONE) {
        const int rc1 = build_host<idx_t>(T, n, p_arg, max_context, SA, LCP, devices[0], stats);
        if (rc1 == CAPS_SA_OK && stats && stats->path_fallback == CAPS_SA_FB_NONE) stats->path_fallback = (uint32_t)fallback;
        return rc1;
    }
    return CAPS_SA_OK;
}

template <typename idx_t>
int verify_device(const void* dT, uint64_t n, const void* dSA, const void* dLCP, void* stream, uint64_t* n_errors,
                  uint64_t cnt = ~0ull, uint32_t head = 1)
{
    if (!n_errors) return fail(CAPS_SA_EINVAL, "null n_errors");
    *n_errors = 0;
    if (cnt == ~0ull) cnt = n;
    if (cnt > n) return fail(CAPS_SA_EINVAL, "more entries than suffixes");
    if (n == 0 || cnt == 0) return CAPS_SA_OK;
    if (!dT || !dSA || !dLCP) return fail(CAPS_SA_EINVAL, "null pointer");
    return guarded([&]() -> int {
        Backend be(static_cast<decltype(Backend::stream)>(stream));
        DevAllocs da(be);
        const size_t words = (n + 31) / 32;
        uint32_t* seen = da.get<uint32_t>(words);
        uint64_t* err = da.get<uint64_t>(1);
        be.memset(seen, 0, words * sizeof(uint32_t));
        be.memset(err, 0, sizeof(uint64_t));
        const uint64_t want = (cnt + 255) / 256;
        CAPS_LAUNCH((verify_kernel<idx_t>), want < 16384 ? want : 16384, 256, be, static_cast<const int8_t*>(dT), n,
                    static_cast<const idx_t*>(dSA), static_cast<const idx_t*>(dLCP), cnt, head, seen, err);
        be.d2h(n_errors, err, sizeof(uint64_t));
        be.sync();
        return CAPS_SA_OK;
    });
}

// Text on the device for the kernel-level entry points.
struct DevText {
    uint8_t* raw = nullptr;
    uint32_t* P = nullptr;
    int bits = 0;
};

inline DevText upload_text(Backend& be, DevAllocs& da, const char* T, uint64_t n)
{
    DevText t;
    t.raw = da.get<uint8_t>(n ? n : 1);
    t.P = da.get<uint32_t>(text_alloc_words(n ? n : 1));
    uint32_t* present = da.get<uint32_t>(16);
    uint8_t* lut = da.get<uint8_t>(256);
    be.h2d(t.raw, T, n);
    t.bits = prepare_text(be, t.raw, n, t.P, present, lut);
    return t;
}

template <typename idx_t> SegBufs one_segment(Backend& be, DevAllocs& da, uint64_t cnt)
{
    SegBufs s;
    s.G = 1;
    s.seg_start = da.get<uint64_t>(2);
    s.tile_off = da.get<uint32_t>(2);
    s.tile_rec = da.get<TileInfo>(tiles_of(cnt) + 2);
    s.out2 = da.get<uint64_t>(2);
    CAPS_LAUNCH(uniform_segments_kernel, 1, 256, be, s.seg_start, 1u, cnt, cnt);
    prepare_segments(be, s, tiles_of(cnt));
    return s;
}

template <typename idx_t> ElemBuf<idx_t> elem_buf(DevAllocs& da, uint64_t cnt)
{
    ElemBuf<idx_t> b;
    b.key = da.get<uint64_t>(cnt ? cnt : 1);
    b.sa = da.get<idx_t>(cnt ? cnt : 1);
    b.lcp = da.get<idx_t>(cnt ? cnt : 1);
    return b;
}

template <typename idx_t> int check_positions(const idx_t* v, uint64_t cnt, uint64_t n, const char* what)
{
    for (uint64_t i = 0; i < cnt; ++i)
        if ((uint64_t)v[i] >= n) return fail(CAPS_SA_EINVAL, what);
    return CAPS_SA_OK;
}

template <typename idx_t>
int sort_suffixes(const char* T, uint64_t n, const idx_t* idx, uint64_t cnt, idx_t* out_sa, idx_t* out_lcp, int device)
{
    if (int rc = check_common<idx_t>(T, n, 0)) return rc;
    if (cnt == 0) return CAPS_SA_OK;
    if (!idx || !out_sa || !out_lcp) return fail(CAPS_SA_EINVAL, "null pointer");
    if (int rc = check_positions(idx, cnt, n, "suffix position out of range")) return rc;
    DeviceScope restore_device_;
    if (int rc = set_device(device)) return rc;
    return guarded([&]() -> int {
        Backend be(nullptr);
        DevAllocs da(be);
        DevText t = upload_text(be, da, T, n);
        ElemBuf<idx_t> a = elem_buf<idx_t>(da, cnt), b = elem_buf<idx_t>(da, cnt);
        TileDesc* desc = da.get<TileDesc>(tiles_of(cnt) + 2);
        idx_t* osa = da.get<idx_t>(cnt);
        idx_t* olcp = da.get<idx_t>(cnt);
        be.h2d(a.sa, idx, cnt * sizeof(idx_t));
        SegBufs s = one_segment<idx_t>(be, da, cnt);
        const uint32_t g = (uint32_t)((cnt + 255) / 256);
        if (t.bits == 2) {
            CAPS_LAUNCH((make_keys_kernel<idx_t, 2>), g, 256, be, (const uint32_t*)t.P, (const idx_t*)a.sa, cnt, a.key);
            SortOpts o;
            o.need_lcp = true;
            SortResult<idx_t> r = segmented_sort<idx_t, 2>(be, t.P, n, desc, s, tiles_of(cnt), cnt, a, b, cnt, o);
            finalize<idx_t, 2>(be, t.P, n, r, osa, olcp);
        } else {
            CAPS_LAUNCH((make_keys_kernel<idx_t, 8>), g, 256, be, (const uint32_t*)t.P, (const idx_t*)a.sa, cnt, a.key);
            SortOpts o;
            o.need_lcp = true;
            SortResult<idx_t> r = segmented_sort<idx_t, 8>(be, t.P, n, desc, s, tiles_of(cnt), cnt, a, b, cnt, o);
            finalize<idx_t, 8>(be, t.P, n, r, osa, olcp);
        }
        be.d2h(out_sa, osa, cnt * sizeof(idx_t));
        be.d2h(out_lcp, olcp, cnt * sizeof(idx_t));
        be.sync();
        return CAPS_SA_OK;
    });
}

// Independent sorts of consecutive segments of a suffix list (the shape of phase 2:
// sort_partition over every partition, src/Suffix_Array.cpp:388-404), with finished segments
// sitting out later passes.  out_lcp at a segment head = lcp with the last suffix of the
// previous non-empty segment (what compute_partition_boundary_lcp produces, cpp:431-447).
template <typename idx_t>
int sort_segments(const char* T, uint64_t n, const idx_t* idx, uint64_t cnt, const uint64_t* seg_start, uint64_t G,
                  idx_t* out_sa, idx_t* out_lcp, int device)
{
    if (int rc = check_common<idx_t>(T, n, 0)) return rc;
    if (cnt == 0) return CAPS_SA_OK;
    if (!idx || !out_sa || !out_lcp || !seg_start || G == 0 || G > 0x7fffffffull) return fail(CAPS_SA_EINVAL, "bad argument");
    if (seg_start[0] != 0 || seg_start[G] != cnt) return fail(CAPS_SA_EINVAL, "segments must cover [0, cnt)");
    uint64_t max_len = 0, n_tiles = 0;
    for (uint64_t g = 0; g < G; ++g) {
        if (seg_start[g + 1] < seg_start[g]) return fail(CAPS_SA_EINVAL, "segment offsets must be non-decreasing");
        const uint64_t len = seg_start[g + 1] - seg_start[g];
        max_len = len > max_len ? len : max_len;
        n_tiles += tiles_of(len);
    }
    if (int rc = check_positions(idx, cnt, n, "suffix position out of range")) return rc;
    DeviceScope restore_device_;
    if (int rc = set_device(device)) return rc;
    return guarded([&]() -> int {
        Backend be(nullptr);
        DevAllocs da(be);
        DevText t = upload_text(be, da, T, n);
        ElemBuf<idx_t> a = elem_buf<idx_t>(da, cnt), b = elem_buf<idx_t>(da, cnt);
        TileDesc* desc = da.get<TileDesc>(n_tiles + 2);
        idx_t* osa = da.get<idx_t>(cnt);
        idx_t* olcp = da.get<idx_t>(cnt);
        SegBufs s;
        s.G = (uint32_t)G;
        s.seg_start = da.get<uint64_t>(G + 1);
        s.tile_off = da.get<uint32_t>(G + 1);
        s.tile_rec = da.get<TileInfo>(n_tiles + 2);
        s.out2 = da.get<uint64_t>(2);
        be.h2d(a.sa, idx, cnt * sizeof(idx_t));
        be.h2d(s.seg_start, seg_start, (G + 1) * sizeof(uint64_t));
        prepare_segments(be, s, n_tiles);
        const uint32_t g = (uint32_t)((cnt + 255) / 256);
        if (t.bits == 2) {
            CAPS_LAUNCH((make_keys_kernel<idx_t, 2>), g, 256, be, (const uint32_t*)t.P, (const idx_t*)a.sa, cnt, a.key);
            SortOpts o;
            o.need_lcp = true;
            o.skip_finished = true;
            SortResult<idx_t> r = segmented_sort<idx_t, 2>(be, t.P, n, desc, s, (uint32_t)n_tiles, max_len, a, b, cnt, o);
            finalize<idx_t, 2>(be, t.P, n, r, osa, olcp);
        } else {
            CAPS_LAUNCH((make_keys_kernel<idx_t, 8>), g, 256, be, (const uint32_t*)t.P, (const idx_t*)a.sa, cnt, a.key);
            SortOpts o;
            o.need_lcp = true;
            o.skip_finished = true;
            SortResult<idx_t> r = segmented_sort<idx_t, 8>(be, t.P, n, desc, s, (uint32_t)n_tiles, max_len, a, b, cnt, o);
            finalize<idx_t, 8>(be, t.P, n, r, osa, olcp);
        }
        be.d2h(out_sa, osa, cnt * sizeof(idx_t));
        be.d2h(out_lcp, olcp, cnt * sizeof(idx_t));
        be.sync();
        return CAPS_SA_OK;
    });
}

template <typename idx_t>
int merge_runs(const char* T, uint64_t n, const idx_t* X, uint64_t len_x, const idx_t* Y, uint64_t len_y, const idx_t* LX,
               const idx_t* LY, idx_t* Z, idx_t* LZ, int device)
{
    if (int rc = check_common<idx_t>(T, n, 0)) return rc;
    const uint64_t cnt = len_x + len_y;
    if (cnt == 0) return CAPS_SA_OK;
    if ((len_x && (!X || !LX)) || (len_y && (!Y || !LY)) || !Z || !LZ) return fail(CAPS_SA_EINVAL, "null pointer");
    if (int rc = check_positions(X, len_x, n, "suffix position out of range")) return rc;
    if (int rc = check_positions(Y, len_y, n, "suffix position out of range")) return rc;
    DeviceScope restore_device_;
    if (int rc = set_device(device)) return rc;
    return guarded([&]() -> int {
        Backend be(nullptr);
        DevAllocs da(be);
        DevText t = upload_text(be, da, T, n);
        ElemBuf<idx_t> a = elem_buf<idx_t>(da, cnt), b = elem_buf<idx_t>(da, cnt);
        TileDesc* desc = da.get<TileDesc>(tiles_of(cnt) + 2);
        be.h2d(a.sa, X, len_x * sizeof(idx_t));
        be.h2d(a.sa + len_x, Y, len_y * sizeof(idx_t));
        // LX / LY are accepted for signature parity with the reference's merge; the kernel
        // rebuilds the LCPs of the merged run from the keys (kernels.h, "LCPs").
        (void)LX; (void)LY;
        SegBufs s = one_segment<idx_t>(be, da, cnt);
        const SegDesc sd = s.desc();
        const uint32_t nt = tiles_of(cnt), g = (uint32_t)((cnt + 255) / 256);
        const uint32_t pg = nt < be.persistent_blocks() ? nt : be.persistent_blocks();
        be.memset(desc + nt, 0, sizeof(uint32_t));          // merge_partition_kernel counts the tiles it lists there
        if (t.bits == 2) {
            CAPS_LAUNCH((make_keys_kernel<idx_t, 2>), g, 256, be, (const uint32_t*)t.P, (const idx_t*)a.sa, cnt, a.key);
            CAPS_LAUNCH((merge_partition_kernel<idx_t, 2, true>), (nt + 255) / 256, 256, be, sd, (const uint32_t*)t.P, n, (uint64_t)TILE_E,
                        len_x, 0u, 1u, (const uint64_t*)a.key, (const idx_t*)a.sa, desc, (uint64_t*)nullptr, 0u, reinterpret_cast<uint32_t*>(desc + nt));
            CAPS_LAUNCH((merge_pass_kernel<idx_t, 2, true>), pg, TILE_NT, be, (const TileDesc*)desc, (const uint32_t*)(desc + nt), (const uint32_t*)t.P, n,
                        (const uint64_t*)a.key, (const idx_t*)a.sa, b.key, b.sa, b.lcp, (uint64_t*)nullptr);
        } else {
            CAPS_LAUNCH((make_keys_kernel<idx_t, 8>), g, 256, be, (const uint32_t*)t.P, (const idx_t*)a.sa, cnt, a.key);
            CAPS_LAUNCH((merge_partition_kernel<idx_t, 8, true>), (nt + 255) / 256, 256, be, sd, (const uint32_t*)t.P, n, (uint64_t)TILE_E,
                        len_x, 0u, 1u, (const uint64_t*)a.key, (const idx_t*)a.sa, desc, (uint64_t*)nullptr, 0u, reinterpret_cast<uint32_t*>(desc + nt));
            CAPS_LAUNCH((merge_pass_kernel<idx_t, 8, true>), pg, TILE_NT, be, (const TileDesc*)desc, (const uint32_t*)(desc + nt), (const uint32_t*)t.P, n,
                        (const uint64_t*)a.key, (const idx_t*)a.sa, b.key, b.sa, b.lcp, (uint64_t*)nullptr);
        }
        be.d2h(Z, b.sa, cnt * sizeof(idx_t));
        be.d2h(LZ, b.lcp, cnt * sizeof(idx_t));
        be.sync();
        return CAPS_SA_OK;
    });
}

template <typename idx_t>
int upper_bounds(const char* T, uint64_t n, const idx_t* X, uint64_t cnt, const idx_t* piv, uint64_t npiv, idx_t* out, int device)
{
    if (int rc = check_common<idx_t>(T, n, 0)) return rc;
    if (npiv == 0) return CAPS_SA_OK;
    if ((cnt && !X) || !piv || !out) return fail(CAPS_SA_EINVAL, "null pointer");
    if (npiv > 0x7fffffffull) return fail(CAPS_SA_EINVAL, "too many pivots");
    if (int rc = check_positions(X, cnt, n, "suffix position out of range")) return rc;
    if (int rc = check_positions(piv, npiv, n, "pivot position out of range")) return rc;
    DeviceScope restore_device_;
    if (int rc = set_device(device)) return rc;
    return guarded([&]() -> int {
        Backend be(nullptr);
        DevAllocs da(be);
        DevText t = upload_text(be, da, T, n);
        ElemBuf<idx_t> a = elem_buf<idx_t>(da, cnt), pv = elem_buf<idx_t>(da, npiv);
        idx_t* Pm = da.get<idx_t>(npiv + 2);
        be.h2d(a.sa, X, cnt * sizeof(idx_t));
        be.h2d(pv.sa, piv, npiv * sizeof(idx_t));
        SegBufs s = one_segment<idx_t>(be, da, cnt);
        const uint32_t np = (uint32_t)npiv, bpr = (np + 255) / 256;
        const uint32_t g1 = (uint32_t)((cnt + 255) / 256), g2 = (uint32_t)((npiv + 255) / 256);
        if (t.bits == 2) {
            if (cnt) CAPS_LAUNCH((make_keys_kernel<idx_t, 2>), g1, 256, be, (const uint32_t*)t.P, (const idx_t*)a.sa, cnt, a.key);
            CAPS_LAUNCH((make_keys_kernel<idx_t, 2>), g2, 256, be, (const uint32_t*)t.P, (const idx_t*)pv.sa, npiv, pv.key);
            CAPS_LAUNCH((locate_kernel<idx_t, 2>), bpr, 256, be, (const uint32_t*)t.P, n, (const uint64_t*)s.seg_start, 1u,
                        (const uint64_t*)a.key, (const idx_t*)a.sa, (const uint64_t*)pv.key, (const idx_t*)pv.sa, np, Pm);
        } else {
            if (cnt) CAPS_LAUNCH((make_keys_kernel<idx_t, 8>), g1, 256, be, (const uint32_t*)t.P, (const idx_t*)a.sa, cnt, a.key);
            CAPS_LAUNCH((make_keys_kernel<idx_t, 8>), g2, 256, be, (const uint32_t*)t.P, (const idx_t*)pv.sa, npiv, pv.key);
            CAPS_LAUNCH((locate_kernel<idx_t, 8>), bpr, 256, be, (const uint32_t*)t.P, n, (const uint64_t*)s.seg_start, 1u,
                        (const uint64_t*)a.key, (const idx_t*)a.sa, (const uint64_t*)pv.key, (const idx_t*)pv.sa, np, Pm);
        }
        be.d2h(out, Pm + 1, npiv * sizeof(idx_t));
        be.sync();
        return CAPS_SA_OK;
    });
}

template <typename idx_t>
int lcp_pairs(const char* T, uint64_t n, const idx_t* a, const idx_t* b, uint64_t cnt, idx_t* out, int device)
{
    if (int rc = check_common<idx_t>(T, n, 0)) return rc;
    if (cnt == 0) return CAPS_SA_OK;
    if (!a || !b || !out) return fail(CAPS_SA_EINVAL, "null pointer");
    if (int rc = check_positions(a, cnt, n, "suffix position out of range")) return rc;
    if (int rc = check_positions(b, cnt, n, "suffix position out of range")) return rc;
    DeviceScope restore_device_;
    if (int rc = set_device(device)) return rc;
    return guarded([&]() -> int {
        Backend be(nullptr);
        DevAllocs da(be);
        DevText t = upload_text(be, da, T, n);
        idx_t* da_ = da.get<idx_t>(cnt);
        idx_t* db_ = da.get<idx_t>(cnt);
        idx_t* dout = da.get<idx_t>(cnt);
        be.h2d(da_, a, cnt * sizeof(idx_t));
        be.h2d(db_, b, cnt * sizeof(idx_t));
        const uint32_t g = (uint32_t)((cnt + 255) / 256);
        if (t.bits == 2)
            CAPS_LAUNCH((lcp_pairs_kernel<idx_t, 2>), g, 256, be, (const uint32_t*)t.P, n, (const idx_t*)da_, (const idx_t*)db_, cnt, dout);
        else
            CAPS_LAUNCH((lcp_pairs_kernel<idx_t, 8>), g, 256, be, (const uint32_t*)t.P, n, (const idx_t*)da_, (const idx_t*)db_, cnt, dout);
        be.d2h(out, dout, cnt * sizeof(idx_t));
        be.sync();
        return CAPS_SA_OK;
    });
}

}  // namespace caps

// ---------------------------------------------------------------------------------------
extern "C" {

const char* CAPS_API(last_error)(void) { return caps::last_error_ref().c_str(); }
const char* CAPS_API(version)(void) { return "caps-sa_amd 0.2 (gfx950)"; }
uint32_t CAPS_API(stats_bytes)(void) { return (uint32_t)sizeof(caps_sa_stats); }
uint32_t CAPS_API(shard_info_bytes)(void) { return (uint32_t)sizeof(caps_sa_shard_info); }

void CAPS_API(release_cache)(void)
{
    caps::HostPathCache& hc = caps::host_cache();
    std::lock_guard<std::mutex> lock(hc.mu);
    caps::DeviceScope restore_device_;
    if (hc.device >= 0 && caps::set_device(hc.device) != CAPS_SA_OK) return;
    caps::release_host_cache_locked(hc);
}

int CAPS_API(gen_rand_seq)(uint32_t seed, uint64_t n, char* out_)
{
    if (!out_ && n) return caps::fail(CAPS_SA_EINVAL, "null pointer");
    char* __restrict__ out = out_;
    // MT19937 (Matsumoto & Nishimura): init_genrand(19650218), then init_by_array with the one-word key {seed} -- what
    // CPython's random.seed(int) does for 0 <= seed < 2^32
    constexpr int N = 624, M = 397;
    alignas(64) uint32_t mt[N];
    mt[0] = 19650218u;
    for (int i = 1; i < N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    {
        int i = 1;
        for (int k = N; k; --k) {                                   // key length 1: j stays 0
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + seed;
            if (++i >= N) { mt[0] = mt[N - 1]; i = 1; }
        }
        for (int k = N - 1; k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= N) { mt[0] = mt[N - 1]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    // a block of 624 outputs at a time: the recurrence in three loops without wrap-around (the first two vectorise: their
    // dependence distance is 227), tempering into a buffer, then a branch-free compaction of the accepted draws
    auto twist = [](uint32_t a, uint32_t b, uint32_t c) {
        const uint32_t y = (a & 0x80000000u) | (b & 0x7FFFFFFFu);
        return c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908B0DFu);
    };
    alignas(64) uint8_t k3[N + 8];
    char tail[N + 8];
    uint64_t done = 0;
    while (done < n) {
        for (int k = 0; k < N - M; ++k) mt[k] = twist(mt[k], mt[k + 1], mt[k + M]);
        for (int k = N - M; k < N - 1; ++k) mt[k] = twist(mt[k], mt[k + 1], mt[k + M - N]);
        mt[N - 1] = twist(mt[N - 1], mt[0], mt[M - 1]);
        for (int k = 0; k < N; ++k) {
            uint32_t y = mt[k];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9D2C5680u;
            y ^= (y << 15) & 0xEFC60000u;
            y ^= y >> 18;
            k3[k] = (uint8_t)(y >> 29);                             // getrandbits(3); a draw >= 4 is rejected
        }
        char* __restrict__ o = n - done >= (uint64_t)N ? out + done : tail;      // room for a whole block: straight into the output
        uint32_t c = 0;
        for (int k = 0; k < N; ++k) { const uint8_t v = k3[k]; o[c] = "ACGTACGT"[v]; c += v < 4 ? 1u : 0u; }
        if (o == tail) { if (c > n - done) c = (uint32_t)(n - done); std::memcpy(out + done, tail, c); }
        done += c;
    }
    return CAPS_SA_OK;
}

void* CAPS_API(host_alloc)(uint64_t bytes)
{
    void* p = nullptr;
    caps::guarded([&]() -> int { p = caps::Backend::host_alloc(bytes); return CAPS_SA_OK; });
    return p;
}

void CAPS_API(host_free)(void* p) { caps::Backend::host_free(p); }

int CAPS_API(workspace_bytes_ex)(uint64_t n, uint64_t subproblem_count, int idx_bytes, int bits_per_char, uint64_t* bytes)
{
    if (!bytes || (idx_bytes != 4 && idx_bytes != 8) || (bits_per_char != 2 && bits_per_char != 8)) return caps::fail(CAPS_SA_EINVAL, "bad argument");
    *bytes = (idx_bytes == 4 ? caps::make_plan<uint32_t>(n, subproblem_count, nullptr, bits_per_char).bytes
                             : caps::make_plan<uint64_t>(n, subproblem_count, nullptr, bits_per_char).bytes) + 256;
    return CAPS_SA_OK;
}
int CAPS_API(workspace_bytes)(uint64_t n, uint64_t subproblem_count, int idx_bytes, uint64_t* bytes)
{
    return CAPS_API(workspace_bytes_ex)(n, subproblem_count, idx_bytes, 8, bytes);
}

#define CAPS_DEFINE_WIDTH(SFX, IDX)                                                                                        \
    int CAPS_API(build_##SFX)(const char* T, uint64_t n, uint64_t p, uint64_t ctx, IDX* SA, IDX* LCP, int device,          \
                              caps_sa_stats* st)                                                                           \
    { return caps::build_host<IDX>(T, n, p, ctx, SA, LCP, device, st); }                                                   \
    int CAPS_API(build_multi_##SFX)(const char* T, uint64_t n, uint64_t p, uint64_t ctx, IDX* SA, IDX* LCP,                \
                                    const int* devices, int n_devices, caps_sa_stats* st)                                  \
    { return caps::build_multi<IDX>(T, n, p, ctx, SA, LCP, devices, n_devices, st); }                                      \
    int CAPS_API(build_device_##SFX)(const void* dT, uint64_t n, uint64_t p, uint64_t ctx, void* dSA, void* dLCP,          \
                                     void* ws, uint64_t ws_bytes, void* stream, caps_sa_stats* st)                         \
    { return caps::build_device<IDX>(dT, n, p, ctx, dSA, dLCP, ws, ws_bytes, stream, st); }                                \
    int CAPS_API(verify_device_##SFX)(const void* dT, uint64_t n, const void* dSA, const void* dLCP, void* stream,          \
                                      uint64_t* n_errors)                                                                  \
    { return caps::verify_device<IDX>(dT, n, dSA, dLCP, stream, n_errors); }                                               \
    int CAPS_API(verify_slice_device_##SFX)(const void* dT, uint64_t n, const void* dSA, const void* dLCP, uint64_t cnt,    \
                                            int is_head, void* stream, uint64_t* n_errors)                                 \
    { return caps::verify_device<IDX>(dT, n, dSA, dLCP, stream, n_errors, cnt, is_head ? 1u : 0u); }                       \
    int CAPS_API(sort_suffixes_##SFX)(const char* T, uint64_t n, const IDX* idx, uint64_t cnt, IDX* osa, IDX* olcp, int d) \
    { return caps::sort_suffixes<IDX>(T, n, idx, cnt, osa, olcp, d); }                                                     \
    int CAPS_API(sort_segments_##SFX)(const char* T, uint64_t n, const IDX* idx, uint64_t cnt, const uint64_t* seg,        \
                                      uint64_t G, IDX* osa, IDX* olcp, int d)                                              \
    { return caps::sort_segments<IDX>(T, n, idx, cnt, seg, G, osa, olcp, d); }                                             \
    int CAPS_API(merge_##SFX)(const char* T, uint64_t n, const IDX* X, uint64_t lx, const IDX* Y, uint64_t ly,             \
                              const IDX* LX, const IDX* LY, IDX* Z, IDX* LZ, int d)                                        \
    { return caps::merge_runs<IDX>(T, n, X, lx, Y, ly, LX, LY, Z, LZ, d); }                                                \
    int CAPS_API(upper_bound_##SFX)(const char* T, uint64_t n, const IDX* X, uint64_t cnt, const IDX* piv, uint64_t np,    \
                                    IDX* out, int d)                                                                       \
    { return caps::upper_bounds<IDX>(T, n, X, cnt, piv, np, out, d); }                                                     \
    int CAPS_API(lcp_##SFX)(const char* T, uint64_t n, const IDX* a, const IDX* b, uint64_t cnt, IDX* out, int d)          \
    { return caps::lcp_pairs<IDX>(T, n, a, b, cnt, out, d); }

CAPS_DEFINE_WIDTH(u32, uint32_t)
CAPS_DEFINE_WIDTH(u64, uint64_t)


struct caps_sa_shard { std::unique_ptr<caps::ShardBase> impl; };

caps_sa_shard* caps_shard_new(const void* dT, uint64_t n, uint64_t p, int idx_bytes, int rank, int world, void* stream)
{
    caps_sa_shard* s = new caps_sa_shard;
    try {
        if (idx_bytes == 4) s->impl.reset(new caps::Shard<uint32_t>(dT, n, p, rank, world, stream));
        else s->impl.reset(new caps::Shard<uint64_t>(dT, n, p, rank, world, stream));
    } catch (...) { delete s; throw; }
    return s;
}

int CAPS_API(shard_create)(const void* dT, uint64_t n, uint64_t p, int idx_bytes, int rank, int world, void* stream,
                           caps_sa_shard** out)
{
    if (!out || !dT || (idx_bytes != 4 && idx_bytes != 8) || world < 1 || rank < 0 || rank >= world)
        return caps::fail(CAPS_SA_EINVAL, "bad argument");
    if (idx_bytes == 4 && n > 0xFFFFFFFFull) return caps::fail(CAPS_SA_EINVAL, "n does not fit 32-bit indices");
    *out = nullptr;
    return caps::guarded([&]() -> int { *out = caps_shard_new(dT, n, p, idx_bytes, rank, world, stream); return CAPS_SA_OK; });
}
void CAPS_API(shard_destroy)(caps_sa_shard* s) { delete s; }
int CAPS_API(shard_info)(const caps_sa_shard* s, caps_sa_shard_info* info)
{
    if (!s || !info) return caps::fail(CAPS_SA_EINVAL, "null pointer");
    s->impl->info(info);
    return CAPS_SA_OK;
}
int CAPS_API(shard_phase1)(caps_sa_shard* s, void* k, void* a)
{
    if (!s) return caps::fail(CAPS_SA_EINVAL, "null shard");
    return caps::guarded([&]() -> int { s->impl->phase1(k, a); return CAPS_SA_OK; });
}
int CAPS_API(shard_pivots)(caps_sa_shard* s, const void* k, const void* a, void* sizes)
{
    if (!s || !k || !a || !sizes) return caps::fail(CAPS_SA_EINVAL, "null pointer");
    return caps::guarded([&]() -> int { s->impl->pivots(k, a, sizes); return CAPS_SA_OK; });
}
int CAPS_API(shard_collate)(caps_sa_shard* s, const uint64_t* all_sizes, void* k, void* a, uint64_t* sc, uint64_t* rc)
{
    if (!s || !all_sizes || !sc || !rc) return caps::fail(CAPS_SA_EINVAL, "null pointer");
    return caps::guarded([&]() -> int { s->impl->collate(all_sizes, k, a, sc, rc); return CAPS_SA_OK; });
}
int CAPS_API(shard_phase2)(caps_sa_shard* s, const void* k, const void* a, void* dSA, void* dLCP)
{
    if (!s) return caps::fail(CAPS_SA_EINVAL, "null shard");
    return caps::guarded([&]() -> int { s->impl->phase2(k, a, dSA, dLCP); return CAPS_SA_OK; });
}
int CAPS_API(shard_scatter)(caps_sa_shard* s, void* k, void* a, void* report)
{
    if (!s || !k || !a || !report) return caps::fail(CAPS_SA_EINVAL, "null pointer");
    return caps::guarded([&]() -> int { s->impl->scatter(k, a, report); return CAPS_SA_OK; });
}
int CAPS_API(shard_plan)(caps_sa_shard* s, const uint64_t* all_reports, uint64_t* sc, uint64_t* rc)
{
    if (!s || !all_reports || !sc || !rc) return caps::fail(CAPS_SA_EINVAL, "null pointer");
    return caps::guarded([&]() -> int { return s->impl->plan(all_reports, sc, rc); });
}
int CAPS_API(shard_sort)(caps_sa_shard* s, const void* k, const void* a, void* dSA, void* dLCP)
{
    if (!s) return caps::fail(CAPS_SA_EINVAL, "null shard");
    return caps::guarded([&]() -> int { return s->impl->sort_owned(k, a, dSA, dLCP); });
}
int CAPS_API(shard_set_key_bits)(caps_sa_shard* s, int bits)
{
    if (!s || (bits != 32 && bits != 64)) return caps::fail(CAPS_SA_EINVAL, "bad argument");
    s->impl->set_key_bits(bits);
    return CAPS_SA_OK;
}
int CAPS_API(shard_phase1_arrays)(caps_sa_shard* s, void* d_keys_out, void* d_sa_out, uint64_t* count, uint64_t* subarray_len)
{
    if (!s || !count || !subarray_len) return caps::fail(CAPS_SA_EINVAL, "null pointer");
    return caps::guarded([&]() -> int { s->impl->phase1_arrays(d_keys_out, d_sa_out, count, subarray_len); return CAPS_SA_OK; });
}
int CAPS_API(shard_last_sa)(caps_sa_shard* s, uint64_t* last_sa)
{
    if (!s || !last_sa) return caps::fail(CAPS_SA_EINVAL, "null pointer");
    return caps::guarded([&]() -> int { *last_sa = s->impl->last_sa(); return CAPS_SA_OK; });
}
int CAPS_API(shard_fix_first_lcp)(caps_sa_shard* s, uint64_t prev_sa, void* dLCP)
{
    if (!s) return caps::fail(CAPS_SA_EINVAL, "null shard");
    return caps::guarded([&]() -> int { s->impl->fix_first_lcp(prev_sa, dLCP); return CAPS_SA_OK; });
}

}  // extern "C"

#if defined(CAPS_PHASE_CLOCK) && !defined(CAPS_EMUL)
// measurement builds only (kernels.h PHASE_MARK): copies the 32 phase counters out and clears them
extern "C" __attribute__((visibility("default"))) int caps_sa_hip_phase_clock(uint64_t* out32)
{
    unsigned long long h[32];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(caps::caps_phase_clock), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 32; ++i) out32[i] = h[i];
    for (int i = 0; i < 32; ++i) h[i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(caps::caps_phase_clock), h, sizeof(h)) == hipSuccess ? 0 : -1;
}
#endif

#if defined(CAPS_EQ_CHECK) && !defined(CAPS_EMUL)
// debugging builds only (kernels.h EQ_OK): copies the 64 check counters out and clears them
extern "C" __attribute__((visibility("default"))) int caps_sa_hip_eq_check(uint32_t* out64)
{
    unsigned int h[64];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(caps::caps_eq_check), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 64; ++i) out64[i] = h[i];
    for (int i = 0; i < 64; ++i) h[i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(caps::caps_eq_check), h, sizeof(h)) == hipSuccess ? 0 : -1;
}
#endif
