// tests/emul/emul_backend.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Host stand-in for hip_backend.h so that the kernel sources of caps-sa_amd/csrc
// (kernels.h, pipeline.h, capi_impl.h) can be compiled with g++ under -DCAPS_EMUL and
// their LOGIC exercised in the GPU-less development container.  "Device memory" is host
// memory, a launch is a loop over blocks, a phase (PAR) is a loop over threads.  The
// library built from this (tests/emul/libcaps_sa_emul.so) exports caps_sa_emul_* symbols;
// the product library never links or loads it, and the -m gpu parity tests never use it.
#pragma once
#define CAPS_BACKEND_DEFINED 1
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace caps {

struct HipError : std::runtime_error {
    explicit HipError(const std::string& m) : std::runtime_error(m) {}
};
struct OomError : std::runtime_error {
    explicit OomError(const std::string& m) : std::runtime_error(m) {}
};

struct BackendEvent { double t = 0; };
struct DeviceScope { DeviceScope() {} ~DeviceScope() {} };

class Backend {
public:
    void* stream = nullptr;
    bool long_runs = false;       // the text prepared last holds a periodic stretch >= RUN_LONG chars (text.h): comparators with the run table
    explicit Backend(void* s) : stream(s) {}
    BackendEvent record()
    {
        using namespace std::chrono;
        return BackendEvent{duration<double, std::milli>(steady_clock::now().time_since_epoch()).count()};
    }
    double elapsed_ms(BackendEvent a, BackendEvent b) { return b.t - a.t; }
    void release_events() {}
    void* alloc(size_t bytes)
    {
        void* p = std::malloc(bytes ? bytes : 1);
        if (!p) throw OomError("malloc");
        std::memset(p, 0xCD, bytes);          // poison: catch reads of unwritten workspace
        return p;
    }
    void free(void* p) { std::free(p); }
    static void* host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
    static void host_free(void* p) { std::free(p); }
    void memset(void* d, int v, size_t bytes) { std::memset(d, v, bytes); }
    void h2d(void* d, const void* h, size_t bytes) { std::memcpy(d, h, bytes); }
    void d2h(void* h, const void* d, size_t bytes) { std::memcpy(h, d, bytes); }
    void d2d(void* d, const void* s, size_t bytes) { std::memmove(d, s, bytes); }
    void sync() {}
    static void* create_stream() { return nullptr; }
    static void destroy_stream(void*) {}
    static void stream_wait(void*, BackendEvent) {}
    static void d2h_on(void*, void* h, const void* d, size_t bytes) { std::memcpy(h, d, bytes); }
    static void sync_stream(void*) {}
    static BackendEvent record_on(void*) { return BackendEvent{}; }
    static void wait_event(BackendEvent) {}
    static void destroy_event(BackendEvent) {}
    static void memset_on(void*, void* d, int v, size_t bytes) { std::memset(d, v, bytes); }
    void peer_copy(void* dst, int, const void* src, int, size_t bytes) { std::memmove(dst, src, bytes); }
    static void peer_copy_on(void*, void* dst, int, const void* src, int, size_t bytes) { std::memmove(dst, src, bytes); }
    static void h2d_on(void*, void* d, const void* h, size_t bytes) { std::memcpy(d, h, bytes); }
    static void enable_peer(int, int) {}
    uint32_t persistent_blocks() { return 3; }     // small on purpose: exercises the tile loop
    void check_launch(const char*) {}
};

// grid of a kernel that loops over its logical blocks: small on purpose, so that the loops run
inline uint32_t capped_grid(uint64_t want, uint32_t) { return (uint32_t)(want < 5 ? (want ? want : 1) : 5); }

template <typename F> inline void emul_launch(uint32_t grid, uint32_t block, F&& body)
{
    for (uint32_t b = 0; b < grid; ++b) {
        EmulCtx c{b, grid, block};
#ifdef CAPS_EMUL_RACE
        caps_race::barrier();                     // every workgroup starts its own epochs, outside any PAR region
#endif
        body(c);
    }
}

}  // namespace caps

#define CAPS_LAUNCH(kernel, grid, block, be, ...)                                                     \
    ::caps::emul_launch((uint32_t)(grid), (uint32_t)(block), [&](const ::caps::EmulCtx& c_) { kernel(c_, __VA_ARGS__); })
