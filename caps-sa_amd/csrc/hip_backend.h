// caps-sa_amd/csrc/hip_backend.h -- HIP runtime plumbing for pipeline.h (product build).
#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>
#include <vector>

namespace caps {

struct HipError : std::runtime_error {
    explicit HipError(const std::string& m) : std::runtime_error(m) {}
};
struct OomError : std::runtime_error {
    explicit OomError(const std::string& m) : std::runtime_error(m) {}
};

inline void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) {
        (void)hipGetLastError();                 // HIP keeps a failed call's code until it is read: the next launch check of this
                                                 // thread (check_launch) would report it as its own
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        if (e == hipErrorOutOfMemory) throw OomError(m);
        throw HipError(m);
    }
}
#define CAPS_HIP(call) ::caps::hip_check((call), #call)

struct BackendEvent { hipEvent_t ev = nullptr; };

// The host-buffer entry points select their device; the caller's current device is restored when they return.
struct DeviceScope {
    int prev = -1;
    DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

class Backend {
public:
    hipStream_t stream = nullptr;
    bool long_runs = false;       // the text prepared last holds a periodic stretch >= RUN_LONG chars (text.h): comparators with the run table
    explicit Backend(hipStream_t s) : stream(s) {}
    ~Backend() { release_events(); }

    BackendEvent record()
    {
        hipEvent_t e;
        CAPS_HIP(hipEventCreate(&e));
        pool_.push_back(e);
        CAPS_HIP(hipEventRecord(e, stream));
        return BackendEvent{e};
    }
    double elapsed_ms(BackendEvent a, BackendEvent b)
    {
        float ms = 0.f;
        CAPS_HIP(hipEventElapsedTime(&ms, a.ev, b.ev));
        return ms;
    }
    void release_events()
    {
        for (hipEvent_t e : pool_) (void)hipEventDestroy(e);
        pool_.clear();
    }
    void* alloc(size_t bytes)
    {
        void* p = nullptr;
        CAPS_HIP(hipMalloc(&p, bytes ? bytes : 1));
        return p;
    }
    void free(void* p) { if (p) (void)hipFree(p); }
    // page-locked host memory (result arrays of the class mirror: D2H at link rate instead of through pageable staging)
    static void* host_alloc(size_t bytes)
    {
        void* p = nullptr;
        CAPS_HIP(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
        return p;
    }
    static void host_free(void* p) { if (p) (void)hipHostFree(p); }
    void memset(void* d, int v, size_t bytes) { CAPS_HIP(hipMemsetAsync(d, v, bytes, stream)); }
    void h2d(void* d, const void* h, size_t bytes) { if (bytes) CAPS_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream)); }
    void d2h(void* h, const void* d, size_t bytes) { if (bytes) CAPS_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream)); }
    void d2d(void* d, const void* s, size_t bytes) { if (bytes) CAPS_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, stream)); }
    void sync() { CAPS_HIP(hipStreamSynchronize(stream)); }
    // multi-device builds (capi_impl.h build_multi): a stream on the current device; a copy between two devices' memories
    // (the same device twice is allowed: a plain device copy) -- SDMA over xGMI, no staging through the host
    static hipStream_t create_stream()
    {
        hipStream_t s = nullptr;
        CAPS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        return s;
    }
    static void destroy_stream(hipStream_t s) { if (s) (void)hipStreamDestroy(s); }
    // result copies that overlap the rest of a build (capi_impl.h build_host): stream `s` waits for an event of this backend's
    // stream, then copies device -> host
    static void stream_wait(hipStream_t s, BackendEvent e) { CAPS_HIP(hipStreamWaitEvent(s, e.ev, 0)); }
    static void d2h_on(hipStream_t s, void* h, const void* d, size_t bytes)
    {
        if (bytes) CAPS_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
    }
    static void sync_stream(hipStream_t s) { CAPS_HIP(hipStreamSynchronize(s)); }
    // an event of the caller's own on any stream (the host waits for it: build_host's widening of LCP bytes follows the copies)
    static BackendEvent record_on(hipStream_t s)
    {
        hipEvent_t e;
        CAPS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        CAPS_HIP(hipEventRecord(e, s));
        return BackendEvent{e};
    }
    static void wait_event(BackendEvent e) { CAPS_HIP(hipEventSynchronize(e.ev)); }
    static void destroy_event(BackendEvent e) { if (e.ev) (void)hipEventDestroy(e.ev); }
    static void memset_on(hipStream_t s, void* d, int v, size_t bytes) { CAPS_HIP(hipMemsetAsync(d, v, bytes, s)); }
    // the same on any stream of the SOURCE device (build_multi's fan-out of the text: one stream per destination, so the copies
    // to different peers run side by side, each over its own xGMI link)
    static void peer_copy_on(hipStream_t s, void* dst, int dst_dev, const void* src, int src_dev, size_t bytes)
    {
        if (!bytes) return;
        if (dst_dev == src_dev) CAPS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
        else CAPS_HIP(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s));
    }
    static void h2d_on(hipStream_t s, void* d, const void* h, size_t bytes)
    {
        if (bytes) CAPS_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
    }
    // direct loads / stores and SDMA copies between two devices of one process (xGMI): enabled once per ordered pair; a pair
    // without peer access keeps working (the runtime stages such copies through the host)
    static void enable_peer(int dev, int peer)
    {
        if (dev == peer) return;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dev, peer) != hipSuccess || !can) { (void)hipGetLastError(); return; }
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) return;
        if (hipSetDevice(dev) == hipSuccess) {
            const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
            if (e != hipSuccess) (void)hipGetLastError();               // (already enabled: fine)
        }
        (void)hipSetDevice(prev);
    }
    void peer_copy(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes)
    {
        if (!bytes) return;
        if (dst_dev == src_dev) CAPS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
        else CAPS_HIP(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, stream));
    }
    // Grid of the persistent kernels: as many workgroups as are resident at once
    // (CAPS_TILE_WAVES waves per SIMD = CAPS_TILE_WAVES * 256 / CAPS_TILE_NT workgroups per CU).
    uint32_t persistent_blocks()
    {
        if (!pblocks_) {
            int dev = 0, cus = 0;
            CAPS_HIP(hipGetDevice(&dev));
            CAPS_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            pblocks_ = (uint32_t)(cus > 0 ? cus : 256) * (CAPS_TILE_WAVES * 256 / CAPS_TILE_NT);
        }
        return pblocks_;
    }
    void check_launch(const char* name)
    {
        hipError_t e = hipGetLastError();
        if (e == hipErrorOutOfMemory)            // memory is committed lazily: an allocation too large can surface here
            throw OomError(std::string("launch of ") + name + ": " + hipGetErrorString(e));
        if (e != hipSuccess) throw HipError(std::string("launch of ") + name + ": " + hipGetErrorString(e));
    }

private:
    std::vector<hipEvent_t> pool_;
    uint32_t pblocks_ = 0;
};

// A dispatch holds its size in WORK-ITEMS as 32 bits (AQL grid_size_x): grid * block >= 2^32 does not fail, it
// silently runs a truncated grid (seen with locate_kernel at p = 72,845: p^2 threads).  Kernels whose natural
// grid can get there loop over their logical blocks; everything else is refused here.
inline uint32_t checked_grid(uint64_t grid, uint64_t block, const char* name)
{
    if (grid == 0 || grid * block > 0xFFFFFFFFull)
        throw HipError(std::string(name) + ": launch of " + std::to_string(grid) + " x " + std::to_string(block) +
                       " work-items does not fit a dispatch");
    return (uint32_t)grid;
}
// largest grid (in workgroups of `block` threads) of a kernel that loops over its logical blocks
inline uint32_t capped_grid(uint64_t want, uint32_t block)
{
    const uint64_t cap = 0xFFFFFFFFull / block;
    return (uint32_t)(want < cap ? (want ? want : 1) : cap);
}

}  // namespace caps

// Launch on the backend's stream.  Host code checks operand shapes before every launch
// (grid sizes are derived from the same n/p that size the buffers).
#define CAPS_LAUNCH(kernel, grid, block, be, ...)                                                        \
    do {                                                                                                 \
        hipLaunchKernelGGL(kernel, dim3(::caps::checked_grid((grid), (block), #kernel)), dim3((uint32_t)(block)), 0, (be).stream, \
                           __VA_ARGS__);                                                                 \
        (be).check_launch(#kernel);                                                                      \
    } while (0)
