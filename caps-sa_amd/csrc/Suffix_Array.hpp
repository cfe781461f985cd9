// caps-sa_amd/csrc/Suffix_Array.hpp
//
// Host-side C++ mirror of the reference's class surface CaPS_SA::Suffix_Array<T_idx_>
// (reference include/Suffix_Array.hpp:148-181): same constructor arguments, same
// accessors, same dump format -- but construct() hands the whole samplesort to the
// MI355X through the C ABI of include/caps_sa_hip.h (libcaps_sa_hip.so).  A program
// written against the reference header (e.g. its CLI, src/main.cpp:78-80) compiles
// against this one unchanged.
//
// Differences, all deliberate:
//  * errors: the reference calls std::exit (src/Suffix_Array.cpp:33-37); construct()
//    here throws std::runtime_error carrying caps_sa_hip_last_error();
//  * valid for every n >= 0 (the reference divides by zero for n < 32 / p_eff < 2);
//  * bounded max_context (0 < max_context < n) runs the reference's own merge sequence on ONE device (csrc/bounded.h);
//  * SA_ / LCP_ are page-locked (caps_sa_hip_host_alloc) when the driver grants it, plain malloc otherwise: the
//    results then leave the GPU at the PCIe link rate (C2: 38 ms instead of 88 ms for the two arrays) -- like the
//    reference's mallocs (src/Suffix_Array.cpp:20-21) the allocation belongs to the constructor, not to construct();
//  * an optional list of devices: construct() then shards the build over them (caps_sa_hip_build_multi_*).
#ifndef CAPS_SA_AMD_SUFFIX_ARRAY_HPP
#define CAPS_SA_AMD_SUFFIX_ARRAY_HPP

#include <cstdint>
#include <cstddef>
#include <cstdlib>
#include <fstream>
#include <new>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/caps_sa_hip.h"

namespace CaPS_SA
{

template <typename T_idx_>
class Suffix_Array
{
    static_assert(std::is_same<T_idx_, uint32_t>::value || std::is_same<T_idx_, uint64_t>::value,
                  "instantiated for uint32_t and uint64_t like the reference (src/Suffix_Array.cpp:543-544)");

public:
    typedef T_idx_ idx_t;

    // Reference: include/Suffix_Array.hpp:155, src/Suffix_Array.cpp:16-38.  T is borrowed
    // and must outlive the object; SA/LCP are allocated here and freed by the destructor.
    Suffix_Array(const char* T, idx_t n, idx_t subproblem_count = 0, idx_t max_context = 0, int device = 0)
        : Suffix_Array(T, n, subproblem_count, max_context, std::vector<int>(1, device)) {}

    // devices: HIP device ordinals the build is sharded over (one: the single-GPU build)
    Suffix_Array(const char* T, idx_t n, idx_t subproblem_count, idx_t max_context, const std::vector<int>& devices)
        : T_(T), n_(n), SA_(nullptr), LCP_(nullptr), pinned_(false),
          subproblem_count_(subproblem_count), max_context_(max_context), devices_(devices), stats_()
    {
        if (devices_.empty()) throw std::invalid_argument("Suffix_Array: no devices");
        const std::size_t bytes = (n ? static_cast<std::size_t>(n) : 1) * sizeof(idx_t);
        SA_ = static_cast<idx_t*>(caps_sa_hip_host_alloc(bytes));
        LCP_ = SA_ ? static_cast<idx_t*>(caps_sa_hip_host_alloc(bytes)) : nullptr;
        pinned_ = SA_ && LCP_;
        if (!pinned_) {                                       // no GPU / no page-locked memory to be had: pageable
            if (SA_) caps_sa_hip_host_free(SA_);
            SA_ = static_cast<idx_t*>(std::malloc(bytes));
            LCP_ = static_cast<idx_t*>(std::malloc(bytes));
            if (!SA_ || !LCP_) { std::free(SA_); std::free(LCP_); throw std::bad_alloc(); }
        }
    }

    Suffix_Array(const Suffix_Array&) = delete;                 // hpp:157-160
    Suffix_Array& operator=(const Suffix_Array&) = delete;
    Suffix_Array(Suffix_Array&&) = delete;
    Suffix_Array& operator=(Suffix_Array&&) = delete;

    ~Suffix_Array()                                             // cpp:41-45
    {
        if (pinned_) { caps_sa_hip_host_free(SA_); caps_sa_hip_host_free(LCP_); }
        else { std::free(SA_); std::free(LCP_); }
    }

    const char* T() const { return T_; }                        // hpp:165
    idx_t n() const { return n_; }                              // hpp:168
    const idx_t* SA() const { return SA_; }                     // hpp:171
    const idx_t* LCP() const { return LCP_; }                   // hpp:174

    // Reference: src/Suffix_Array.cpp:466-494.  May be called more than once.
    void construct()
    {
        int rc;
        const int nd = static_cast<int>(devices_.size());
        if (std::is_same<idx_t, uint32_t>::value)
            rc = caps_sa_hip_build_multi_u32(T_, n_, subproblem_count_, max_context_, reinterpret_cast<uint32_t*>(SA_),
                                             reinterpret_cast<uint32_t*>(LCP_), devices_.data(), nd, &stats_);
        else
            rc = caps_sa_hip_build_multi_u64(T_, n_, subproblem_count_, max_context_, reinterpret_cast<uint64_t*>(SA_),
                                             reinterpret_cast<uint64_t*>(LCP_), devices_.data(), nd, &stats_);
        if (rc != CAPS_SA_OK)
            throw std::runtime_error(std::string("caps_sa_hip_build: ") + caps_sa_hip_last_error());
    }

    // Reference: src/Suffix_Array.cpp:497-509 -- u64 n, then SA, then LCP, native endianness.
    void dump(std::ofstream& output)
    {
        const std::size_t n = n_;
        output.write(reinterpret_cast<const char*>(&n), sizeof(std::size_t));
        output.write(reinterpret_cast<const char*>(SA_), n * sizeof(idx_t));
        output.write(reinterpret_cast<const char*>(LCP_), n * sizeof(idx_t));
    }

    // Per-phase record of the last construct() (replaces the reference's stderr timing lines).
    const caps_sa_stats& stats() const { return stats_; }

private:
    const char* const T_;
    const idx_t n_;
    idx_t* SA_;
    idx_t* LCP_;
    bool pinned_;
    const idx_t subproblem_count_;
    const idx_t max_context_;
    const std::vector<int> devices_;
    caps_sa_stats stats_;
};

}  // namespace CaPS_SA

#endif
