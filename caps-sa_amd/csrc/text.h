// caps-sa_amd/csrc/text.h
//
// Device text representation and the suffix comparator (the GPU counterpart of
// the reference's LCP<8> comparator, include/Suffix_Array.hpp:195-241, and of the
// "shorter suffix first" rule, src/Suffix_Array.cpp:75-77).
//
// Text layout in HBM ("packed text" P): chars are stored as BITS-wide codes, first
// char in the MOST significant bits of each 32-bit word, so that comparing words as
// unsigned integers is comparing strings.  Codes preserve the reference's signed-char
// order (Suffix_Array.cpp:77,289):
//   BITS = 2 : alphabets of <= 4 distinct bytes (always true behind the reference CLI,
//              src/main.cpp:61-70); code = rank of the byte in signed-char order.
//   BITS = 8 : any bytes; code = byte ^ 0x80.
// Everything past the end of the text is code 0, which is <= every real code; a
// suffix that is a prefix of another one therefore never compares greater, and the
// exact "one is a prefix of the other" case is resolved by text positions.
//
// Every suffix carries a 64-bit KEY = its first 64/BITS chars (32 bases for DNA).
// Keys travel with the suffix indices through every pass, so a comparison touches
// the text only when two keys are equal (never, for all practical purposes, on
// random DNA; constantly on repeats, where deep_lcp() continues in 64-bit windows).
#pragma once
#include "kernel_lang.h"

namespace caps {

template <int BITS> struct TextTraits {
    static constexpr uint32_t CPW = 32 / BITS;     // chars per 32-bit word
    static constexpr uint32_t KCH = 64 / BITS;     // chars per 64-bit key/window
};

// Number of 32-bit words of the packed text for n chars (3 readable words past the
// last text word are needed by window64; 4 are kept).
HD uint64_t packed_words(uint64_t n, int bits) { return (n + (32 / bits) - 1) / (32 / bits) + 4; }

// 64-bit window of chars [pos, pos + KCH) (pos < n).
template <int BITS>
HD uint64_t window64(const uint32_t* __restrict__ P, uint64_t pos)
{
    constexpr uint32_t CPW = TextTraits<BITS>::CPW;
    const uint64_t w = pos / CPW;
    const uint32_t sh = (uint32_t)(pos % CPW) * BITS;          // 0..31
    const uint64_t x0 = P[w], x1 = P[w + 1], x2 = P[w + 2];
    return (((x0 << 32) | x1) << sh) | ((x2 << sh) >> 32);
}

// lcp(suffix a, suffix b) given that their first `from` chars are known equal
// (from is a multiple of KCH or anything <= the true lcp).  Exact, clamped to the
// length of the shorter suffix.
template <int BITS>
HD uint64_t deep_lcp(const uint32_t* __restrict__ P, uint64_t n, uint64_t a, uint64_t b, uint64_t from)
{
    constexpr uint32_t KCH = TextTraits<BITS>::KCH;
    if (a >= n || b >= n) return 0;              // corrupt index: do not scan
    const uint64_t maxlen = n - (a > b ? a : b);
    uint64_t l = from;
    while (l < maxlen) {
        const uint64_t x = window64<BITS>(P, a + l) ^ window64<BITS>(P, b + l);
        if (x) { l += (uint32_t)caps_clz64(x) / BITS; break; }
        l += KCH;
    }
    return l < maxlen ? l : maxlen;
}

// lcp of two distinct suffixes from their keys, falling back to the text when the
// keys are equal.
template <int BITS>
HD uint64_t pair_lcp(const uint32_t* __restrict__ P, uint64_t n, uint64_t ka, uint64_t a, uint64_t kb, uint64_t b)
{
    const uint64_t x = ka ^ kb;
    if (x) {
        const uint64_t maxlen = n - (a > b ? a : b);
        const uint64_t l = (uint32_t)caps_clz64(x) / BITS;
        return l < maxlen ? l : maxlen;
    }
    return deep_lcp<BITS>(P, n, a, b, TextTraits<BITS>::KCH);
}

// Order of two distinct suffixes whose KEYS are equal: continue in 64-bit windows of the
// packed text (the rare path on low-LCP texts, the common one on repeats).  Inlined: a call
// inside the merge kernels would force the registers that hold the prefetched next tile to
// be spilled for the whole rank phase.
template <int BITS>
HD bool suffix_less_tie(const uint32_t* __restrict__ P, uint64_t n, uint64_t a, uint64_t b)
{
    constexpr uint32_t KCH = TextTraits<BITS>::KCH;
    if (a >= n || b >= n) return a > b;          // never loop on a corrupt index (keeps a bad input from hanging the GPU)
    const uint64_t maxlen = n - (a > b ? a : b);
    for (uint64_t l = KCH; l < maxlen; l += KCH) {
        const uint64_t wa = window64<BITS>(P, a + l), wb = window64<BITS>(P, b + l);
        if (wa != wb) return wa < wb;
    }
    return a > b;
}

// Strict total order on suffixes: true iff suffix a sorts before suffix b.
// (a == b -> false.)  Shorter suffix first when one is a prefix of the other.
template <int BITS>
HD bool suffix_less(const uint32_t* __restrict__ P, uint64_t n, uint64_t ka, uint64_t a, uint64_t kb, uint64_t b)
{
    if (ka != kb) return ka < kb;
    if (a == b) return false;
    return suffix_less_tie<BITS>(P, n, a, b);
}

}  // namespace caps
