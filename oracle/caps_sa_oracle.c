/*
 * oracle/caps_sa_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99 + OpenMP) of the hot path of jamshed/CaPS-SA:
 * samplesort over suffixes with LCP-merge (reference: src/Suffix_Array.cpp,
 * include/Suffix_Array.hpp; every function cites the lines it follows).
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg -- only as the checker / the reported CPU baseline, never as the thing
 * shipped or measured as the product.  Nothing under caps-sa_amd/ links, loads
 * or calls this file.
 *
 * Parity pinning: the reference itself cannot be built in this image (its only
 * dependency, parlaylib, is fetched by git at configure time, CMakeLists.txt:89-98,
 * and is not on disk; writing a stand-in header is not allowed).  The oracle is
 * therefore pinned against (i) the reference's own correctness tool
 * chatgpt_baseline.py, imported in the dev container to generate the fixtures in
 * tests/golden/ (script: tests/golden/make_golden.py), (ii) the reference's data
 * file data/simpletest2, and (iii) the sha256 digests of dump files the real
 * reference produced during the survey session, see
 * tests/test_oracle_pins.py.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <ctype.h>
#include <omp.h>

/* nested_par_grain_size, include/Suffix_Array.hpp:43 */
#define CAPS_ORACLE_GRAIN 8192u

/* ---- a2: longest common prefix of two byte strings, at most min_len ----------
 * Follows LCP<8>/LCP_unrolled (include/Suffix_Array.hpp:195-241): compare wide
 * blocks while at least a block remains, finish with narrower steps, never read
 * past min_len.  The reference uses 32-byte AVX2 compares; here 8-byte words and
 * a count-trailing-zeros on the XOR do the same job portably.                   */
uint64_t caps_oracle_lcp(const char *x, const char *y, uint64_t min_len)
{
    uint64_t l = 0;
    while (min_len - l >= 8) {
        uint64_t a, b;
        memcpy(&a, x + l, 8);
        memcpy(&b, y + l, 8);
        const uint64_t d = a ^ b;
        if (d) return l + (uint64_t)(__builtin_ctzll(d) >> 3);   /* little-endian host */
        l += 8;
    }
    while (l < min_len && x[l] == y[l]) ++l;                     /* hpp:201-208, N == 1 */
    return l;
}

/* ---- CLI byte remap (src/main.cpp:61-70): every byte -> {A,C,T,G} ------------- */
void caps_oracle_remap(char *text, uint64_t n)
{
    static const char lookup[4] = { 'A', 'C', 'T', 'G' };        /* main.cpp:61 */
    for (uint64_t i = 0; i < n; ++i)
        text[i] = lookup[(toupper((unsigned char)text[i]) & 0x6) >> 1]; /* main.cpp:67-68 */
}

int caps_oracle_max_threads(void) { return omp_get_max_threads(); }

#define IDX uint32_t
#define SFX _u32
#include "caps_sa_oracle_impl.inc"
#undef IDX
#undef SFX

#define IDX uint64_t
#define SFX _u64
#include "caps_sa_oracle_impl.inc"
#undef IDX
#undef SFX
