// tests/emul/race_selftest.cpp -- TEST INFRASTRUCTURE: the race detector must see a hand-off without a barrier (and must not see
// one with it), read-after-write, write-after-read, different-value write-write, and atomics racing with plain accesses.
#define CAPS_EMUL 1
#include "../../caps-sa_amd/csrc/kernel_lang.h"

namespace {
// mode 0: neighbour exchange WITH the barrier; 1: without it (read-after-write); 2: write-after-read; 3: write-write of
// different values; 4: "every writer stores 1" (benign); 5: atomics only (clean); 6: plain read of a word others bump atomically
void kernel(const caps::EmulCtx& kctx_, int mode, uint32_t* out)
{
    SHARED_ARRAY(uint32_t, a, 64);
    SHARED_ARRAY(uint32_t, f, 2);
    PAR(tid) { a[tid] = tid * 3u; if (tid < 2) f[tid] = 0; }
    SYNC();
    if (mode == 0 || mode == 1) {
        PAR(tid) { a[tid] = tid + 100u; }
        if (mode == 0) SYNC();
        PAR(tid) { out[tid] = a[(tid + 1) % K_BLOCK_DIM]; }
    } else if (mode == 2) {
        PAR(tid) { out[tid] = a[(tid + 1) % K_BLOCK_DIM]; }
        PAR(tid) { a[tid] = 7u; }
    } else if (mode == 3) {
        PAR(tid) { f[0] = tid; }
    } else if (mode == 4) {
        PAR(tid) { if (tid & 1) f[0] = 1; }
    } else if (mode == 5) {
        PAR(tid) { FETCH_ADD_U32(&f[0], 1u); ATOMIC_MAX_U32(&f[1], tid); }
        SYNC();
        PAR(tid) { out[tid] = f[0] + f[1]; }
    } else if (mode == 6) {
        PAR(tid) { FETCH_ADD_U32(&f[0], 1u); out[tid] = f[0]; }
    }
}
}  // namespace

extern "C" unsigned long long caps_sa_emul_races_found(void);
extern "C" void caps_sa_emul_races_reset(void);
extern "C" unsigned long long caps_race_selftest(int mode)
{
    uint32_t out[64];
    caps_sa_emul_races_reset();
    caps::EmulCtx c{0, 1, 64};
    caps_race::barrier();
    kernel(c, mode, out);
    const unsigned long long r = caps_sa_emul_races_found();
    caps_sa_emul_races_reset();
    return r;
}
