// tests/emul/emul_lib.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// Host emulation of the kernels in caps-sa_amd/csrc (see emul_backend.h, kernel_lang.h):
// the same sources compiled with g++ -DCAPS_EMUL, exported as caps_sa_emul_*.  Used by
// tests/test_emul_*.py to debug kernel logic in the GPU-less dev container.
#define CAPS_EMUL 1
#define CAPS_API(name) caps_sa_emul_##name
#include <cstdint>
// statistics of the tile sort paths (read by tests)
static uint64_t g_tile_stats[8];
extern "C" void caps_emul_count_tile(bool fast, bool known_range) { ++g_tile_stats[(fast ? 1 : 0) + (known_range ? 2 : 0)]; }
extern "C" void caps_emul_count_tile2(bool ok) { ++g_tile_stats[4 + (ok ? 1 : 0)]; }
extern "C" void caps_emul_count_tile3(bool ok) { ++g_tile_stats[6 + (ok ? 1 : 0)]; }     // tile_sort_eq_kernel: gave up / finished
extern "C" void caps_sa_emul_tile_stats8(uint64_t* out, int reset) { for (int i = 0; i < 8; ++i) { out[i] = g_tile_stats[i]; if (reset) g_tile_stats[i] = 0; } }
extern "C" void caps_sa_emul_tile_stats6(uint64_t* out, int reset) { for (int i = 0; i < 6; ++i) { out[i] = g_tile_stats[i]; if (reset) g_tile_stats[i] = 0; } }
extern "C" void caps_sa_emul_tile_stats(uint64_t* out, int reset) { for (int i = 0; i < 4; ++i) { out[i] = g_tile_stats[i]; if (reset) g_tile_stats[i] = 0; } }
// the emulation's backend first: pipeline.h then skips the product's hip_backend.h (CAPS_BACKEND_DEFINED)
#include "../../caps-sa_amd/csrc/kernel_lang.h"
#include "emul_backend.h"
#include "../../caps-sa_amd/csrc/capi_impl.h"

namespace caps {
int set_device(int) { return CAPS_SA_OK; }
}
extern "C" int caps_sa_emul_device_count(void) { return 1; }
