// caps-sa_amd/csrc/caps_sa_hip.hip -- the product library: HIP kernels for gfx950 plus the
// C ABI of include/caps_sa_hip.h.  Build: caps-sa_amd/Makefile (hipcc --offload-arch=gfx950).
#define CAPS_API(name) caps_sa_hip_##name
#include "capi_impl.h"

namespace caps {
int set_device(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(CAPS_SA_ENODEVICE, "no HIP device");
    if (device < 0 || device >= count) return fail(CAPS_SA_EINVAL, "device ordinal out of range");
    if (hipSetDevice(device) != hipSuccess) return fail(CAPS_SA_EHIP, "hipSetDevice failed");
    return CAPS_SA_OK;
}
}  // namespace caps

extern "C" int caps_sa_hip_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}
