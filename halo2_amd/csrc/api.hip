// Library-level entry points of libhalo2_mi355x.so: device discovery, error reporting.
// The MSM entry points live in msm.hip, the NTT entry points in ntt.hip (see include/halo2_mi355x.h).
#include <cstdio>
#include <cstring>

#include "common.h"

namespace h2 {

static thread_local char g_err[256] = "no error";

void set_last_hip_error(hipError_t e, const char *file, int line) {
    const char *base = strrchr(file, '/');
    snprintf(g_err, sizeof g_err, "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), base ? base + 1 : file, line);
}

int ensure_device() {
    static thread_local int checked = 0;
    if (checked) return H2_OK;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        snprintf(g_err, sizeof g_err, "no HIP device available (%s): the MI355X path has no CPU fallback",
                 e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return H2_ERR_NODEV;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        snprintf(g_err, sizeof g_err, "cannot query the current HIP device");
        return H2_ERR_NODEV;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof g_err, "device %d is %s; this library ships gfx950 (MI355X) code only", dev, prop.gcnArchName);
        return H2_ERR_NODEV;
    }
    checked = 1;
    return H2_OK;
}

}  // namespace h2

extern "C" int h2_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

extern "C" int h2_init(int device) {
    int count = h2_device_count();
    if (count <= 0) {
        snprintf(h2::g_err, sizeof h2::g_err, "no HIP device available");
        return H2_ERR_NODEV;
    }
    if (device < 0 || device >= count) return H2_ERR_ARGS;
    H2_HIP(hipSetDevice(device));
    return h2::ensure_device();
}

extern "C" const char *h2_last_error(void) { return h2::g_err; }
