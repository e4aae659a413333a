// Error plumbing and device queries for libmvedit_amd.
#include "common.h"

static thread_local char g_err[1024] = "";

void mve_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char* mve_last_error(void) { return g_err; }

int mve_version(void) { return 1; }

int mve_device_info(int* n_cu, int* wave_size, char* arch, int arch_len) {
    int dev = 0;
    MVE_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    MVE_HIP(hipGetDeviceProperties(&prop, dev));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return MVE_OK;
}

}  // extern "C"
