// Thread-local error message + version for libfmc_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

#include "../../include/fmc_hip.h"

static thread_local char g_err[512] = "";

void fmc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int fmc_version(void) { return FMC_VERSION; }
extern "C" const char* fmc_last_error(void) { return g_err; }

int fmc_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 ? dev : 0;
}

int fmc_cu_count() {
    static int n[64] = {0};
    const int dev = fmc_device() & 63;
    if (!n[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n[dev] = cus;
    }
    return n[dev];
}
