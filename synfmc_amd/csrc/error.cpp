// Thread-local error message + version for libfmc_hip.so.
#include <stdarg.h>
#include <stdio.h>

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
