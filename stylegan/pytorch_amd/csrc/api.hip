// Error plumbing + version of the C ABI (include/sgx.h).
#include "common.h"

#define SGX_VERSION 100   // 0.1.0

static thread_local char g_err[512] = "";

void sgx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int sgx_version(void) { return SGX_VERSION; }
extern "C" const char* sgx_last_error(void) { return g_err; }
