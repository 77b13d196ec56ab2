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

// Hardware self-test of the LDS transpose read the bf16 weight-gradient kernel relies on: LDS holds lds[e] = e, lane l
// passes the address of elements [4l, 4l+4); out[l*4 + j] = what lane l received in element j.
__global__ void selftest_tr16_kernel(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
extern "C" int sgx_selftest_tr16(void* out256, void* stream) {
    hipLaunchKernelGGL(selftest_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (short*)out256);
    SGX_LAUNCH_CHECK("selftest_tr16");
    return 0;
}

extern "C" int sgx_version(void) { return SGX_VERSION; }
extern "C" const char* sgx_last_error(void) { return g_err; }
