// Internal helpers shared by the gfx950 kernels of libsgx_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <stdarg.h>
#include "sgx.h"

typedef unsigned short bf16_t;                                       // raw bf16 storage
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define SGX_LRELU 0.2f
#define WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {                    // round to nearest even
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two fp32 -> packed bf16x2, round to nearest even: one VALU instruction on gfx950 (the software path above is ~6)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : SGX_LRELU * v; }
__device__ __forceinline__ float lrelu_slope(float out_or_in) { return out_or_in > 0.f ? 1.f : SGX_LRELU; }
// 8 packed bf16 times the LeakyReLU slope selected by 8 packed bf16 of the activation's output: exactly the arithmetic of
// sgx_lrelu_bwd on the stored tensor (bf16 -> fp32, multiply, round to nearest even), so fusing it into a producer's store
// leaves every bit unchanged
__device__ __forceinline__ uint4 lrelu_mask_bf16x8(uint4 v, uint4 m) {
    unsigned vv[4] = {v.x, v.y, v.z, v.w};
    const unsigned mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(vv[i] << 16) * lrelu_slope(__uint_as_float(mm[i] << 16));
        const float hi = __uint_as_float(vv[i] & 0xffff0000u) * lrelu_slope(__uint_as_float(mm[i] & 0xffff0000u));
        vv[i] = pack_bf16x2(lo, hi);
    }
    return make_uint4(vv[0], vv[1], vv[2], vv[3]);
}
// SGX_ACT_NONE / SGX_ACT_LRELU / SGX_ACT_RELU (the reference's 'lrelu' | 'relu' nonlinearity, models/GAN.py:67-68)
__device__ __forceinline__ float act_apply(float v, int act) {
    return (act == SGX_ACT_NONE || v > 0.f) ? v : (act == SGX_ACT_RELU ? 0.f : SGX_LRELU * v);
}

// 16-byte vector access, VE elements of T per vector
template <typename T> struct VecTraits;
template <> struct VecTraits<float> {
    static constexpr int VE = 4;
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
        float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct VecTraits<bf16_t> {
    static constexpr int VE = 8;
    __device__ static __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        uint4 t = *reinterpret_cast<const uint4*>(p);
        unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    __device__ static __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
// 16-byte global accesses with the nontemporal hint (round 6, tools/stream_probe.hip: a read + write loop over tensors far larger than the
// caches runs 5-10 % faster when BOTH its loads and its stores carry it; either alone does nothing)
typedef unsigned sgx_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16(const void* p, bool nt) {
    if (nt) { const sgx_v4u v = __builtin_nontemporal_load(reinterpret_cast<const sgx_v4u*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
    return *reinterpret_cast<const uint4*>(p);
}
__device__ __forceinline__ void st16(void* p, const uint4& q, bool nt) {
    if (nt) { const sgx_v4u v = {q.x, q.y, q.z, q.w}; __builtin_nontemporal_store(v, reinterpret_cast<sgx_v4u*>(p)); }
    else *reinterpret_cast<uint4*>(p) = q;
}
// launches use the hint for tensors of at least this many bytes (SGX_NT_MIN_MB; 0 = never): smaller ones are the next kernel's cache hits
static inline bool sgx_nt_for(double tensor_bytes) {
    static const double min_bytes = [] { const char* e = getenv("SGX_NT_MIN_MB"); const double mb = e ? atof(e) : 192.0; return mb > 0 ? mb * 1e6 : 1e300; }();
    return tensor_bytes >= min_bytes;
}
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- host side error plumbing -------------------------------------------------------------
void sgx_set_error(const char* fmt, ...);
#define SGX_REQUIRE(cond, code, ...)                 \
    do {                                             \
        if (!(cond)) {                               \
            sgx_set_error(__VA_ARGS__);              \
            return (code);                           \
        }                                            \
    } while (0)
#define SGX_LAUNCH_CHECK(name)                                                         \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            sgx_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
            return (int)e_;                                                            \
        }                                                                              \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
int sgx_ncu();                                             // compute units of the current device (api.hip)

// ---- per-launch profiler (sgx_prof_* in include/sgx.h) --------------------------------------
// Every kernel launch of the library goes through sgx_launch.  When profiling is on it is bracketed by two HIP events
// recorded on the launch stream itself; records are resolved to kernel names and milliseconds by sgx_prof_get.
extern int sgx_prof_mode;                                  // 0 off, 1 every kernel, 2 only sgx_prof_only_fn
extern const void* sgx_prof_only_fn;
void sgx_prof_begin(const void* fn, hipStream_t st, int* slot);
void sgx_prof_end(int slot, hipStream_t st);
void sgx_prof_note(double flops, double bytes, const char* fmt, ...);   // describes the NEXT launch of this thread
template <typename K, typename... Args>
static inline void sgx_launch(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    int slot = -1;
    if (sgx_prof_mode) sgx_prof_begin(reinterpret_cast<const void*>(kern), st, &slot);
    kern<<<grid, block, lds, st>>>(args...);
    if (slot >= 0) sgx_prof_end(slot, st);
}
// The > 64 KB dynamic-LDS opt-in of a kernel is per DEVICE: applied once for every device a launch of this kernel is made on (a
// process that drives several GPUs would otherwise fail its first big-LDS launch on the second one).  Keyed by the kernel itself.
template <auto Kern>
static inline void sgx_lds_opt_in(int lds_bytes) {
    static bool done[32] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    if (!done[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        done[dev] = true;
    }
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern, grid, block, lds, st, ...) sgx_launch(kern, grid, block, lds, st, __VA_ARGS__)
#define SGX_NOTE(flops, bytes, ...)                                     \
    do {                                                                \
        if (sgx_prof_mode) sgx_prof_note((flops), (bytes), __VA_ARGS__); \
    } while (0)
