// Streaming-pass geometry probe (round 6): y = f(x) over a 537 MB bf16 tensor (read + write), the shape of the step's bandwidth-class passes.
// aten's elementwise add moves the same bytes at 6.3 TB/s on this chip (tools/overlap_probe.py); the library's passes sit at 4.9-5.2 TB/s.
// Which launch geometry / access pattern closes the gap?        build: hipcc --offload-arch=gfx950 -O3 -o tools/stream_probe tools/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint4 f(uint4 v) { v.x ^= 0x00010001u; v.y += 1u; return v; }
__device__ __forceinline__ uint2 f2(uint2 v) { v.x ^= 0x00010001u; v.y += 1u; return v; }

// V0: the library's pattern: capped grid, grid-stride loop, one 16-byte vector per lane and iteration
__global__ __launch_bounds__(256) void k_gridstride(const uint4* __restrict__ x, uint4* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = f(x[i]);
}
// V1: grid-stride, U independent loads in flight per lane
template <int U>
__global__ __launch_bounds__(256) void k_gridstride_u(const uint4* __restrict__ x, uint4* __restrict__ y, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) y[i + u * stride] = f(v[u]);
    }
    for (; i < n; i += stride) y[i] = f(x[i]);
}
// V2: one block per contiguous chunk of 256 * U vectors, no loop (short-lived blocks, huge grid)
template <int U, int BS>
__global__ __launch_bounds__(BS) void k_chunk(const uint4* __restrict__ x, uint4* __restrict__ y, size_t n) {
    const size_t base = (size_t)blockIdx.x * (BS * U) + threadIdx.x;
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * BS < n) v[u] = x[base + u * BS];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * BS < n) y[base + u * BS] = f(v[u]);
}
// V3: the same with nontemporal loads and stores
template <int U, int BS, int NTL, int NTS>
__global__ __launch_bounds__(BS) void k_chunk_nt(const uint4* __restrict__ x, uint4* __restrict__ y, size_t n) {
    const size_t base = (size_t)blockIdx.x * (BS * U) + threadIdx.x;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * BS < n) v[u] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const v4u*>(x) + base + u * BS) : reinterpret_cast<const v4u*>(x)[base + u * BS];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * BS < n) {
            v4u o = v[u]; o.x ^= 0x00010001u; o.y += 1u;
            if (NTS) __builtin_nontemporal_store(o, reinterpret_cast<v4u*>(y) + base + u * BS);
            else reinterpret_cast<v4u*>(y)[base + u * BS] = o;
        }
}
// V4: aten-like: 8-byte vectors, 4 per thread
template <int U, int BS>
__global__ __launch_bounds__(BS) void k_chunk8(const uint2* __restrict__ x, uint2* __restrict__ y, size_t n2) {
    const size_t base = (size_t)blockIdx.x * (BS * U) + threadIdx.x;
    uint2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * BS < n2) v[u] = x[base + u * BS];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * BS < n2) y[base + u * BS] = f2(v[u]);
}
// V5: persistent blocks (grid = CUs * k), each walking CONTIGUOUS chunks (block-contiguous instead of grid-strided addresses)
template <int U>
__global__ __launch_bounds__(256) void k_persist_chunks(const uint4* __restrict__ x, uint4* __restrict__ y, size_t n) {
    const size_t nchunk = (n + 256 * U - 1) / (256 * U);
    for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
        const size_t base = c * (256 * U) + threadIdx.x;
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = x[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * 256 < n) y[base + u * 256] = f(v[u]);
    }
}


// ---- realistic pass shapes -------------------------------------------------------------------------------------------------
// apply-like: y[p][c] = ((lrelu(x + kb[c] + kw[c] nz[p]) - km[b,c]) kr[b,c]) ks[b,c] + k1[b,c]; C channels (NHWC), 8 bf16 per lane
__device__ __forceinline__ float bfl(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bfh(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned pk(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
__device__ __forceinline__ uint4 apply8(uint4 v, float nz, const float* kb, const float* kw, const float* km, const float* kr, const float* ks, const float* k1) {
    unsigned w[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a0 = bfl(w[q]) + kb[2 * q] + kw[2 * q] * nz, a1 = bfh(w[q]) + kb[2 * q + 1] + kw[2 * q + 1] * nz;
        a0 = a0 > 0.f ? a0 : 0.2f * a0; a1 = a1 > 0.f ? a1 : 0.2f * a1;
        o[q] = pk(((a0 - km[2 * q]) * kr[2 * q]) * ks[2 * q] + k1[2 * q], ((a1 - km[2 * q + 1]) * kr[2 * q + 1]) * ks[2 * q + 1] + k1[2 * q + 1]);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// A0: per-thread coefficient loads from global (6 arrays x 32 bytes), U pixels of the same channel vector per thread, no loop
template <int U>
__global__ __launch_bounds__(256) void k_apply_glob(const uint4* __restrict__ x, uint4* __restrict__ y, const float* __restrict__ nzp, const float* __restrict__ coef,
                                                    size_t n, int cv, size_t vec_per_img) {
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    if (base >= n) return;
    const int v = (int)(base % cv);
    const size_t b = base / vec_per_img;
    const float* cb = coef + (b * cv + v) * 48;
    float k[48];
#pragma unroll
    for (int j = 0; j < 12; ++j) { const float4 t = reinterpret_cast<const float4*>(cb)[j]; k[4 * j] = t.x; k[4 * j + 1] = t.y; k[4 * j + 2] = t.z; k[4 * j + 3] = t.w; }
    uint4 d[U]; float nz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { d[u] = x[base + u * 256]; nz[u] = nzp[(base + u * 256) / cv]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) y[base + u * 256] = apply8(d[u], nz[u], k, k + 8, k + 16, k + 24, k + 32, k + 40);
}
// A1: the block's coefficient table through LDS (cv * 48 floats), one pixel vector per thread
template <int U>
__global__ __launch_bounds__(256) void k_apply_lds(const uint4* __restrict__ x, uint4* __restrict__ y, const float* __restrict__ nzp, const float* __restrict__ coef,
                                                   size_t n, int cv, size_t vec_per_img) {
    __shared__ float tab[16 * 48];
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    const size_t b = ((size_t)blockIdx.x * (256 * U)) / vec_per_img;
    uint4 d[U]; float nz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { d[u] = x[base + u * 256]; nz[u] = nzp[(base + u * 256) / cv]; }
    for (int i = threadIdx.x; i < cv * 12; i += 256) reinterpret_cast<float4*>(tab)[i] = reinterpret_cast<const float4*>(coef + b * cv * 48)[i];
    __syncthreads();
    if (base >= n) return;
    const float* k = tab + (base % cv) * 48;
    float kk[48];
#pragma unroll
    for (int j = 0; j < 48; ++j) kk[j] = k[j];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) y[base + u * 256] = apply8(d[u], nz[u], kk, kk + 8, kk + 16, kk + 24, kk + 32, kk + 40);
}
// A2: the library's structure: a block owns a contiguous chunk of ROWS-strided pixels, RPT rows per thread in a loop, coefficients in registers
__global__ __launch_bounds__(256) void k_apply_lib(const uint4* __restrict__ x, uint4* __restrict__ y, const float* __restrict__ nzp, const float* __restrict__ coef,
                                                   int HW, int cv, int rpt) {
    const int b = blockIdx.y, rows = 256 / cv, tc = threadIdx.x % cv, tr = threadIdx.x / cv;
    const float* cb = coef + ((size_t)b * cv + tc) * 48;
    float k[48];
#pragma unroll
    for (int j = 0; j < 12; ++j) { const float4 t = reinterpret_cast<const float4*>(cb)[j]; k[4 * j] = t.x; k[4 * j + 1] = t.y; k[4 * j + 2] = t.z; k[4 * j + 3] = t.w; }
    const int p0 = blockIdx.x * rows * rpt, p1 = min(HW, p0 + rows * rpt);
#pragma unroll 4
    for (int p = p0 + tr; p < p1; p += rows) {
        const size_t off = ((size_t)b * HW + p) * cv + tc;
        y[off] = apply8(x[off], nzp[(size_t)b * HW + p], k, k + 8, k + 16, k + 24, k + 32, k + 40);
    }
}
// T0: two inputs, one output (the backward apply's traffic), one vector per thread
template <int U>
__global__ __launch_bounds__(256) void k_two_in(const uint4* __restrict__ x, const uint4* __restrict__ g, uint4* __restrict__ y, size_t n) {
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    uint4 a[U], c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { a[u] = x[base + u * 256]; c[u] = g[base + u * 256]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { uint4 o = f(a[u]); o.z ^= c[u].z; o.w += c[u].w; o.x += c[u].x; o.y ^= c[u].y; y[base + u * 256] = o; }
}
__global__ __launch_bounds__(256) void k_two_in_loop(const uint4* __restrict__ x, const uint4* __restrict__ g, uint4* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 o = f(x[i]); const uint4 c = g[i]; o.z ^= c.z; o.w += c.w; o.x += c.x; o.y ^= c.y; y[i] = o;
    }
}
// R: read-only reduction.  R0 grid-stride loop with one partial per block; R1 chunk blocks with U vectors per thread and one partial per block
__device__ __forceinline__ float vsum(uint4 v) { return bfl(v.x) + bfh(v.x) + bfl(v.y) + bfh(v.y) + bfl(v.z) + bfh(v.z) + bfl(v.w) + bfh(v.w); }
__device__ __forceinline__ float block_sum(float s, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void k_red_loop(const uint4* __restrict__ x, float* __restrict__ part, size_t n) {
    __shared__ float sh[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += vsum(x[i]);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
template <int U>
__global__ __launch_bounds__(256) void k_red_chunk(const uint4* __restrict__ x, float* __restrict__ part, size_t n) {
    __shared__ float sh[4];
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = base + u * 256 < n ? x[base + u * 256] : make_uint4(0, 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += vsum(v[u]);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
template <int BS>
__global__ __launch_bounds__(BS) void k_chunk_bs(const uint4* __restrict__ x, uint4* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    if (i < n) y[i] = f(x[i]);
}


// W: write-heavy mixes.  W0 pure write (537 MB), W1 read 1 : write 2 (a transposed convolution 32 -> 16 channels to 4x the pixels), W2 read 2 : write 1
template <int BS>
__global__ __launch_bounds__(BS) void k_write_only(uint4* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    if (i < n) y[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ __launch_bounds__(256) void k_write_loop(uint4* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ __launch_bounds__(256) void k_r1w2(const uint4* __restrict__ x, uint4* __restrict__ y, uint4* __restrict__ y2, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const uint4 v = f(x[i]); y[i] = v; y2[i] = v; }
}


// NT: the loop forms with nontemporal loads / stores (do the library's long-lived kernels gain what the short-lived probe blocks gain from them?)
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k_gridstride_nt(const uint4* __restrict__ x, uint4* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        v4u_t v = NTL ? __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(x) + i) : reinterpret_cast<const v4u_t*>(x)[i];
        v.x ^= 0x00010001u; v.y += 1u;
        if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<v4u_t*>(y) + i); else reinterpret_cast<v4u_t*>(y)[i] = v;
    }
}
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k_apply_lib_nt(const uint4* __restrict__ x, uint4* __restrict__ y, const float* __restrict__ nzp, const float* __restrict__ coef,
                                                      int HW, int cv, int rpt) {
    const int b = blockIdx.y, rows = 256 / cv, tc = threadIdx.x % cv, tr = threadIdx.x / cv;
    const float* cb = coef + ((size_t)b * cv + tc) * 48;
    float k[48];
#pragma unroll
    for (int j = 0; j < 12; ++j) { const float4 t = reinterpret_cast<const float4*>(cb)[j]; k[4 * j] = t.x; k[4 * j + 1] = t.y; k[4 * j + 2] = t.z; k[4 * j + 3] = t.w; }
    const int p0 = blockIdx.x * rows * rpt, p1 = min(HW, p0 + rows * rpt);
#pragma unroll 4
    for (int p = p0 + tr; p < p1; p += rows) {
        const size_t off = ((size_t)b * HW + p) * cv + tc;
        const v4u_t r = NTL ? __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(x) + off) : reinterpret_cast<const v4u_t*>(x)[off];
        const uint4 o = apply8(make_uint4(r.x, r.y, r.z, r.w), nzp[(size_t)b * HW + p], k, k + 8, k + 16, k + 24, k + 32, k + 40);
        const v4u_t ov = {o.x, o.y, o.z, o.w};
        if (NTS) __builtin_nontemporal_store(ov, reinterpret_cast<v4u_t*>(y) + off); else reinterpret_cast<v4u_t*>(y)[off] = ov;
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)32 * 512 * 512 * 32 * 2;      // 537 MB
    const size_t n = bytes / 16;
    const int NB = 3, reps = 12;
    std::vector<uint4*> xs(NB), ys(NB);
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&xs[i], bytes)); CK(hipMalloc(&ys[i], bytes)); CK(hipMemset(xs[i], 1, bytes)); CK(hipMemset(ys[i], 0, bytes)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto bench = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch(xs[i % NB], ys[i % NB]);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch(xs[i % NB], ys[i % NB]);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        const double us = best * 1e3 / reps;
        printf("%-52s %8.1f us  %6.2f TB/s\n", name, us, 2.0 * bytes / us / 1e6);
        fflush(stdout);
    };
    bench("V0 grid-stride 8192x256, 16 B x1", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_gridstride, dim3(8192), dim3(256), 0, 0, x, y, n); });
    bench("V0 grid-stride 2048x256", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_gridstride, dim3(2048), dim3(256), 0, 0, x, y, n); });
    bench("V0 grid-stride 32768x256", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_gridstride, dim3(32768), dim3(256), 0, 0, x, y, n); });
    bench("V1 grid-stride 8192x256, 16 B x4 in flight", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_gridstride_u<4>, dim3(8192), dim3(256), 0, 0, x, y, n); });
    bench("V1 grid-stride 2048x256, 16 B x4 in flight", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_gridstride_u<4>, dim3(2048), dim3(256), 0, 0, x, y, n); });
    bench("V1 grid-stride 2048x256, 16 B x8 in flight", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_gridstride_u<8>, dim3(2048), dim3(256), 0, 0, x, y, n); });
    bench("V2 chunk blocks 256 thr, 16 B x1 (no loop)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk<1, 256>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, y, n); });
    bench("V2 chunk blocks 256 thr, 16 B x2", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk<2, 256>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, x, y, n); });
    bench("V2 chunk blocks 256 thr, 16 B x4", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk<4, 256>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, x, y, n); });
    bench("V2 chunk blocks 256 thr, 16 B x8", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk<8, 256>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, x, y, n); });
    bench("V2 chunk blocks 512 thr, 16 B x4", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk<4, 512>), dim3((unsigned)((n + 2047) / 2048)), dim3(512), 0, 0, x, y, n); });
    bench("V2 chunk blocks 1024 thr, 16 B x2", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk<2, 1024>), dim3((unsigned)((n + 2047) / 2048)), dim3(1024), 0, 0, x, y, n); });
    bench("V3 chunk 256 thr x4, nt load", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk_nt<4, 256, 1, 0>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, x, y, n); });
    bench("V3 chunk 256 thr x4, nt store", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk_nt<4, 256, 0, 1>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, x, y, n); });
    bench("V3 chunk 256 thr x4, nt load + store", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk_nt<4, 256, 1, 1>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, x, y, n); });
    bench("V4 chunk 256 thr, 8 B x4 (aten-like)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk8<4, 256>), dim3((unsigned)((2 * n + 1023) / 1024)), dim3(256), 0, 0, (const uint2*)x, (uint2*)y, 2 * n); });
    bench("V4 chunk 256 thr, 8 B x8", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_chunk8<8, 256>), dim3((unsigned)((2 * n + 2047) / 2048)), dim3(256), 0, 0, (const uint2*)x, (uint2*)y, 2 * n); });
    bench("V5 persistent 2048 blocks, contiguous chunks x4", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_persist_chunks<4>, dim3(2048), dim3(256), 0, 0, x, y, n); });
    bench("V5 persistent 8192 blocks, contiguous chunks x4", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_persist_chunks<4>, dim3(8192), dim3(256), 0, 0, x, y, n); });
    bench("V5 persistent 1024 blocks, contiguous chunks x8", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_persist_chunks<8>, dim3(1024), dim3(256), 0, 0, x, y, n); });

    // ---- block size of the one-vector-per-thread form
    bench("V2 x1, 64-thread blocks", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_chunk_bs<64>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, x, y, n); });
    bench("V2 x1, 128-thread blocks", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_chunk_bs<128>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, 0, x, y, n); });
    bench("V2 x1, 512-thread blocks", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_chunk_bs<512>, dim3((unsigned)((n + 511) / 512)), dim3(512), 0, 0, x, y, n); });
    bench("V2 x1, 1024-thread blocks", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_chunk_bs<1024>, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, 0, x, y, n); });
    // ---- apply-like pass: [32][512*512][32 channels] -> cv = 4 vectors per pixel
    {
        const int C = 32, cv = C / 8, B = 32, HW = 512 * 512;
        const size_t vpi = (size_t)HW * cv;
        float *nz, *coef;
        CK(hipMalloc(&nz, (size_t)B * HW * 4)); CK(hipMemset(nz, 0, (size_t)B * HW * 4));
        CK(hipMalloc(&coef, (size_t)B * cv * 48 * 4)); CK(hipMemset(coef, 0, (size_t)B * cv * 48 * 4));
        bench("A0 apply, coefficients per thread from global, x1", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_apply_glob<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, y, nz, coef, n, cv, vpi); });
        bench("A0 apply, coefficients per thread from global, x2", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_apply_glob<2>, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, x, y, nz, coef, n, cv, vpi); });
        bench("A0 apply, coefficients per thread from global, x4", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_apply_glob<4>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, x, y, nz, coef, n, cv, vpi); });
        bench("A0 apply, coefficients per thread from global, x8", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_apply_glob<8>, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, x, y, nz, coef, n, cv, vpi); });
        bench("A1 apply, coefficient table through LDS, x1", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_apply_lds<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, y, nz, coef, n, cv, vpi); });
        bench("A1 apply, coefficient table through LDS, x4", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_apply_lds<4>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, x, y, nz, coef, n, cv, vpi); });
        for (int rpt : {64, 16, 4}) {
            char nm[96]; snprintf(nm, sizeof nm, "A2 apply, library structure, %d rows per thread", rpt);
            const int rows = 256 / cv;
            bench(nm, [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_apply_lib, dim3((HW + rows * rpt - 1) / (rows * rpt), B), dim3(256), 0, 0, x, y, nz, coef, HW, cv, rpt); });
        }
    }
    // ---- two inputs, one output (1.61 GB per launch: TB/s printed for 2 x 537 MB, scale by 1.5)
    bench("T0 two inputs x1 (bytes x1.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_two_in<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, xs[(x == xs[0]) ? 1 : 0], y, n); });
    bench("T0 two inputs x2 (bytes x1.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_two_in<2>, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, x, xs[(x == xs[0]) ? 1 : 0], y, n); });
    bench("T0 two inputs, grid-stride 8192 (bytes x1.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_two_in_loop, dim3(8192), dim3(256), 0, 0, x, xs[(x == xs[0]) ? 1 : 0], y, n); });
    // ---- read-only reduction (537 MB per launch: TB/s printed for 2 x 537 MB, scale by 0.5)
    {
        float* part; CK(hipMalloc(&part, ((n + 255) / 256) * 4));
        bench("R0 reduce, grid-stride 8192 (bytes x0.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_red_loop, dim3(8192), dim3(256), 0, 0, x, part, n); });
        bench("R0 reduce, grid-stride 2048 (bytes x0.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_red_loop, dim3(2048), dim3(256), 0, 0, x, part, n); });
        bench("R1 reduce, chunk blocks x1 (bytes x0.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_red_chunk<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, part, n); });
        bench("R1 reduce, chunk blocks x4 (bytes x0.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_red_chunk<4>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, x, part, n); });
        bench("R1 reduce, chunk blocks x16 (bytes x0.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_red_chunk<16>, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, 0, x, part, n); });
    }

    bench("W0 write only, x1 chunk blocks (bytes x0.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_write_only<256>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, y, n); });
    bench("W0 write only, grid-stride 8192 (bytes x0.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_write_loop, dim3(8192), dim3(256), 0, 0, y, n); });
    bench("W1 read 1 : write 2, x1 chunk blocks (bytes x1.5)", [&](uint4* x, uint4* y) { hipLaunchKernelGGL(k_r1w2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, y, ys[(y == ys[0]) ? 1 : 0], n); });

    bench("NT grid-stride 8192x256, nt load", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_gridstride_nt<1, 0>), dim3(8192), dim3(256), 0, 0, x, y, n); });
    bench("NT grid-stride 8192x256, nt store", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_gridstride_nt<0, 1>), dim3(8192), dim3(256), 0, 0, x, y, n); });
    bench("NT grid-stride 8192x256, nt load + store", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_gridstride_nt<1, 1>), dim3(8192), dim3(256), 0, 0, x, y, n); });
    {
        const int C = 32, cv = C / 8, B = 32, HW = 512 * 512, rows = 256 / cv, rpt = 64;
        float *nz, *coef;
        CK(hipMalloc(&nz, (size_t)B * HW * 4)); CK(hipMemset(nz, 0, (size_t)B * HW * 4));
        CK(hipMalloc(&coef, (size_t)B * cv * 48 * 4)); CK(hipMemset(coef, 0, (size_t)B * cv * 48 * 4));
        bench("NT apply, library structure 64 rows, plain", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_apply_lib_nt<0, 0>), dim3((HW + rows * rpt - 1) / (rows * rpt), B), dim3(256), 0, 0, x, y, nz, coef, HW, cv, rpt); });
        bench("NT apply, library structure 64 rows, nt load", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_apply_lib_nt<1, 0>), dim3((HW + rows * rpt - 1) / (rows * rpt), B), dim3(256), 0, 0, x, y, nz, coef, HW, cv, rpt); });
        bench("NT apply, library structure 64 rows, nt store", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_apply_lib_nt<0, 1>), dim3((HW + rows * rpt - 1) / (rows * rpt), B), dim3(256), 0, 0, x, y, nz, coef, HW, cv, rpt); });
        bench("NT apply, library structure 64 rows, nt load + store", [&](uint4* x, uint4* y) { hipLaunchKernelGGL((k_apply_lib_nt<1, 1>), dim3((HW + rows * rpt - 1) / (rows * rpt), B), dim3(256), 0, 0, x, y, nz, coef, HW, cv, rpt); });
    }
    return 0;
}
