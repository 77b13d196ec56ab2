// Small fp32 pieces: PixelNorm rows, minibatch-stddev (forward / backward / second-order backward for R1),
// and the MFMA fp32 GEMM behind EqualizedLinear (mapping network, style affine maps, discriminator head).
#include "common.h"

// ---------------------------------------------------------------- PixelNorm over the feature dim of [B][C]
__global__ void pixelnorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C) {
    const int row = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
    if (row >= B) return;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = x[(size_t)row * C + c]; ss += v * v; }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / C + 1e-8f);
    for (int c = lane; c < C; c += 64) y[(size_t)row * C + c] = x[(size_t)row * C + c] * r;
}
__global__ void pixelnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int B, int C) {
    const int row = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
    if (row >= B) return;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[(size_t)row * C + c];
        ss += v * v; dot += v * dy[(size_t)row * C + c];
    }
    ss = wave_sum(ss); dot = wave_sum(dot);
    const float r = rsqrtf(ss / C + 1e-8f);
    const float k = r * r * r * dot / C;
    for (int c = lane; c < C; c += 64) dx[(size_t)row * C + c] = dy[(size_t)row * C + c] * r - x[(size_t)row * C + c] * k;
}
extern "C" int sgx_pixelnorm_fwd(const float* x, float* y, int B, int C, void* stream) {
    hipLaunchKernelGGL(pixelnorm_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, B, C);
    SGX_LAUNCH_CHECK("pixelnorm_fwd");
    return 0;
}
extern "C" int sgx_pixelnorm_bwd(const float* dy, const float* x, float* dx, int B, int C, void* stream) {
    hipLaunchKernelGGL(pixelnorm_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, x, dx, B, C);
    SGX_LAUNCH_CHECK("pixelnorm_bwd");
    return 0;
}

// ---------------------------------------------------------------- minibatch stddev
// x [B][HW][C], y/dy [B][HW][Cpad].  Group g, slot m: sample g*M+m.  One block per slot m.
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    return s;
}

template <typename T>
__global__ __launch_bounds__(1024) void mbstd_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int HW, int C, int Cpad) {
    __shared__ double sh[16];
    const int G = B < 4 ? B : 4, M = B / G, m = blockIdx.x, n = HW * C;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float v[4], mu = 0.f;
        for (int g = 0; g < G; ++g) { v[g] = to_f(x[(size_t)(g * M + m) * n + i]); mu += v[g]; }
        mu /= G;
        float var = 0.f;
        for (int g = 0; g < G; ++g) { const float d = v[g] - mu; var += d * d; }
        acc += (double)sqrtf(var / G + 1e-8f);
    }
    const float stat = (float)(block_sum_d(acc, sh) / n);
    // every block of the slot computed the statistic (redundantly: it is a 32-KB reduction); the padded copy is split over them
    for (int g = 0; g < G; ++g) {
        const size_t b = (size_t)(g * M + m);
        for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < HW * Cpad; i += gridDim.y * blockDim.x) {
            const int p = i / Cpad, c = i % Cpad;
            float v = 0.f;
            if (c < C) v = to_f(x[(b * HW + p) * C + c]);
            else if (c == C) v = stat;
            y[(b * HW + p) * Cpad + c] = from_f<T>(v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void mbstd_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, int B,
                                                        int HW, int C, int Cpad) {
    __shared__ double sh[16];
    const int G = B < 4 ? B : 4, M = B / G, m = blockIdx.x, n = HW * C;
    double acc = 0.0;
    for (int i = threadIdx.x; i < G * HW; i += blockDim.x) {
        const int g = i / HW, p = i % HW;
        acc += (double)to_f(dy[((size_t)(g * M + m) * HW + p) * Cpad + C]);
    }
    const float k = (float)(block_sum_d(acc, sh) / n) / G;              // gy / (C*HW*G)
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
        const int p = i / C, c = i % C;
        float v[4], mu = 0.f;
        for (int g = 0; g < G; ++g) { v[g] = to_f(x[(size_t)(g * M + m) * n + i]); mu += v[g]; }
        mu /= G;
        float var = 0.f;
        for (int g = 0; g < G; ++g) { const float d = v[g] - mu; var += d * d; }
        const float s = sqrtf(var / G + 1e-8f);
        for (int g = 0; g < G; ++g) {
            const size_t b = (size_t)(g * M + m);
            dx[b * n + i] = from_f<T>(to_f(dy[(b * HW + p) * Cpad + c]) + k * (v[g] - mu) / s);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void mbstd_bwd2_kernel(const T* __restrict__ ggx, const T* __restrict__ dy, const T* __restrict__ x,
                                                         T* __restrict__ ddy, T* __restrict__ gx, int B, int HW, int C, int Cpad) {
    __shared__ double sh[16];
    const int G = B < 4 ? B : 4, M = B / G, m = blockIdx.x, n = HW * C;
    double a_gy = 0.0, a_dd = 0.0;
    for (int i = threadIdx.x; i < G * HW; i += blockDim.x) {
        const int g = i / HW, p = i % HW;
        a_gy += (double)to_f(dy[((size_t)(g * M + m) * HW + p) * Cpad + C]);
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float v[4], mu = 0.f;
        for (int g = 0; g < G; ++g) { v[g] = to_f(x[(size_t)(g * M + m) * n + i]); mu += v[g]; }
        mu /= G;
        float var = 0.f, dot = 0.f;
        for (int g = 0; g < G; ++g) {
            const float d = v[g] - mu;
            var += d * d;
            dot += d * to_f(ggx[(size_t)(g * M + m) * n + i]);
        }
        a_dd += (double)(dot / sqrtf(var / G + 1e-8f));
    }
    const float k = (float)(block_sum_d(a_gy, sh) / n) / G;             // gy / (C*HW*G)
    const float dstat = (float)(block_sum_d(a_dd, sh) / n) / G;         // d L / d gy[m]
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
        float v[4], gg[4], mu = 0.f;
        for (int g = 0; g < G; ++g) {
            v[g] = to_f(x[(size_t)(g * M + m) * n + i]); mu += v[g];
            gg[g] = to_f(ggx[(size_t)(g * M + m) * n + i]);
        }
        mu /= G;
        float var = 0.f, dot = 0.f;
        for (int g = 0; g < G; ++g) { const float d = v[g] - mu; var += d * d; dot += d * gg[g]; }
        const float s = sqrtf(var / G + 1e-8f);
        float u[4], um = 0.f;
        for (int g = 0; g < G; ++g) { u[g] = gg[g] / s - dot * (v[g] - mu) / (G * s * s * s); um += u[g]; }
        um /= G;
        for (int g = 0; g < G; ++g) gx[(size_t)(g * M + m) * n + i] = from_f<T>(k * (u[g] - um));
    }
    for (int g = 0; g < G; ++g) {
        const size_t b = (size_t)(g * M + m);
        for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < HW * Cpad; i += gridDim.y * blockDim.x) {
            const int p = i / Cpad, c = i % Cpad;
            float v = 0.f;
            if (c < C) v = to_f(ggx[(b * HW + p) * C + c]);
            else if (c == C) v = dstat;
            ddy[(b * HW + p) * Cpad + c] = from_f<T>(v);
        }
    }
}

#ifndef MBSTD_NB
#define MBSTD_NB 8                                             // blocks per slot: each recomputes the scalars, writes 1/8 of the output
#endif
static int mbstd_check(int B, int C, int Cpad) {
    SGX_REQUIRE(B > 0 && (B < 4 || B % 4 == 0), SGX_EINVAL, "mbstd: batch %d not divisible by group size", B);
    SGX_REQUIRE(Cpad > C, SGX_EINVAL, "mbstd: Cpad must exceed C");
    return 0;
}
extern "C" int sgx_mbstd_fwd(const void* x, void* y, int B, int HW, int C, int Cpad, int dtype, void* stream) {
    int rc = mbstd_check(B, C, Cpad); if (rc) return rc;
    const int M = B / (B < 4 ? B : 4);
    if (dtype == SGX_F32) hipLaunchKernelGGL(mbstd_fwd_kernel<float>, dim3(M, MBSTD_NB), dim3(1024), 0, (hipStream_t)stream, (const float*)x, (float*)y, B, HW, C, Cpad);
    else hipLaunchKernelGGL(mbstd_fwd_kernel<bf16_t>, dim3(M, MBSTD_NB), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, HW, C, Cpad);
    SGX_LAUNCH_CHECK("mbstd_fwd");
    return 0;
}
extern "C" int sgx_mbstd_bwd(const void* dy, const void* x, void* dx, int B, int HW, int C, int Cpad, int dtype, void* stream) {
    int rc = mbstd_check(B, C, Cpad); if (rc) return rc;
    const int M = B / (B < 4 ? B : 4);
    if (dtype == SGX_F32) hipLaunchKernelGGL(mbstd_bwd_kernel<float>, dim3(M, MBSTD_NB), dim3(1024), 0, (hipStream_t)stream, (const float*)dy, (const float*)x, (float*)dx, B, HW, C, Cpad);
    else hipLaunchKernelGGL(mbstd_bwd_kernel<bf16_t>, dim3(M, MBSTD_NB), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, B, HW, C, Cpad);
    SGX_LAUNCH_CHECK("mbstd_bwd");
    return 0;
}
extern "C" int sgx_mbstd_bwd2(const void* ggx, const void* dy, const void* x, void* ddy, void* gx, int B, int HW, int C, int Cpad,
                              int dtype, void* stream) {
    int rc = mbstd_check(B, C, Cpad); if (rc) return rc;
    const int M = B / (B < 4 ? B : 4);
    if (dtype == SGX_F32) hipLaunchKernelGGL(mbstd_bwd2_kernel<float>, dim3(M, MBSTD_NB), dim3(1024), 0, (hipStream_t)stream, (const float*)ggx, (const float*)dy, (const float*)x, (float*)ddy, (float*)gx, B, HW, C, Cpad);
    else hipLaunchKernelGGL(mbstd_bwd2_kernel<bf16_t>, dim3(M, MBSTD_NB), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)ggx, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)ddy, (bf16_t*)gx, B, HW, C, Cpad);
    SGX_LAUNCH_CHECK("mbstd_bwd2");
    return 0;
}

// ---------------------------------------------------------------- fp32 GEMM on v_mfma_f32_16x16x4_f32
// C[M][N] = alpha * op(A)[M][K] * op(B)[K][N].  One 16x16 output tile per block; the block's 4 waves split K and
// reduce through LDS (fixed order: deterministic).  Operand fragments are read straight from global memory: these
// GEMMs have M = batch (4..64) and are weight-bandwidth/latency bound.
// Optional fusions for the EqualizedLinear layers (mapping network, style affines):
//   amask  : A is read as A * slope(amask) (same layout as A) -- the LeakyReLU backward folded into the operand load
//   bias   : C = act(alpha * A B + bscale * bias[column])      -- bias and LeakyReLU folded into the store
//   colsum : colsum[row of C] = cscale * sum_k A'[row][k]      -- with A = gy^T this is the bias gradient, out of the
//            same launch as the weight gradient (written by the blocks of column tile 0)
struct GemmFuse { const float* amask; const float* bias; float bscale; int act; float* colsum; float cscale; };
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ Cm,
                                                       int M, int N, int K, int ta, int tb, float alpha, GemmFuse fu) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, l15 = lane & 15;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const int i = i0 + l15, j = j0 + l15;
    // split K: blockIdx.z owns [z*kz, (z+1)*kz) (partials go to Cm + z*M*N, summed by gemm_splitk_reduce), its 4 waves
    // split that range again
    const int kz = ((K + (int)gridDim.z - 1) / (int)gridDim.z + 15) / 16 * 16;
    const int kz0 = blockIdx.z * kz, kz1 = (kz0 + kz < K) ? kz0 + kz : K;
    const int kper = (((kz1 > kz0 ? kz1 - kz0 : 0) + 3) / 4 + 15) / 16 * 16;   // K slice per wave, multiple of 16
    const int kb = kz0 + wave * kper, ke = (kb + kper < kz1) ? kb + kper : kz1;
    Cm += (size_t)blockIdx.z * M * N;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t sa_i = ta ? 1 : (size_t)K, sa_k = ta ? (size_t)M : 1;
    const size_t sb_k = tb ? 1 : (size_t)N, sb_j = tb ? (size_t)K : 1;
    // 16 k per iteration: lane group q owns k = k0+4q..+3 (any k order is valid as long as A and B agree), so an
    // operand whose k axis is contiguous is fetched with one 16-byte load per lane instead of four 4-byte ones.
    const bool va = !ta && (K % 4 == 0), vb = tb && (K % 4 == 0);
#pragma unroll 8
    for (int k = kb; k < ke; k += 16) {
        const int kk = k + 4 * q;
        float a4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (i < M) {
            if (va && kk + 3 < ke) {
                const float4 t = *reinterpret_cast<const float4*>(A + i * sa_i + kk);
                a4[0] = t.x; a4[1] = t.y; a4[2] = t.z; a4[3] = t.w;
                if (fu.amask) {
                    const float4 m = *reinterpret_cast<const float4*>(fu.amask + i * sa_i + kk);
                    a4[0] *= lrelu_slope(m.x); a4[1] *= lrelu_slope(m.y); a4[2] *= lrelu_slope(m.z); a4[3] *= lrelu_slope(m.w);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (kk + s < ke) {
                        const size_t o = i * sa_i + (kk + s) * sa_k;
                        a4[s] = A[o];
                        if (fu.amask) a4[s] *= lrelu_slope(fu.amask[o]);
                    }
            }
        }
        if (j < N) {
            if (vb && kk + 3 < ke) {
                const float4 t = *reinterpret_cast<const float4*>(Bm + j * sb_j + kk);
                b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) if (kk + s < ke) b4[s] = Bm[(kk + s) * sb_k + j * sb_j];
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s], b4[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][q * 4 + r][l15] = acc[r];
    __syncthreads();
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    if (threadIdx.x < 256 && i0 + r < M && j0 + c < N) {
        const float s = ((red[0][r][c] + red[1][r][c]) + red[2][r][c]) + red[3][r][c];
        float v = alpha * s;
        if (fu.bias) v += fu.bscale * fu.bias[j0 + c];
        if (fu.act == SGX_ACT_LRELU) v = lrelu(v);
        Cm[(size_t)(i0 + r) * N + j0 + c] = v;
    }
    if (fu.colsum && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x < 16 && i0 + (int)threadIdx.x < M) {
        const int row = i0 + threadIdx.x;                     // sum over the whole K axis of row `row` of op(A) (masked)
        float t = 0.f;
        for (int k = 0; k < K; ++k) {
            const size_t o = row * sa_i + k * sa_k;
            t += fu.amask ? A[o] * lrelu_slope(fu.amask[o]) : A[o];
        }
        fu.colsum[row] = fu.cscale * t;
    }
}
__global__ void gemm_splitk_reduce(const float* __restrict__ part, float* __restrict__ Cm, int MN, int splits, float alpha) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MN) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += part[(size_t)z * MN + i];     // fixed order: deterministic
    Cm[i] = alpha * s;
}
// Skinny problems (M = batch: few 16x16 output tiles) with a long K are split over blockIdx.z so that the weight
// matrix is streamed by many CUs instead of a handful.
static int gemm_splits(int M, int N, int K) {
    const int tiles = ((M + 15) / 16) * ((N + 15) / 16);
    if (tiles >= 256 || K < 1024) return 1;
    int s = K / 256, cap = (512 + tiles - 1) / tiles;
    if (s > cap) s = cap;
    return s < 1 ? 1 : s;
}
extern "C" size_t sgx_gemm_ws_bytes(int M, int N, int K) {
    const int s = gemm_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}
static int gemm_launch(const float* A, const float* Bm, float* C, int M, int N, int K, int ta, int tb, float alpha, GemmFuse fu, void* ws,
                       size_t ws_bytes, hipStream_t st) {
    SGX_REQUIRE(M > 0 && N > 0 && K > 0, SGX_EINVAL, "gemm: bad shape %d %d %d", M, N, K);
    int splits = gemm_splits(M, N, K);
    if (splits > 1 && (!ws || ws_bytes < (size_t)splits * M * N * sizeof(float) || fu.bias || fu.act || fu.colsum)) splits = 1;
    SGX_NOTE(2.0 * M * N * K, 4.0 * ((double)M * K + (double)K * N + (double)M * N), "gemm %dx%dx%d t%d%d%s", M, N, K, ta, tb,
             fu.amask || fu.bias || fu.colsum ? " fused" : "");
    if (splits == 1) {
        hipLaunchKernelGGL(gemm_f32_kernel, dim3((N + 15) / 16, (M + 15) / 16, 1), dim3(256), 0, st, A, Bm, C, M, N, K, ta, tb, alpha, fu);
        SGX_LAUNCH_CHECK("gemm_f32");
        return 0;
    }
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((N + 15) / 16, (M + 15) / 16, splits), dim3(256), 0, st, A, Bm, (float*)ws, M, N, K, ta, tb, 1.0f, fu);
    SGX_LAUNCH_CHECK("gemm_f32");
    hipLaunchKernelGGL(gemm_splitk_reduce, dim3((M * N + 255) / 256), dim3(256), 0, st, (const float*)ws, C, M * N, splits, alpha);
    SGX_LAUNCH_CHECK("gemm_splitk_reduce");
    return 0;
}
extern "C" int sgx_gemm_f32(const float* A, const float* Bm, float* C, int M, int N, int K, int ta, int tb, float alpha, void* ws,
                            size_t ws_bytes, void* stream) {
    return gemm_launch(A, Bm, C, M, N, K, ta, tb, alpha, GemmFuse{nullptr, nullptr, 0.f, SGX_ACT_NONE, nullptr, 0.f}, ws, ws_bytes,
                       (hipStream_t)stream);
}
// EqualizedLinear (models/CustomLayers.py:64-103) as three launches instead of nine:
//   forward : y[B][N] = act(w_mul * x[B][K] W[N][K]^T + b_mul * bias)
//   backward: gz = gy * slope(y) (if act) folded into the operand loads of both
//             gx[B][K] = w_mul * gz W        and        dW[N][K] = w_mul * gz^T x,  db[N] = b_mul * sum_b gz
extern "C" int sgx_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int N, int K, float w_mul, float b_mul,
                              int act, void* stream) {
    return gemm_launch(x, w, y, B, N, K, 0, 1, w_mul, GemmFuse{nullptr, bias, b_mul, act, nullptr, 0.f}, nullptr, 0, (hipStream_t)stream);
}
extern "C" int sgx_linear_bwd_data(const float* gy, const float* y_act, const float* w, float* gx, int B, int N, int K, float w_mul,
                                   void* stream) {
    return gemm_launch(gy, w, gx, B, K, N, 0, 0, w_mul, GemmFuse{y_act, nullptr, 0.f, SGX_ACT_NONE, nullptr, 0.f}, nullptr, 0,
                       (hipStream_t)stream);
}
extern "C" int sgx_linear_bwd_param(const float* gy, const float* y_act, const float* x, float* dw, float* db, int B, int N, int K,
                                    float w_mul, float b_mul, void* stream) {
    return gemm_launch(gy, x, dw, N, K, B, 1, 0, w_mul, GemmFuse{y_act, nullptr, 0.f, SGX_ACT_NONE, db, b_mul}, nullptr, 0,
                       (hipStream_t)stream);
}

// ---------------------------------------------------------------- all style affines of a generator forward in one launch
// StyleMod.lin (models/CustomLayers.py:203-216) of every active layer: y_g[B][N_g] = w_mul_g * x_g[B][D] W_g[N_g][D]^T +
// b_mul_g * bias_g, with x_g = lm[layer_g] (the layer-major dlatents [L][B][D]).  18 small GEMMs (M = batch) are latency, not
// work: one launch forward, two backward.  tab: G rows of SGX_STYLE_ROW 64-bit words
//   [W, bias, N, y offset (elements, = B * sum of earlier N), first tile, w_mul bits, b_mul bits, layer]
#define STYLE_MAXB 32
__device__ __forceinline__ int style_group(const long long* __restrict__ tab, int G, unsigned tile) {
    int g = 0;
    while (g + 1 < G && (unsigned)tab[(g + 1) * SGX_STYLE_ROW + 4] <= tile) ++g;
    return g;
}
// block: 16 output columns x 16 k-lanes; x_g staged in LDS
template <int MB>
__global__ __launch_bounds__(256) void style_fwd_kernel(const float* __restrict__ lm, const long long* __restrict__ tab, float* __restrict__ y,
                                                        int G, int B, int D) {
    extern __shared__ float xs[];                                  // [B][D]
    const int g = style_group(tab, G, blockIdx.x);
    const long long* r = tab + (size_t)g * SGX_STYLE_ROW;
    const float* W = reinterpret_cast<const float*>(r[0]);
    const float* bias = reinterpret_cast<const float*>(r[1]);
    const int N = (int)r[2], layer = (int)r[7];
    const float w_mul = __uint_as_float((unsigned)r[5]), b_mul = __uint_as_float((unsigned)r[6]);
    const float* x = lm + (size_t)layer * B * D;
    for (int i = threadIdx.x; i < B * D / 4; i += 256) reinterpret_cast<float4*>(xs)[i] = reinterpret_cast<const float4*>(x)[i];
    __syncthreads();
    const int col = threadIdx.x >> 4, kl = threadIdx.x & 15;
    const int n = ((int)blockIdx.x - (int)r[4]) * 16 + col;
    float acc[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[b] = 0.f;
    if (n < N)
        for (int k = kl * 4; k < D; k += 64) {
            const float4 w = *reinterpret_cast<const float4*>(W + (size_t)n * D + k);
#pragma unroll
            for (int b = 0; b < MB; ++b)
                if (b < B) {
                    const float4 v = *reinterpret_cast<const float4*>(xs + b * D + k);
                    acc[b] += (w.x * v.x + w.y * v.y) + (w.z * v.z + w.w * v.w);
                }
        }
#pragma unroll
    for (int b = 0; b < MB; ++b)
        if (b < B) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) acc[b] += __shfl_xor(acc[b], o, 64);
        }
    if (n < N && kl == 0) {
        const float bb = bias ? b_mul * bias[n] : 0.f;
        float* yo = y + (size_t)r[3] + n;
#pragma unroll
        for (int b = 0; b < MB; ++b)
            if (b < B) yo[(size_t)b * N] = w_mul * acc[b] + bb;
    }
}
// data gradient: glm[layer][b][k] = w_mul * sum_n gy_g[b][n] W_g[n][k].  grid (D/64, G); the 4 waves split n, LDS reduce.
template <int MB>
__global__ __launch_bounds__(256) void style_bwd_data_kernel(const float* __restrict__ gy, const long long* __restrict__ tab,
                                                             float* __restrict__ glm, int G, int B, int D) {
    extern __shared__ float sh[];                                  // gy_g [B][N] then the wave partials [4][B][64]
    const int g = blockIdx.y;
    const long long* r = tab + (size_t)g * SGX_STYLE_ROW;
    const float* W = reinterpret_cast<const float*>(r[0]);
    const int N = (int)r[2], layer = (int)r[7];
    const float w_mul = __uint_as_float((unsigned)r[5]);
    const float* gyg = gy + (size_t)r[3];
    // gy_g transposed to [n][MB] (rows of images, zero beyond B): the MB values of one n are consecutive, so a lane fetches them with
    // MB / 4 broadcast ds_read_b128 instead of MB ds_read_b32 -- the loop is LDS-issue bound (batch 32: 293 us with the [b][n] image)
    for (int i = threadIdx.x; i < MB * N; i += 256) {
        const int n = i / MB, b = i % MB;
        sh[i] = b < B ? gyg[(size_t)b * N + n] : 0.f;
    }
    __syncthreads();
    const int kk = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
    float acc[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[b] = 0.f;
    if (kk < D) {
        // 16 independent weight loads in flight per lane, then the FMAs (a dependent one-load-per-iteration loop is pure latency)
        int n = wave;
        for (; n + 4 * 15 < N; n += 4 * 16) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = W[(size_t)(n + 4 * u) * D + kk];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float4* row = reinterpret_cast<const float4*>(sh + (size_t)(n + 4 * u) * MB);
#pragma unroll
                for (int q = 0; q < MB / 4; ++q) {
                    const float4 g4 = row[q];
                    acc[4 * q] += g4.x * w[u]; acc[4 * q + 1] += g4.y * w[u]; acc[4 * q + 2] += g4.z * w[u]; acc[4 * q + 3] += g4.w * w[u];
                }
            }
        }
        for (; n < N; n += 4) {
            const float w = W[(size_t)n * D + kk];
            const float4* row = reinterpret_cast<const float4*>(sh + (size_t)n * MB);
#pragma unroll
            for (int q = 0; q < MB / 4; ++q) {
                const float4 g4 = row[q];
                acc[4 * q] += g4.x * w; acc[4 * q + 1] += g4.y * w; acc[4 * q + 2] += g4.z * w; acc[4 * q + 3] += g4.w * w;
            }
        }
    }
    __syncthreads();
    float* part = sh;                                              // reuse: [4][B][64]
#pragma unroll
    for (int b = 0; b < MB; ++b)
        if (b < B) part[(wave * B + b) * 64 + (threadIdx.x & 63)] = acc[b];
    __syncthreads();
    for (int i = threadIdx.x; i < B * 64; i += 256) {
        const int b = i / 64, kl = i % 64, k = blockIdx.x * 64 + kl;
        if (k < D) {
            const float sum = (part[(0 * B + b) * 64 + kl] + part[(1 * B + b) * 64 + kl]) + (part[(2 * B + b) * 64 + kl] + part[(3 * B + b) * 64 + kl]);
            glm[((size_t)layer * B + b) * D + k] = w_mul * sum;
        }
    }
}
// parameter gradients: dW_g[n][k] = w_mul * sum_b gy_g[b][n] x_g[b][k],  db_g[n] = b_mul * sum_b gy_g[b][n].  Flat outputs
// dw[(sum of earlier N + n) * D + k], db[sum of earlier N + n]; block = 16 rows n x all k.
template <int MB>
__global__ __launch_bounds__(256) void style_bwd_param_kernel(const float* __restrict__ gy, const float* __restrict__ lm,
                                                              const long long* __restrict__ tab, float* __restrict__ dw,
                                                              float* __restrict__ db, int G, int B, int D) {
    __shared__ float gs[MB][16];
    const int g = style_group(tab, G, blockIdx.x);
    const long long* r = tab + (size_t)g * SGX_STYLE_ROW;
    const int N = (int)r[2], layer = (int)r[7];
    const float w_mul = __uint_as_float((unsigned)r[5]), b_mul = __uint_as_float((unsigned)r[6]);
    const int n0 = ((int)blockIdx.x - (int)r[4]) * 16;
    const size_t noff = (size_t)r[3] / B;                          // sum of earlier N
    const float* gyg = gy + (size_t)r[3];
    for (int i = threadIdx.x; i < B * 16; i += 256) {
        const int b = i / 16, j = i % 16;
        gs[b][j] = (n0 + j < N) ? gyg[(size_t)b * N + n0 + j] : 0.f;
    }
    __syncthreads();
    const float* x = lm + (size_t)layer * B * D;
    for (int k = threadIdx.x; k < D; k += 256) {
        float xv[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) xv[b] = b < B ? x[(size_t)b * D + k] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (n0 + j < N) {
                float s = 0.f;
#pragma unroll
                for (int b = 0; b < MB; ++b)
                    if (b < B) s += gs[b][j] * xv[b];
                dw[(noff + n0 + j) * D + k] = w_mul * s;
            }
    }
    if (db && threadIdx.x < 16 && n0 + (int)threadIdx.x < N) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += gs[b][threadIdx.x];
        db[noff + n0 + threadIdx.x] = b_mul * s;
    }
}
static int style_check(int G, int B, int D) {
    SGX_REQUIRE(G > 0 && B > 0 && B <= STYLE_MAXB && D > 0 && D % 64 == 0, SGX_EUNSUPPORTED, "style affines: G=%d B=%d D=%d", G, B, D);
    return 0;
}
extern "C" int sgx_style_fwd(const float* lm, const void* table, float* y, int G, int B, int D, int total_tiles, void* stream) {
    int rc = style_check(G, B, D); if (rc) return rc;
    SGX_NOTE(0.0, 0.0, "style_fwd G%d B%d", G, B);
    if (B <= 4) hipLaunchKernelGGL(style_fwd_kernel<4>, dim3(total_tiles), dim3(256), (size_t)B * D * sizeof(float), (hipStream_t)stream, lm, (const long long*)table, y, G, B, D);
    else if (B <= 8) hipLaunchKernelGGL(style_fwd_kernel<8>, dim3(total_tiles), dim3(256), (size_t)B * D * sizeof(float), (hipStream_t)stream, lm, (const long long*)table, y, G, B, D);
    else if (B <= 16) hipLaunchKernelGGL(style_fwd_kernel<16>, dim3(total_tiles), dim3(256), (size_t)B * D * sizeof(float), (hipStream_t)stream, lm, (const long long*)table, y, G, B, D);
    else hipLaunchKernelGGL(style_fwd_kernel<32>, dim3(total_tiles), dim3(256), (size_t)B * D * sizeof(float), (hipStream_t)stream, lm, (const long long*)table, y, G, B, D);
    SGX_LAUNCH_CHECK("style_fwd");
    return 0;
}
extern "C" int sgx_style_bwd_data(const float* gy, const void* table, float* glm, int G, int B, int D, int max_n, void* stream) {
    int rc = style_check(G, B, D); if (rc) return rc;
    const int mb = B <= 4 ? 4 : (B <= 8 ? 8 : (B <= 16 ? 16 : 32));
    const size_t a = (size_t)mb * max_n, b = (size_t)4 * B * 64;  // the transposed gy image [max_n][MB], then the wave partials
    SGX_NOTE(0.0, 0.0, "style_bwd_data G%d B%d", G, B);
    if (B <= 4) hipLaunchKernelGGL(style_bwd_data_kernel<4>, dim3(D / 64, G), dim3(256), (a > b ? a : b) * sizeof(float), (hipStream_t)stream, gy, (const long long*)table, glm, G, B, D);
    else if (B <= 8) hipLaunchKernelGGL(style_bwd_data_kernel<8>, dim3(D / 64, G), dim3(256), (a > b ? a : b) * sizeof(float), (hipStream_t)stream, gy, (const long long*)table, glm, G, B, D);
    else if (B <= 16) hipLaunchKernelGGL(style_bwd_data_kernel<16>, dim3(D / 64, G), dim3(256), (a > b ? a : b) * sizeof(float), (hipStream_t)stream, gy, (const long long*)table, glm, G, B, D);
    else hipLaunchKernelGGL(style_bwd_data_kernel<32>, dim3(D / 64, G), dim3(256), (a > b ? a : b) * sizeof(float), (hipStream_t)stream, gy, (const long long*)table, glm, G, B, D);
    SGX_LAUNCH_CHECK("style_bwd_data");
    return 0;
}
extern "C" int sgx_style_bwd_param(const float* gy, const float* lm, const void* table, float* dw, float* db, int G, int B, int D,
                                   int total_tiles, void* stream) {
    int rc = style_check(G, B, D); if (rc) return rc;
    SGX_NOTE(0.0, 0.0, "style_bwd_param G%d B%d", G, B);
    if (B <= 4) hipLaunchKernelGGL(style_bwd_param_kernel<4>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, gy, lm, (const long long*)table, dw, db, G, B, D);
    else if (B <= 8) hipLaunchKernelGGL(style_bwd_param_kernel<8>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, gy, lm, (const long long*)table, dw, db, G, B, D);
    else if (B <= 16) hipLaunchKernelGGL(style_bwd_param_kernel<16>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, gy, lm, (const long long*)table, dw, db, G, B, D);
    else hipLaunchKernelGGL(style_bwd_param_kernel<32>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, gy, lm, (const long long*)table, dw, db, G, B, D);
    SGX_LAUNCH_CHECK("style_bwd_param");
    return 0;
}

// ---------------------------------------------------------------- R1 penalty head: out[0] = sum(x^2)   (models/Losses.py:210)
// and its backward  out = alpha * s[0] * x  with the upstream scalar s kept on the device.
#define SUMSQ_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_f32_stage1(const float* __restrict__ x, size_t n, double* __restrict__ partial) {
    __shared__ double sh[16];
    const size_t nvec = n / 4;
    double acc = 0.0;
    float part = 0.f; int cnt = 0;
    const size_t gs = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * gs < nvec; i += 4 * gs) {                      // four 16-byte loads in flight per lane
        const float4 v0 = reinterpret_cast<const float4*>(x)[i], v1 = reinterpret_cast<const float4*>(x)[i + gs];
        const float4 v2 = reinterpret_cast<const float4*>(x)[i + 2 * gs], v3 = reinterpret_cast<const float4*>(x)[i + 3 * gs];
        part += ((v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w)) + ((v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w));
        part += ((v2.x * v2.x + v2.y * v2.y) + (v2.z * v2.z + v2.w * v2.w)) + ((v3.x * v3.x + v3.y * v3.y) + (v3.z * v3.z + v3.w * v3.w));
        if (++cnt == 4) { acc += (double)part; part = 0.f; cnt = 0; }
    }
    for (; i < nvec; i += gs) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        part += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nvec * 4; i < n; ++i) part += x[i] * x[i];
    acc += (double)part;
    const double s = block_sum_d(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_f32_stage2(const double* __restrict__ partial, int nblk, float* __restrict__ out) {
    __shared__ double sh[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += partial[i];
    const double s = block_sum_d(acc, sh);
    if (threadIdx.x == 0) out[0] = (float)s;
}
extern "C" size_t sgx_sumsq_ws_bytes(void) { return SUMSQ_BLOCKS * sizeof(double); }
extern "C" int sgx_sumsq_f32(const float* x, size_t n, void* ws, size_t ws_bytes, float* out, void* stream) {
    SGX_REQUIRE(ws_bytes >= SUMSQ_BLOCKS * sizeof(double), SGX_EWORKSPACE, "sumsq: workspace");
    int nblk = (int)((n / 4 + 255) / 256);
    if (nblk > SUMSQ_BLOCKS) nblk = SUMSQ_BLOCKS;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(sumsq_f32_stage1, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, n, (double*)ws);
    SGX_LAUNCH_CHECK("sumsq_stage1");
    hipLaunchKernelGGL(sumsq_f32_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, nblk, out);
    SGX_LAUNCH_CHECK("sumsq_stage2");
    return 0;
}
// ---- logistic GAN loss heads (models/Losses.py:213-229) in one launch: the scalar loss AND its derivative w.r.t. every logit
//   discriminator: loss = scale * (mean softplus(fake) + mean softplus(-real)),  d/dfake = scale * sigmoid(fake) / n_fake,
//                                                                                   d/dreal = -scale * sigmoid(-real) / n_real
//   generator:     loss = scale * mean softplus(-fake),                            d/dfake = -scale * sigmoid(-fake) / n_fake
// softplus as torch computes it (threshold 20: beyond it the identity).  One wave: a batch is at most a few hundred logits.
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__global__ void logistic_loss_kernel(const float* __restrict__ fake, int nf, const float* __restrict__ real, int nr, float scale, int gen,
                                     float* __restrict__ loss, float* __restrict__ gfake, float* __restrict__ greal) {
    const int l = threadIdx.x;
    float sf = 0.f, sr = 0.f;
    const float sgn = gen ? -1.f : 1.f;                     // the generator wants its fakes called real
    for (int i = l; i < nf; i += 64) {
        const float x = sgn * fake[i];
        sf += softplus_f(x);
        gfake[i] = sgn * scale * sigmoid_f(x) / (float)nf;
    }
    for (int i = l; i < nr; i += 64) {
        const float x = -real[i];
        sr += softplus_f(x);
        greal[i] = -scale * sigmoid_f(x) / (float)nr;
    }
    sf = wave_sum(sf); sr = wave_sum(sr);
    if (l == 0) loss[0] = scale * (sf / (float)(nf > 0 ? nf : 1) + (nr > 0 ? sr / (float)nr : 0.f));
}
extern "C" int sgx_logistic_loss(const float* fake, int n_fake, const float* real, int n_real, float scale, int generator, float* loss,
                                 float* g_fake, float* g_real, void* stream) {
    SGX_REQUIRE(fake && n_fake > 0 && loss && g_fake, SGX_EINVAL, "logistic_loss: fake logits, loss and g_fake are required");
    SGX_REQUIRE(generator ? n_real == 0 : (real && g_real && n_real > 0), SGX_EINVAL, "logistic_loss: real logits %s", generator ? "are not part of the generator loss" : "are required");
    hipLaunchKernelGGL(logistic_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, fake, n_fake, real, n_real, scale, generator, loss, g_fake, g_real);
    SGX_LAUNCH_CHECK("logistic_loss");
    return 0;
}
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s, float alpha, float* __restrict__ out, size_t n) {
    const float k = alpha * s[0];
    const size_t nvec = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x *= k; v.y *= k; v.z *= k; v.w *= k;
        reinterpret_cast<float4*>(out)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nvec * 4; i < n; ++i) out[i] = k * x[i];
}
extern "C" int sgx_scale_dev_f32(const float* x, const float* s, float alpha, float* out, size_t n, void* stream) {
    size_t g = (n / 4 + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(scale_dev_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, s, alpha, out, n);
    SGX_LAUNCH_CHECK("scale_dev");
    return 0;
}
