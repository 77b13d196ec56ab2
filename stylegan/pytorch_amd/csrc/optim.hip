// Multi-tensor optimizer step for the G and D parameter sets (fp32 master parameters):
// global-norm gradient clip (models/GAN.py:651), Adam (models/GAN.py:529-533,616-618,652) and the generator EMA
// (models/__init__.py:31-36).  ~150 small tensors per step: one launch per pass instead of 3-4 per tensor.
#include "common.h"

#define MT_BLOCKS_X 32

// partial[t*MT_BLOCKS_X + bx] = sum of squares of a slice of tensor t (double) ; deterministic two-stage reduce
__global__ __launch_bounds__(256) void sumsq_stage1(const float* const* __restrict__ grads, const int64_t* __restrict__ sizes,
                                                    double* __restrict__ partial) {
    __shared__ double sh[4];
    const int t = blockIdx.y;
    const float* g = grads[t];
    const int64_t n = sizes[t];
    double acc = 0.0;
    float part = 0.f; int cnt = 0;
    const int64_t gs = (int64_t)MT_BLOCKS_X * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {            // 16-byte loads, four in flight per lane
        const int64_t nv = n / 4;
        const float4* g4 = reinterpret_cast<const float4*>(g);
        for (; i + 3 * gs < nv; i += 4 * gs) {
            const float4 a = g4[i], b = g4[i + gs], c = g4[i + 2 * gs], d = g4[i + 3 * gs];
            part += ((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w)) + ((b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w));
            part += ((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w)) + ((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
            if (++cnt == 4) { acc += part; part = 0.f; cnt = 0; }
        }
        for (; i < nv; i += gs) {
            const float4 a = g4[i];
            part += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int64_t k = nv * 4; k < n; ++k) part += g[k] * g[k];
    } else {
        for (; i < n; i += gs) {
            const float v = g[i];
            part += v * v;
            if (++cnt == 32) { acc += part; part = 0.f; cnt = 0; }
        }
    }
    acc += part;
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)t * MT_BLOCKS_X + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// out[0] = total sum of squares; out[1] = clip coefficient min(1, max_norm / (sqrt(total) + 1e-6))
__global__ void sumsq_stage2(const double* __restrict__ partial, int count, float max_norm, float* __restrict__ out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) acc += partial[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        out[0] = (float)tot;
        const float norm = (float)sqrt(tot);
        const float coef = max_norm / (norm + 1e-6f);
        out[1] = coef < 1.f ? coef : 1.f;
    }
}
extern "C" int sgx_gradnorm_clip_coef(const float* const* grads, const int64_t* sizes, int n, float max_norm, double* partial,
                                      float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(n > 0, SGX_EINVAL, "gradnorm: n");
    hipLaunchKernelGGL(sumsq_stage1, dim3(MT_BLOCKS_X, n), dim3(256), 0, st, grads, sizes, partial);
    SGX_LAUNCH_CHECK("sumsq_stage1");
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, st, (const double*)partial, n * MT_BLOCKS_X, max_norm, out);
    SGX_LAUNCH_CHECK("sumsq_stage2");
    return 0;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(float* const* __restrict__ params, const float* const* __restrict__ grads,
                                                         float* const* __restrict__ exp_avg, float* const* __restrict__ exp_avg_sq,
                                                         const int64_t* __restrict__ sizes, float beta1, float beta2, float eps,
                                                         const float* __restrict__ step_sizes, const float* __restrict__ bc2_sqrts,
                                                         const float* __restrict__ grad_scale) {
    const int t = blockIdx.y;
    const float step_size = step_sizes[t], bc2_sqrt = bc2_sqrts[t];
    float* p = params[t]; const float* g = grads[t]; float* m = exp_avg[t]; float* v = exp_avg_sq[t];
    const int64_t n = sizes[t];
    const float gs = grad_scale ? grad_scale[0] : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * gs;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}
extern "C" int sgx_adam_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                              const int64_t* sizes, int n, float beta1, float beta2, float eps, const float* step_sizes,
                              const float* bc2_sqrts, const float* grad_scale, void* stream) {
    SGX_REQUIRE(n > 0, SGX_EINVAL, "adam: n=%d", n);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(256, n), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       sizes, beta1, beta2, eps, step_sizes, bc2_sqrts, grad_scale);
    SGX_LAUNCH_CHECK("adam_multi");
    return 0;
}

__global__ __launch_bounds__(256) void ema_multi_kernel(float* const* __restrict__ tgt, const float* const* __restrict__ src,
                                                        const int64_t* __restrict__ sizes, float beta) {
    const int t = blockIdx.y;
    float* a = tgt[t]; const float* b = src[t];
    const int64_t n = sizes[t];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        a[i] = beta * a[i] + (1.f - beta) * b[i];
}
extern "C" int sgx_ema_multi(float* const* tgt, const float* const* src, const int64_t* sizes, int n, float beta, void* stream) {
    SGX_REQUIRE(n > 0, SGX_EINVAL, "ema: n");
    hipLaunchKernelGGL(ema_multi_kernel, dim3(256, n), dim3(256), 0, (hipStream_t)stream, tgt, src, sizes, beta);
    SGX_LAUNCH_CHECK("ema_multi");
    return 0;
}
