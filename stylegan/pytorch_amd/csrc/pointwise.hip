// HBM-bound pieces of the StyleGAN layers (NHWC): bias/activation, LeakyReLU backward, fade-in lerp, depthwise blur,
// 2x2 pooling, nearest upsample, per-channel sums, 1x1 RGB convolutions.  One 16-byte vector per lane per access;
// grid-stride loops capped at 256 CUs x 8 blocks.
#include "common.h"

static inline unsigned grid_for(size_t nvec, int block = 256) {
    size_t g = (nvec + block - 1) / block;
    static const size_t cap = [] { const char* e = getenv("SGX_GRID_CAP"); const long v = e ? atol(e) : 8192; return (size_t)(v > 0 ? v : 0x7fffffffL); }();
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}
// The grid of a pass whose threads have NO per-thread setup worth amortising (pure elementwise kernels, the blur's strips): one trip of the
// kernel's grid-stride loop per thread.  Round 6, tools/stream_probe.hip + the step's layer tables with SGX_GRID_CAP=0: a read + write stream
// runs at 6.1-6.2 TB/s as short-lived blocks against 5.0-5.1 under the 8192-block cap above (lrelu_bwd_bits 214 -> 184 us, lrelu_bwd 148 ->
// 126, up2+add 164 -> 146, the blurs -2..4 %); kernels that hoist weights per thread (rgb_in / rgb_out / fade_rgb_bwd2) lose and keep the cap.
// SGX_GRID_ALL_CAP (A/B): the cap of these launches, default none.
static inline unsigned grid_all(size_t nvec, int block = 256) {
    size_t g = (nvec + block - 1) / block;
    static const size_t cap = [] { const char* e = getenv("SGX_GRID_ALL_CAP"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 0x7fffffffL); }();
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---------------------------------------------------------------- y = act(x + bias[c])
template <typename T>
__global__ void bias_act_kernel(const T* __restrict__ x, const float* __restrict__ bias, float bscale, T* __restrict__ y,
                                size_t nvec, int C, int act) {
    constexpr int VE = VecTraits<T>::VE;
    const int cv = C / VE;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float v[VE];
        VecTraits<T>::load(x + i * VE, v);
        const int c0 = (int)(i % cv) * VE;
#pragma unroll
        for (int j = 0; j < VE; ++j) {
            float t = v[j] + (bias ? bscale * bias[c0 + j] : 0.f);
            v[j] = act_apply(t, act);
        }
        VecTraits<T>::store(y + i * VE, v);
    }
}
__global__ void bias_act_scalar_kernel(const float* __restrict__ x, const float* __restrict__ bias, float bscale, float* __restrict__ y,
                                       size_t n, int C, int act) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float t = x[i] + (bias ? bscale * bias[i % C] : 0.f);
        y[i] = act_apply(t, act);
    }
}
extern "C" int sgx_bias_act(const void* x, const float* bias, float bscale, void* y, size_t npix, int C, int act, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(act >= SGX_ACT_NONE && act <= SGX_ACT_RELU, SGX_EINVAL, "bias_act: activation %d", act);
    SGX_NOTE(0.0, 2.0 * (dtype == SGX_F32 ? 4.0 : 2.0) * npix * C, "bias_act %zux%d", npix, C);
    if (dtype == SGX_F32 && C % 4 != 0) {                    // e.g. the [B,1] discriminator output
        hipLaunchKernelGGL(bias_act_scalar_kernel, dim3(grid_for(npix * C)), dim3(256), 0, st, (const float*)x, bias, bscale, (float*)y, npix * C, C, act);
        SGX_LAUNCH_CHECK("bias_act_scalar");
        return 0;
    }
    if (dtype == SGX_F32) {
        SGX_REQUIRE(C % 4 == 0, SGX_EUNSUPPORTED, "bias_act: C %% 4");
        size_t nvec = npix * C / 4;
        hipLaunchKernelGGL(bias_act_kernel<float>, dim3(grid_all(nvec)), dim3(256), 0, st, (const float*)x, bias, bscale, (float*)y, nvec, C, act);
    } else {
        SGX_REQUIRE(C % 8 == 0, SGX_EUNSUPPORTED, "bias_act: C %% 8");
        size_t nvec = npix * C / 8;
        hipLaunchKernelGGL(bias_act_kernel<bf16_t>, dim3(grid_all(nvec)), dim3(256), 0, st, (const bf16_t*)x, bias, bscale, (bf16_t*)y, nvec, C, act);
    }
    SGX_LAUNCH_CHECK("bias_act");
    return 0;
}

// ---------------------------------------------------------------- dx = scale * dy * (y > 0 ? 1 : slope)   (slope 0.2: LeakyReLU; 0: ReLU)
// scale (1 = exact no-op): the fade-in coefficient of the branch the activation sits on, so that "lerp backward, then
// activation backward" on the discriminator's newest block is one pass
template <typename T>
__global__ void lrelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, size_t nvec, float slope, float scale,
                                 const float* __restrict__ scale_dev) {
    if (scale_dev) scale = scale_dev[0];
    constexpr int VE = VecTraits<T>::VE;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float g[VE], a[VE];
        VecTraits<T>::load(dy + i * VE, g);
        VecTraits<T>::load(y + i * VE, a);
#pragma unroll
        for (int j = 0; j < VE; ++j) g[j] = (scale * g[j]) * (a[j] > 0.f ? 1.f : slope);
        VecTraits<T>::store(dx + i * VE, g);
    }
}
extern "C" int sgx_lrelu_bwd(const void* dy, const void* y, void* dx, size_t n, float slope, float scale, const float* scale_dev, int dtype,
                             void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_NOTE(0.0, 3.0 * (dtype == SGX_F32 ? 4.0 : 2.0) * n, "lrelu_bwd %zu", n);
    if (dtype == SGX_F32) {
        SGX_REQUIRE(n % 4 == 0, SGX_EUNSUPPORTED, "lrelu_bwd: n %% 4");
        hipLaunchKernelGGL(lrelu_bwd_kernel<float>, dim3(grid_all(n / 4)), dim3(256), 0, st, (const float*)dy, (const float*)y, (float*)dx, n / 4, slope, scale, scale_dev);
    } else {
        SGX_REQUIRE(n % 8 == 0, SGX_EUNSUPPORTED, "lrelu_bwd: n %% 8");
        hipLaunchKernelGGL(lrelu_bwd_kernel<bf16_t>, dim3(grid_all(n / 8)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dx, n / 8, slope, scale, scale_dev);
    }
    SGX_LAUNCH_CHECK("lrelu_bwd");
    return 0;
}

// the same with the activation given as SIGN BITS (bits[i] = the signs of the 8 channels of vector i, as sgx_conv3x3_signbits /
// sgx_conv4x4s2_down_fade write them): the activation itself need not exist
__global__ void lrelu_bwd_bits_kernel(const bf16_t* __restrict__ dy, const unsigned char* __restrict__ bits, bf16_t* __restrict__ dx, size_t nvec, float slope,
                                      float scale, const float* __restrict__ scale_dev, int nt) {
    if (scale_dev) scale = scale_dev[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        float g[8];
        const uint4 raw = ld16(dy + i * 8, nt != 0);                // (nt: both streams of a >= 192 MB tensor with the nontemporal hint)
        const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { g[2 * k] = __uint_as_float(w4[k] << 16); g[2 * k + 1] = __uint_as_float(w4[k] & 0xffff0000u); }
        const unsigned bb = bits[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = (scale * g[j]) * (((bb >> j) & 1u) ? 1.f : slope);
        st16(dx + i * 8, make_uint4(pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3]), pack_bf16x2(g[4], g[5]), pack_bf16x2(g[6], g[7])), nt != 0);
    }
}
extern "C" int sgx_lrelu_bwd_bits(const void* dy, const void* bits, void* dx, size_t n, float slope, float scale, const float* scale_dev, int dtype,
                                  void* stream) {
    SGX_REQUIRE(dtype == SGX_BF16 && n % 8 == 0, SGX_EUNSUPPORTED, "lrelu_bwd_bits: bf16, n %% 8 == 0");
    SGX_REQUIRE(dy && bits && dx, SGX_EINVAL, "lrelu_bwd_bits: null argument");
    SGX_NOTE(0.0, 4.125 * n, "lrelu_bwd_bits %zu", n);
    hipLaunchKernelGGL(lrelu_bwd_bits_kernel, dim3(grid_all(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const unsigned char*)bits, (bf16_t*)dx,
                       n / 8, slope, scale, scale_dev, sgx_nt_for(2.0 * (double)n) ? 1 : 0);
    SGX_LAUNCH_CHECK("lrelu_bwd_bits");
    return 0;
}

// ---------------------------------------------------------------- backward of the fade-in tail with the residual computed in the store
// Forward (sgx_conv4x4s2_down_fade_rgb): out = a * lrelu(z) + b * (rb' + ws * pimg . W^T), models/GAN.py:423-427.  Given g = dL/dout:
//   gy[p][c]    = (a * g[p][c]) * slope(bits[p][c])             bit for bit sgx_lrelu_bwd_bits (the stride-2 convolution's upstream gradient)
//   dW[c][j]   += b * ws * sum_p g[p][c] * pimg[p][j],   drb[c] += b * bs * sum_p g[p][c]          (from_rgb's parameters)
//   gpimg[p][j] = b * ws * sum_c g[p][c] * W[c][j]                                                  (optional: the image gradient)
// in ONE pass over g instead of four (lrelu_bwd_bits, rgb_wgrad, colsum, rgb_out).  CV = C / 8 lanes per pixel; a block walks a
// contiguous pixel range; fp32 partial sums per thread (flushed to fp64 every 64 pixels), fp64 across lanes, waves and blocks in a
// fixed order: deterministic, no atomics.
#define FRB_BLOCKS 2048
template <int CV>
__global__ __launch_bounds__(256) void fade_rgb_bwd_kernel(const bf16_t* __restrict__ g, const unsigned char* __restrict__ bits, const float* __restrict__ pimg,
                                                           const float* __restrict__ wr, float ws, float alpha, float beta, const float* __restrict__ ab_dev,
                                                           bf16_t* __restrict__ gy, float* __restrict__ gpimg, double* __restrict__ part, size_t npix, int nt) {
    constexpr int PPI = 256 / CV;                            // pixels per block iteration
    __shared__ double sh[4][CV][32];
    if (ab_dev) { alpha = ab_dev[0]; beta = ab_dev[1]; }
    const int tid = threadIdx.x, v = tid % CV, pl = tid / CV, lane = tid & 63, wave = tid >> 6;
    const size_t per = (npix + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    float wv[8][3];
    if (gpimg) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) wv[j][k] = (beta * ws) * wr[(v * 8 + j) * 3 + k];
    }
    double acc[8][4];
    float q[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[j][k] = 0.0; q[j][k] = 0.f; }
    int cnt = 0;
    for (size_t pb = p0; pb < p1; pb += PPI) {               // (block-uniform trip count: the shuffles below run converged)
        const size_t p = pb + pl;
        const bool ok = p < p1;
        float gv[8];
        float i0 = 0.f, i1 = 0.f, i2 = 0.f;
        unsigned bb = 0;
        if (ok) {
            {   // (nt: both streams of a >= 192 MB tensor with the nontemporal hint, see ld16 / st16)
                const uint4 raw = ld16(g + (p * CV + v) * 8, nt != 0);
                const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) { gv[2 * i] = __uint_as_float(w4[i] << 16); gv[2 * i + 1] = __uint_as_float(w4[i] & 0xffff0000u); }
            }
            bb = bits[p * CV + v];
            i0 = pimg[p * 3]; i1 = pimg[p * 3 + 1]; i2 = pimg[p * 3 + 2];
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (alpha * gv[j]) * (((bb >> j) & 1u) ? 1.f : SGX_LRELU);
            st16(gy + (p * CV + v) * 8, make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])), nt != 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { q[j][0] += gv[j] * i0; q[j][1] += gv[j] * i1; q[j][2] += gv[j] * i2; q[j][3] += gv[j]; }
        if (++cnt == 64) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) { acc[j][k] += (double)q[j][k]; q[j][k] = 0.f; }
            cnt = 0;
        }
        if (gpimg) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0 += gv[j] * wv[j][0]; s1 += gv[j] * wv[j][1]; s2 += gv[j] * wv[j][2]; }
#pragma unroll
            for (int o = 1; o < CV; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            if (ok && v == 0) { gpimg[p * 3] = s0; gpimg[p * 3 + 1] = s1; gpimg[p * 3 + 2] = s2; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double t = acc[j][k] + (double)q[j][k];
#pragma unroll
            for (int o = CV; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);          // lanes of one channel vector: lane % CV
            if (lane < CV) sh[wave][lane][j * 4 + k] = t;
        }
    __syncthreads();
    for (int e = tid; e < CV * 32; e += 256) {
        const int vv = e / 32, r = e % 32;
        part[((size_t)blockIdx.x * CV + vv) * 32 + r] = (sh[0][vv][r] + sh[1][vv][r]) + (sh[2][vv][r] + sh[3][vv][r]);
    }
}
// part[blk][C/8][8 channels][4] -> dW[c][j] (j < 3, from_rgb layout [C][3]) and drb[c] (j == 3); acc bit 0 / 1: accumulate into dW / drb
__global__ __launch_bounds__(256) void fade_rgb_bwd_finish(const double* __restrict__ part, float* __restrict__ dw, float* __restrict__ drb, int nblk, int C,
                                                           float ws, float bs, float beta, const float* __restrict__ ab_dev, int acc) {
    __shared__ double sh[64][5];
    if (ab_dev) beta = ab_dev[1];
    const int el = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const int e = blockIdx.x * 4 + el;                       // e = c * 4 + j
    double s = 0.0;
    if (e < 4 * C)
        for (int b = pl; b < nblk; b += 64) s += part[(size_t)b * 4 * C + e];
    sh[pl][el] = s;
    __syncthreads();
    if (pl == 0 && e < 4 * C) {
        double t = 0.0;
        for (int k = 0; k < 64; ++k) t += sh[k][el];
        const int c = e >> 2, j = e & 3;
        if (j < 3) {
            if (dw) { const float val = (float)(t * ((double)beta * (double)ws)); dw[c * 3 + j] = (acc & 1) ? dw[c * 3 + j] + val : val; }
        } else if (drb) {
            const float val = (float)(t * ((double)beta * (double)bs));
            drb[c] = (acc & 2) ? drb[c] + val : val;
        }
    }
}
extern "C" size_t sgx_fade_rgb_bwd_ws_bytes(size_t npix, int C) { (void)npix; return (size_t)FRB_BLOCKS * 4 * C * sizeof(double); }
extern "C" int sgx_fade_rgb_bwd(const void* g, const void* bits, const float* pimg, const float* wr, float ws, float bs, float alpha, float beta,
                                const float* ab_dev, void* gy, float* dwr, float* drb, int acc, float* gpimg, void* wsbuf, size_t ws_bytes, size_t npix, int C,
                                int dtype, void* stream) {
    SGX_REQUIRE(dtype == SGX_BF16 && (C == 32 || C == 64 || C == 128), SGX_EUNSUPPORTED, "fade_rgb_bwd: bf16, C in {32, 64, 128} (C=%d dtype %d)", C, dtype);
    SGX_REQUIRE(g && bits && pimg && wr && gy && wsbuf && npix > 0, SGX_EINVAL, "fade_rgb_bwd: null argument");
    SGX_REQUIRE(ws_bytes >= sgx_fade_rgb_bwd_ws_bytes(npix, C), SGX_EWORKSPACE, "fade_rgb_bwd: workspace");
    SGX_NOTE(8.0 * npix * C, npix * (4.125 * C + 12.0 + (gpimg ? 12.0 : 0.0)), "fade_rgb_bwd %zux%d", npix, C);
    hipStream_t st = (hipStream_t)stream;
    const int ppi = 256 / (C / 8);
    long nblk = (long)((npix + (size_t)ppi * 8 - 1) / ((size_t)ppi * 8));
    if (nblk > FRB_BLOCKS) nblk = FRB_BLOCKS;
    if (nblk < 1) nblk = 1;
    const bf16_t* gp = static_cast<const bf16_t*>(g);
    const unsigned char* bp = static_cast<const unsigned char*>(bits);
    bf16_t* yp = static_cast<bf16_t*>(gy);
    double* part = static_cast<double*>(wsbuf);
    const int nt = sgx_nt_for(2.0 * (double)npix * C) ? 1 : 0;
    if (C == 32) hipLaunchKernelGGL(fade_rgb_bwd_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, st, gp, bp, pimg, wr, ws, alpha, beta, ab_dev, yp, gpimg, part, npix, nt);
    else if (C == 64) hipLaunchKernelGGL(fade_rgb_bwd_kernel<8>, dim3((unsigned)nblk), dim3(256), 0, st, gp, bp, pimg, wr, ws, alpha, beta, ab_dev, yp, gpimg, part, npix, nt);
    else hipLaunchKernelGGL(fade_rgb_bwd_kernel<16>, dim3((unsigned)nblk), dim3(256), 0, st, gp, bp, pimg, wr, ws, alpha, beta, ab_dev, yp, gpimg, part, npix, nt);
    SGX_LAUNCH_CHECK("fade_rgb_bwd_kernel");
    if (dwr || drb) {
        hipLaunchKernelGGL(fade_rgb_bwd_finish, dim3((unsigned)((4 * C + 3) / 4)), dim3(256), 0, st, part, dwr, drb, (int)nblk, C, ws, bs, beta, ab_dev, acc);
        SGX_LAUNCH_CHECK("fade_rgb_bwd_finish");
    }
    return 0;
}
// The parameter-gradient half of sgx_fade_rgb_bwd on its own: sums the block partials a call with dwr = drb = NULL left in `wsbuf` and
// writes / accumulates from_rgb's gradients.  The caller orders it behind that call and may run it on another stream -- the one every
// other accumulation into .grad of the step runs on, so that two backward branches never read-modify-write the same gradient unordered.
extern "C" int sgx_fade_rgb_bwd_finish(const void* wsbuf, size_t ws_bytes, size_t npix, int C, float ws, float bs, float beta, const float* ab_dev,
                                       float* dwr, float* drb, int acc, void* stream) {
    SGX_REQUIRE(C == 32 || C == 64 || C == 128, SGX_EUNSUPPORTED, "fade_rgb_bwd_finish: C in {32, 64, 128} (C=%d)", C);
    SGX_REQUIRE(wsbuf && npix > 0 && (dwr || drb), SGX_EINVAL, "fade_rgb_bwd_finish: null argument");
    SGX_REQUIRE(ws_bytes >= sgx_fade_rgb_bwd_ws_bytes(npix, C), SGX_EWORKSPACE, "fade_rgb_bwd_finish: workspace");
    const int ppi = 256 / (C / 8);
    long nblk = (long)((npix + (size_t)ppi * 8 - 1) / ((size_t)ppi * 8));      // as sgx_fade_rgb_bwd
    if (nblk > FRB_BLOCKS) nblk = FRB_BLOCKS;
    if (nblk < 1) nblk = 1;
    SGX_NOTE(0.0, (double)nblk * 4 * C * sizeof(double), "fade_rgb_bwd_finish C%d", C);
    hipLaunchKernelGGL(fade_rgb_bwd_finish, dim3((unsigned)((4 * C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, static_cast<const double*>(wsbuf), dwr, drb,
                       (int)nblk, C, ws, bs, beta, ab_dev, acc);
    SGX_LAUNCH_CHECK("fade_rgb_bwd_finish");
    return 0;
}

// The adjoint of sgx_fade_rgb_bwd's data half -- what the R1 double backward needs of the newest block's tail -- in ONE pass:
//   out[p][c] = bf16( t1 + t2 ),  t1 = bf16((alpha ggy[p][c]) slope(bits[p][c]))            (bit for bit sgx_lrelu_bwd_bits on ggy)
//                                 t2 = bf16(sum_j ggp[p][j] (ws W[c][j]))                       (bit for bit sgx_rgb_in on ggp, no bias)
//                                      [ab_dev: then bf16(beta t2), the unfused path's scaling pass]
// i.e. the three passes of the differentiable composition (mask pass, from_rgb of the image-gradient's gradient, autograd's add) with their
// roundings kept, so every bit of the result is theirs.  Either operand may be NULL (its term is zero).  thread = 8 channels of a pixel.
__global__ __launch_bounds__(256) void fade_rgb_bwd2_kernel(const bf16_t* __restrict__ ggy, const float* __restrict__ ggp, const unsigned char* __restrict__ bits,
                                                            const float* __restrict__ wr, float ws, float alpha, float beta, const float* __restrict__ ab_dev,
                                                            bf16_t* __restrict__ out, size_t npix, int C) {
    const int cv = C / 8;                                    // power of two <= 16: divides the grid stride
    const size_t nvec = npix * cv;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = (int)(i0 % cv) * 8;
    if (ab_dev) { alpha = ab_dev[0]; beta = ab_dev[1]; }
    float w0[8], w1[8], w2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w0[j] = ws * wr[(c0 + j) * 3]; w1[j] = ws * wr[(c0 + j) * 3 + 1]; w2[j] = ws * wr[(c0 + j) * 3 + 2]; }
    for (size_t i = i0; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cv;
        float t1[8], t2[8], o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { t1[j] = 0.f; t2[j] = 0.f; }
        if (ggy) {
            float g[8];
            VecTraits<bf16_t>::load(ggy + i * 8, g);
            const unsigned bb = bits[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) t1[j] = bf2f(f2bf((alpha * g[j]) * (((bb >> j) & 1u) ? 1.f : SGX_LRELU)));
        }
        if (ggp) {
            const float r = ggp[p * 3], g = ggp[p * 3 + 1], b = ggp[p * 3 + 2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                t2[j] = bf2f(f2bf(0.f + (r * w0[j] + g * w1[j] + b * w2[j])));
                if (ab_dev) t2[j] = bf2f(f2bf(beta * t2[j]));
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = t1[j] + t2[j];
        VecTraits<bf16_t>::store(out + i * 8, o);
    }
}
extern "C" int sgx_fade_rgb_bwd2(const void* ggy, const float* ggp, const void* bits, const float* wr, float ws, float alpha, float beta, const float* ab_dev,
                                 void* out, size_t npix, int C, int dtype, void* stream) {
    SGX_REQUIRE(dtype == SGX_BF16 && (C == 32 || C == 64 || C == 128), SGX_EUNSUPPORTED, "fade_rgb_bwd2: bf16, C in {32, 64, 128} (C=%d dtype %d)", C, dtype);
    SGX_REQUIRE((ggy || ggp) && wr && out && npix > 0 && (!ggy || bits), SGX_EINVAL, "fade_rgb_bwd2: null argument");
    SGX_REQUIRE(ab_dev || beta == 1.0f, SGX_EUNSUPPORTED, "fade_rgb_bwd2: a host beta other than 1 rides in ws (as the forward's)");
    SGX_NOTE(6.0 * npix * C, npix * ((ggy ? 4.125 : 2.0) * C + (ggp ? 12.0 : 0.0)), "fade_rgb_bwd2 %zux%d", npix, C);
    const int cv = C / 8;
    unsigned grid = grid_for(npix * cv);
    hipLaunchKernelGGL(fade_rgb_bwd2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, static_cast<const bf16_t*>(ggy), ggp,
                       static_cast<const unsigned char*>(bits), wr, ws, alpha, beta, ab_dev, static_cast<bf16_t*>(out), npix, C);
    SGX_LAUNCH_CHECK("fade_rgb_bwd2_kernel");
    return 0;
}

// ---------------------------------------------------------------- out = alpha*a + beta*b
template <typename T>
__global__ void axpby_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, float alpha, float beta,
                             const float* __restrict__ alpha_dev, const float* __restrict__ beta_dev, size_t nvec, size_t n) {
    constexpr int VE = VecTraits<T>::VE;
    if (alpha_dev) alpha = alpha_dev[0];                       // coefficients kept on the device (graph replay: the
    if (beta_dev) beta = beta_dev[0];                          // fade-in alpha changes every iteration)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        if ((i + 1) * VE <= n) {
            float va[VE], vb[VE];
            VecTraits<T>::load(a + i * VE, va);
            if (b) {
                VecTraits<T>::load(b + i * VE, vb);
#pragma unroll
                for (int j = 0; j < VE; ++j) va[j] = alpha * va[j] + beta * vb[j];
            } else {
#pragma unroll
                for (int j = 0; j < VE; ++j) va[j] = alpha * va[j];
            }
            VecTraits<T>::store(out + i * VE, va);
        } else {
            for (size_t k = i * VE; k < n; ++k) {
                float t = alpha * to_f(a[k]) + (b ? beta * to_f(b[k]) : 0.f);
                out[k] = from_f<T>(t);
            }
        }
    }
}
static int axpby_launch(const void* a, const void* b, void* out, float alpha, float beta, const float* alpha_dev, const float* beta_dev,
                        size_t n, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_NOTE(0.0, (b ? 3.0 : 2.0) * (dtype == SGX_F32 ? 4.0 : 2.0) * n, "axpby %zu", n);
    if (n == 0) return 0;
    if (dtype == SGX_F32) {
        size_t nvec = (n + 3) / 4;
        hipLaunchKernelGGL(axpby_kernel<float>, dim3(grid_all(nvec)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, alpha, beta, alpha_dev, beta_dev, nvec, n);
    } else {
        size_t nvec = (n + 7) / 8;
        hipLaunchKernelGGL(axpby_kernel<bf16_t>, dim3(grid_all(nvec)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, alpha, beta, alpha_dev, beta_dev, nvec, n);
    }
    SGX_LAUNCH_CHECK("axpby");
    return 0;
}
extern "C" int sgx_axpby(const void* a, const void* b, void* out, float alpha, float beta, size_t n, int dtype, void* stream) {
    return axpby_launch(a, b, out, alpha, beta, nullptr, nullptr, n, dtype, stream);
}
extern "C" int sgx_axpby_dev(const void* a, const void* b, void* out, const float* alpha_dev, const float* beta_dev, size_t n, int dtype,
                             void* stream) {
    SGX_REQUIRE(alpha_dev && (beta_dev || !b), SGX_EINVAL, "axpby_dev: missing device coefficient");
    return axpby_launch(a, b, out, 0.f, 0.f, alpha_dev, b ? beta_dev : nullptr, n, dtype, stream);
}

// ---------------------------------------------------------------- depthwise blur [1,2,1]x[1,2,1]/16, zero pad
// Separable, sliding window down a strip of BLUR_ROWS output rows per thread: per input row three 16-byte loads
// (left / centre / right pixel; the neighbours are L1 hits of the adjacent lanes' centres) give the horizontal sum,
// three consecutive horizontal sums give one output row: 3.75 loads per output vector instead of 9.
#define BLUR_ROWS 8
// MODE 0: y = blur(x)            1: y = blur(lrelu(x))            2: y = blur(x) * slope(z)            3: y = blur(x * slope(z))
// 1..3 are the forward, backward and double-backward of "LeakyReLU then blur" (discriminator block: conv0 -> act -> blur)
// with the activation folded into the blur pass: no separate activation-backward pass over the tensor.
// MODE 4 / 5 (bf16): modes 2 / 3 with z given as SIGN BITS (one byte per 8-channel vector, bit j = z[8v + j] > 0; written by the
// convolution that produced z, sgx_conv3x3_signbits): the backward passes read 1/16 of the mask bytes.
template <typename T, int MODE, int ROWS = BLUR_ROWS>
__global__ __launch_bounds__(256) void blur3x3_kernel(const T* __restrict__ x, const T* __restrict__ z, T* __restrict__ y, int B, int H,
                                                      int W, int C) {
    constexpr int VE = VecTraits<T>::VE;
    const int cv = C / VE;
    const int strips = (H + ROWS - 1) / ROWS;
    const size_t nthr = (size_t)B * strips * W * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nthr; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        size_t p = i / cv;
        const int w = (int)(p % W); p /= W;
        const int sidx = (int)(p % strips);
        const int b = (int)(p / strips);
        const int h0 = sidx * ROWS;
        const bool hasl = w > 0, hasr = w + 1 < W;
        const size_t col = (((size_t)b * H * W + w) * cv + c) * VE;          // (b, row 0, w, c)
        const size_t rstride = (size_t)W * cv * VE;
        auto fetch = [&](size_t off, float (&v)[VE]) {                       // one input vector, with the mode's pre-op
            VecTraits<T>::load(x + off, v);
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < VE; ++j) v[j] = lrelu(v[j]);
            } else if (MODE == 3) {
                float m[VE];
                VecTraits<T>::load(z + off, m);
#pragma unroll
                for (int j = 0; j < VE; ++j) v[j] *= lrelu_slope(m[j]);
            } else if (MODE == 5) {
                const unsigned bits = reinterpret_cast<const unsigned char*>(z)[off / VE];
#pragma unroll
                for (int j = 0; j < VE; ++j) v[j] *= ((bits >> j) & 1u) ? 1.f : SGX_LRELU;
            }
        };
        float ha[VE], hb[VE], hc[VE];                                         // horizontal sums of rows r-2, r-1, r
#pragma unroll
        for (int j = 0; j < VE; ++j) { ha[j] = 0.f; hb[j] = 0.f; }
        // MODE 4: the strip's mask bytes are requested up front (a one-byte load issued when its output row is ready would put
        // its latency on every row's critical path: 572 us instead of the 2-tensor streaming time at 1024^2, batch 32)
        unsigned mbits[ROWS];
        if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                mbits[r] = h0 + r < H ? reinterpret_cast<const unsigned char*>(z)[((((size_t)b * H + h0 + r) * W + w) * cv * VE + c * VE) / VE] : 0u;
        }
#pragma unroll
        for (int k = 0; k < ROWS + 2; ++k) {
            const int r = h0 - 1 + k;                                         // input row
            if ((unsigned)r < (unsigned)H) {
                const size_t src = col + (size_t)r * rstride;
                float l[VE], m[VE], rr[VE];
                fetch(src, m);
                if (hasl) fetch(src - cv * VE, l);
                if (hasr) fetch(src + cv * VE, rr);
#pragma unroll
                for (int j = 0; j < VE; ++j) hc[j] = (hasl ? l[j] : 0.f) + 2.f * m[j] + (hasr ? rr[j] : 0.f);
            } else {
#pragma unroll
                for (int j = 0; j < VE; ++j) hc[j] = 0.f;
            }
            if (k >= 2) {
                const int ro = r - 1;                                         // output row
                if (ro < H) {
                    const size_t dst = (((size_t)b * H + ro) * W + w) * cv * VE + c * VE;
                    float o[VE];
#pragma unroll
                    for (int j = 0; j < VE; ++j) o[j] = (ha[j] + 2.f * hb[j] + hc[j]) * (1.f / 16.f);
                    if (MODE == 2) {
                        float m[VE];
                        VecTraits<T>::load(z + dst, m);
#pragma unroll
                        for (int j = 0; j < VE; ++j) o[j] *= lrelu_slope(m[j]);
                    } else if (MODE == 4) {
                        const unsigned bits = mbits[k - 2];
#pragma unroll
                        for (int j = 0; j < VE; ++j) o[j] *= ((bits >> j) & 1u) ? 1.f : SGX_LRELU;
                    }
                    VecTraits<T>::store(y + dst, o);
                }
            }
#pragma unroll
            for (int j = 0; j < VE; ++j) { ha[j] = hb[j]; hb[j] = hc[j]; }
        }
    }
}
// ---- the same pass with ONE global load per input vector (large tensors; C / VE a power of two <= 16 and W * C / VE a multiple of 64,
// so that a wave never straddles an image row).  The kernel above asks L1 for the left and right pixel again: 3 loads per input row of
// which one is HBM traffic, so a wave has a third of its outstanding requests working on the stream, and the pre-op (LeakyReLU / mask) is
// evaluated three times per element -- the pass sits at 3.0-4.6 TB/s where the one-load streaming passes reach 5.4.  Here a lane loads its
// own vector only, PF rows ahead, and takes the neighbours' values from the lanes that loaded them (ds_bpermute: lane -+ C/VE): the raw
// 16 bytes where there is no pre-op, the pre-processed fp32 values where there is one (MODE 1 / 5: evaluated once per element).  The first /
// last C/VE lanes of a wave load their outer neighbour themselves (one exec-masked load; its pre-op rides along on all lanes).  The strip
// and row bookkeeping is wave-uniform (scalar registers, scalar branches), the sums are packed fp32 pairs.  Same values, same order of
// operations as above: bit-identical output.  MODE 3 (full-tensor mask on the input, fp32 networks) stays on the kernel above.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void raw_decode(const uint4& t, f32x2 (&v)[4]) {                  // 8 bf16
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i].x = __uint_as_float(w[i] << 16); v[i].y = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ void raw_decode(const uint4& t, f32x2 (&v)[2]) {                  // 4 fp32
    v[0].x = __uint_as_float(t.x); v[0].y = __uint_as_float(t.y); v[1].x = __uint_as_float(t.z); v[1].y = __uint_as_float(t.w);
}
__device__ __forceinline__ unsigned lane_fetch(int byte_addr, unsigned v) { return (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v); }
__device__ __forceinline__ float lane_fetch(int byte_addr, float v) { return __uint_as_float(lane_fetch(byte_addr, __float_as_uint(v))); }
__device__ __forceinline__ uint4 lane_fetch(int byte_addr, const uint4& v) {
    return make_uint4(lane_fetch(byte_addr, v.x), lane_fetch(byte_addr, v.y), lane_fetch(byte_addr, v.z), lane_fetch(byte_addr, v.w));
}
// v * (bit j of bits ? 1 : SGX_LRELU) without a compare: the bit, sign-extended, selects between the two products' bit patterns
__device__ __forceinline__ float mask_mul(float v, float v_slope, unsigned bits, int j) {
    const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)bits, j, 1);
    return __uint_as_float((__float_as_uint(v) & m) | (__float_as_uint(v_slope) & ~m));
}
template <typename T, int MODE, int ROWS, int PF, bool NT = false>
__global__ __launch_bounds__(256) void blur3x3s_kernel(const T* __restrict__ x, const T* __restrict__ z, T* __restrict__ y, int B, int H,
                                                       int W, int C) {
    static_assert(MODE != 3, "blur3x3s: mode 3 is not built");
    constexpr int VE = VecTraits<T>::VE, VP = VE / 2;
    constexpr bool SHARE_RAW = sizeof(T) == 2 && MODE != 1 && MODE != 5;       // (fp32 tensors: the raw vector IS the four values)
    const int cv = C / VE;
    const unsigned rowv = (unsigned)W * cv;                                   // vectors per image row
    const int strips = (H + ROWS - 1) / ROWS;
    const size_t nthr = (size_t)B * strips * rowv;
    const int lane = threadIdx.x & 63;
    const bool edge_l = lane < cv, edge_r = lane >= 64 - cv;
    const int src_l = ((lane - cv) & 63) << 2, src_r = ((lane + cv) & 63) << 2;
    const unsigned char* zb = reinterpret_cast<const unsigned char*>(z);
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    const f32x2 two = {2.f, 2.f}, sixteenth = {1.f / 16.f, 1.f / 16.f}, slope = {SGX_LRELU, SGX_LRELU};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nthr; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned iv = (unsigned)(i % rowv);                              // w * cv + c: this lane's vector within the row
        const size_t q = i / rowv;
        const int sidx = __builtin_amdgcn_readfirstlane((int)(q % strips)), b = __builtin_amdgcn_readfirstlane((int)(q / strips));
        const int h0 = sidx * ROWS;
        const bool hasl = iv >= (unsigned)cv, hasr = iv + cv < rowv;
        const bool has_edge = (edge_l && hasl) || (edge_r && hasr);           // this lane fetches its outer neighbour itself
        const unsigned ev = edge_l ? iv - cv : iv + cv;
        const size_t img_row0 = (size_t)b * H;
        uint4 rc[PF], re[PF];
        unsigned mc[PF], me[PF];                                              // MODE 5: the mask byte of the own / the outer-neighbour vector
#pragma unroll
        for (int s = 0; s < PF; ++s) { rc[s] = zero4; re[s] = zero4; mc[s] = 0u; me[s] = 0u; }
        auto issue = [&](int k) {
            const int r = h0 - 1 + k, s = k % PF;
            if ((unsigned)r < (unsigned)H) {                                  // (wave-uniform)
                const uint4* xr = reinterpret_cast<const uint4*>(x) + (img_row0 + r) * rowv;
                rc[s] = ld16(xr + iv, NT);
                if (has_edge) re[s] = xr[ev];
                if (MODE == 5) {
                    const unsigned char* zr = zb + (img_row0 + r) * rowv;
                    mc[s] = zr[iv];
                    if (has_edge) me[s] = zr[ev];
                }
            }
            if (MODE == 4 && k >= 2 && r - 1 < H) mc[s] = zb[(img_row0 + r - 1) * rowv + iv];   // the mask byte of the output row this step completes
        };
#pragma unroll
        for (int k = 0; k < PF; ++k) issue(k);
        f32x2 ha[VP], hb[VP], hc[VP];
#pragma unroll
        for (int j = 0; j < VP; ++j) { ha[j] = 0.f; hb[j] = 0.f; }
#pragma unroll
        for (int k = 0; k < ROWS + 2; ++k) {
            const int r = h0 - 1 + k, s = k % PF;
            if ((unsigned)r < (unsigned)H) {                                  // (wave-uniform: every lane of the wave takes part in the exchange)
                f32x2 l[VP], m[VP], rv[VP];
                raw_decode(rc[s], m);
                if (SHARE_RAW) {
                    uint4 l4 = lane_fetch(src_l, rc[s]), r4 = lane_fetch(src_r, rc[s]);
                    if (edge_l) l4 = re[s];                                   // (zero where there is no such pixel: never loaded)
                    if (edge_r) r4 = re[s];
                    raw_decode(l4, l); raw_decode(r4, rv);
                } else {
                    f32x2 e[VP];
                    raw_decode(re[s], e);
                    if (MODE == 1) {
#pragma unroll
                        for (int j = 0; j < VP; ++j) { m[j].x = lrelu(m[j].x); m[j].y = lrelu(m[j].y); e[j].x = lrelu(e[j].x); e[j].y = lrelu(e[j].y); }
                    } else if (MODE == 5) {
#pragma unroll
                        for (int j = 0; j < VP; ++j) {
                            const f32x2 ms = m[j] * slope, es = e[j] * slope;
                            m[j].x = mask_mul(m[j].x, ms.x, mc[s], 2 * j); m[j].y = mask_mul(m[j].y, ms.y, mc[s], 2 * j + 1);
                            e[j].x = mask_mul(e[j].x, es.x, me[s], 2 * j); e[j].y = mask_mul(e[j].y, es.y, me[s], 2 * j + 1);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < VP; ++j) {
                        l[j].x = lane_fetch(src_l, m[j].x); l[j].y = lane_fetch(src_l, m[j].y);
                        rv[j].x = lane_fetch(src_r, m[j].x); rv[j].y = lane_fetch(src_r, m[j].y);
                        if (edge_l) l[j] = e[j];
                        if (edge_r) rv[j] = e[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < VP; ++j) hc[j] = (l[j] + two * m[j]) + rv[j];
            } else {
#pragma unroll
                for (int j = 0; j < VP; ++j) hc[j] = 0.f;
            }
            const unsigned obits = mc[s];
            if (k + PF < ROWS + 2) issue(k + PF);
            if (k >= 2) {
                const int ro = r - 1;
                if (ro < H) {
                    f32x2 o2[VP];
#pragma unroll
                    for (int j = 0; j < VP; ++j) o2[j] = ((ha[j] + two * hb[j]) + hc[j]) * sixteenth;
                    T* yo = y + ((img_row0 + ro) * rowv + iv) * VE;
                    if (MODE == 2) {
                        float mz[VE];
                        VecTraits<T>::load(z + ((img_row0 + ro) * rowv + iv) * VE, mz);
#pragma unroll
                        for (int j = 0; j < VP; ++j) { o2[j].x *= lrelu_slope(mz[2 * j]); o2[j].y *= lrelu_slope(mz[2 * j + 1]); }
                    } else if (MODE == 4) {
                        const unsigned bits = obits;
#pragma unroll
                        for (int j = 0; j < VP; ++j) {
                            const f32x2 os = o2[j] * slope;
                            o2[j].x = mask_mul(o2[j].x, os.x, bits, 2 * j); o2[j].y = mask_mul(o2[j].y, os.y, bits, 2 * j + 1);
                        }
                    }
                    float o[VE];
#pragma unroll
                    for (int j = 0; j < VP; ++j) { o[2 * j] = o2[j].x; o[2 * j + 1] = o2[j].y; }
                    if constexpr (NT && sizeof(T) == 2) {
                        st16(yo, make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4 % VE], o[5 % VE]), pack_bf16x2(o[6 % VE], o[7 % VE])), true);
                    } else VecTraits<T>::store(yo, o);
                }
            }
#pragma unroll
            for (int j = 0; j < VP; ++j) { ha[j] = hb[j]; hb[j] = hc[j]; }
        }
    }
}
// SGX_BLUR_SHFL: 0 = the three-load kernel everywhere, 1..4 = the one-load kernel at that prefetch depth; unset: the depth measured best
// per mode (tools/blur_probe.py, profiles/r04_blur_probe.txt)
static int blur_shfl_pf(int mode) {
    static const int env = [] { const char* e = getenv("SGX_BLUR_SHFL"); return e ? atoi(e) : -1; }();     // read once per process
    if (env >= 0) return env;
    return (mode == 1 || mode == 5) ? 1 : 3;                 // (the modes with a pre-op are the register-heavier ones: more waves beat a deeper ring)
}
template <typename T, int PF, int ROWS = BLUR_ROWS>
static void blur_launch_shfl(const void* x, const void* z, void* y, int B, int H, int W, int C, int mode, hipStream_t st) {
    constexpr int VE = VecTraits<T>::VE;
    const dim3 grid(grid_all((size_t)B * ((H + ROWS - 1) / ROWS) * W * C / VE)), block(256);
    const T* xp = (const T*)x; const T* zp = (const T*)z; T* yp = (T*)y;
    // nontemporal accesses (bf16, the default prefetch depths): a PROBE (SGX_BLUR_NT=1), off by default.  Alone at batch 32
    // (tools/blur_rows_probe.py) they gain: 512^2 x 32 204.6 -> 194.8 us, 256^2 x 64 107.1 -> 98.9 (1024^2 x 16 406 -> 432: the strips' halo
    // rows are the neighbour block's L2 hits there) -- but INSIDE the step they lose: blur1 512^2 209.7 -> 230.1 us, blur0 204.4 -> 213.5,
    // 256^2 103 -> 110..113 (same box, single-stream layer tables): the blur's input was written by the launch before it and its output is
    // read by the launch after it, and a good part of a 268-537 MB tensor is still in the 256 MB memory-side cache unless the hint evicts it
    static const int blur_nt = [] { const char* e = getenv("SGX_BLUR_NT"); return e ? atoi(e) : 0; }();
    if constexpr (sizeof(T) == 2 && (PF == 1 || PF == 3)) {
        if (blur_nt && C >= 32 && sgx_nt_for((double)sizeof(T) * B * H * W * C)) {
            switch (mode) {
                case 0: hipLaunchKernelGGL((blur3x3s_kernel<T, 0, ROWS, PF, true>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
                case 1: hipLaunchKernelGGL((blur3x3s_kernel<T, 1, ROWS, PF, true>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
                case 2: hipLaunchKernelGGL((blur3x3s_kernel<T, 2, ROWS, PF, true>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
                case 4: hipLaunchKernelGGL((blur3x3s_kernel<T, 4, ROWS, PF, true>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
                default: hipLaunchKernelGGL((blur3x3s_kernel<T, 5, ROWS, PF, true>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
            }
            return;
        }
    }
    switch (mode) {
        case 0: hipLaunchKernelGGL((blur3x3s_kernel<T, 0, ROWS, PF>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
        case 1: hipLaunchKernelGGL((blur3x3s_kernel<T, 1, ROWS, PF>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
        case 2: hipLaunchKernelGGL((blur3x3s_kernel<T, 2, ROWS, PF>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
        case 4: hipLaunchKernelGGL((blur3x3s_kernel<T, 4, ROWS, PF>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
        default: hipLaunchKernelGGL((blur3x3s_kernel<T, 5, ROWS, PF>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
    }
}
template <typename T, int ROWS>
static void blur_launch_rows(const void* x, const void* z, void* y, int B, int H, int W, int C, int mode, hipStream_t st) {
    constexpr int VE = VecTraits<T>::VE;
    const dim3 grid(grid_all((size_t)B * ((H + ROWS - 1) / ROWS) * W * C / VE)), block(256);
    const T* xp = (const T*)x; const T* zp = (const T*)z; T* yp = (T*)y;
    switch (mode) {
        case 0: hipLaunchKernelGGL((blur3x3_kernel<T, 0, ROWS>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
        case 1: hipLaunchKernelGGL((blur3x3_kernel<T, 1, ROWS>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
        case 2: hipLaunchKernelGGL((blur3x3_kernel<T, 2, ROWS>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
        case 3: hipLaunchKernelGGL((blur3x3_kernel<T, 3, ROWS>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
        case 4: hipLaunchKernelGGL((blur3x3_kernel<T, 4, ROWS>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
        default: hipLaunchKernelGGL((blur3x3_kernel<T, 5, ROWS>), grid, block, 0, st, xp, zp, yp, B, H, W, C); break;
    }
}
// 8-row strips amortise the vertical halo (3.75 loads per output vector) where the tensor streams from HBM; a small tensor
// (4x4 .. 64x64 at batch 4: a few MB, L2 resident) with 8-row strips is a hundred blocks each walking a 10-row dependent
// chain -- 16-20 us for 2 MB.  There 2-row strips give 4x the lanes and a 4-row chain.
template <typename T>
static void blur_launch(const void* x, const void* z, void* y, int B, int H, int W, int C, int mode, hipStream_t st) {
    const size_t lanes8 = (size_t)B * ((H + BLUR_ROWS - 1) / BLUR_ROWS) * W * C / VecTraits<T>::VE;
    const int cv = C / VecTraits<T>::VE, pf = blur_shfl_pf(mode);
    if (lanes8 < (size_t)256 * 512) blur_launch_rows<T, 2>(x, z, y, B, H, W, C, mode, st);
    else if (pf > 0 && mode != 3 && cv <= 16 && (cv & (cv - 1)) == 0 && ((size_t)W * cv) % 64 == 0 && lanes8 < ((size_t)1 << 31)) {
        // (strip heights 2 / 4 / 16 of this kernel measured in round 6, tools/blur_rows_probe.py: 8 rows stays best -- 5.1-5.2 TB/s of algorithmic bytes
        // is 5.8 with the 10 / 8 rows it reads; the short-lived-block form of the elementwise passes has nothing to give a stencil)
        if (pf == 1) blur_launch_shfl<T, 1>(x, z, y, B, H, W, C, mode, st);
        else if (pf == 2) blur_launch_shfl<T, 2>(x, z, y, B, H, W, C, mode, st);
        else if (pf == 3) blur_launch_shfl<T, 3>(x, z, y, B, H, W, C, mode, st);
        else blur_launch_shfl<T, 4>(x, z, y, B, H, W, C, mode, st);
    } else blur_launch_rows<T, BLUR_ROWS>(x, z, y, B, H, W, C, mode, st);
}
extern "C" int sgx_blur3x3_act(const void* x, const void* z, void* y, int B, int H, int W, int C, int mode, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(mode >= 0 && mode <= 3 && (mode < 2 || z), SGX_EINVAL, "blur3x3_act: mode %d", mode);
    SGX_REQUIRE(dtype == SGX_F32 ? C % 4 == 0 : C % 8 == 0, SGX_EUNSUPPORTED, "blur: C=%d", C);
    SGX_NOTE(0.0, (mode >= 2 ? 3.0 : 2.0) * (dtype == SGX_F32 ? 4.0 : 2.0) * B * H * W * C, "blur%d B%d %dx%d C%d", mode, B, H, W, C);
    if (dtype == SGX_F32) blur_launch<float>(x, z, y, B, H, W, C, mode, st);
    else blur_launch<bf16_t>(x, z, y, B, H, W, C, mode, st);
    SGX_LAUNCH_CHECK("blur3x3");
    return 0;
}
extern "C" int sgx_blur3x3(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    return sgx_blur3x3_act(x, nullptr, y, B, H, W, C, 0, dtype, stream);
}
// modes 2 / 3 of sgx_blur3x3_act with the pre-activation given as sign bits ([pixel][C / 8] bytes, sgx_conv3x3_signbits); bf16
extern "C" int sgx_blur3x3_bits(const void* x, const void* bits, void* y, int B, int H, int W, int C, int mode, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE((mode == 2 || mode == 3) && bits, SGX_EINVAL, "blur3x3_bits: mode %d", mode);
    SGX_REQUIRE(dtype == SGX_BF16 && C % 8 == 0, SGX_EUNSUPPORTED, "blur3x3_bits: bf16 with C %% 8 == 0 (C=%d dtype=%d)", C, dtype);
    SGX_NOTE(0.0, (2.0 * 2.0 + 1.0 / 8.0) * B * H * W * C, "blur%db B%d %dx%d C%d", mode, B, H, W, C);
    blur_launch<bf16_t>(x, bits, y, B, H, W, C, mode + 2, st);
    SGX_LAUNCH_CHECK("blur3x3_bits");
    return 0;
}

// ---------------------------------------------------------------- depthwise K x K correlation, zero padded, any filter
// BlurLayer with a blur_filter other than [1,2,1] (models/CustomLayers.py:251-276: kernel = outer(f, f), optionally
// normalised / flipped; F.conv2d(groups=C, padding=(K-1)//2)):  y[oy][ox] = sum_{i,j} k[i][j] * x[oy+i-pad][ox+j-pad].
// The same kernel serves its adjoint (flipped taps, pad' = K-1-pad, output size = the forward's input size), so the op is
// closed under differentiation.  Not a hot path (the networks of every shipped config use [1,2,1]): one output vector per
// lane, K*K loads.
#define SGX_BLUR_MAXK 7
struct BlurTaps { float k[SGX_BLUR_MAXK * SGX_BLUR_MAXK]; };
template <typename T>
__global__ __launch_bounds__(256) void blur_kxk_kernel(const T* __restrict__ x, T* __restrict__ y, BlurTaps taps, int K, int pad, int B,
                                                       int IH, int IW, int OH, int OW, int C) {
    constexpr int VE = VecTraits<T>::VE;
    const int cv = C / VE;
    const size_t n = (size_t)B * OH * OW * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        size_t p = i / cv;
        const int ox = (int)(p % OW); p /= OW;
        const int oy = (int)(p % OH);
        const int b = (int)(p / OH);
        float acc[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) acc[j] = 0.f;
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy + ky - pad;
            if ((unsigned)iy >= (unsigned)IH) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox + kx - pad;
                if ((unsigned)ix >= (unsigned)IW) continue;
                float v[VE];
                VecTraits<T>::load(x + ((((size_t)b * IH + iy) * IW + ix) * cv + c) * VE, v);
                const float w = taps.k[ky * K + kx];
#pragma unroll
                for (int j = 0; j < VE; ++j) acc[j] += w * v[j];
            }
        }
        VecTraits<T>::store(y + i * VE, acc);
    }
}
extern "C" int sgx_blur_kxk(const void* x, void* y, const float* taps_host, int K, int pad, int B, int IH, int IW, int OH, int OW, int C,
                            int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(taps_host && K >= 1 && K <= SGX_BLUR_MAXK && pad >= 0 && pad < K, SGX_EINVAL, "blur_kxk: K=%d pad=%d (K <= %d)", K, pad, SGX_BLUR_MAXK);
    SGX_REQUIRE(OH >= 1 && OW >= 1 && OH <= IH + 2 * pad - K + 1 && OW <= IW + 2 * pad - K + 1, SGX_EINVAL,
                "blur_kxk: output %dx%d does not fit input %dx%d with K=%d pad=%d", OH, OW, IH, IW, K, pad);
    SGX_REQUIRE(dtype == SGX_F32 ? C % 4 == 0 : (dtype == SGX_BF16 && C % 8 == 0), SGX_EUNSUPPORTED, "blur_kxk: C=%d dtype=%d", C, dtype);
    BlurTaps t;
    for (int i = 0; i < K * K; ++i) t.k[i] = taps_host[i];
    SGX_NOTE(0.0, (dtype == SGX_F32 ? 4.0 : 2.0) * B * C * ((double)IH * IW + (double)OH * OW), "blur_k%d B%d %dx%d C%d", K, B, IH, IW, C);
    if (dtype == SGX_F32)
        hipLaunchKernelGGL(blur_kxk_kernel<float>, dim3(grid_for((size_t)B * OH * OW * C / 4)), dim3(256), 0, st, (const float*)x, (float*)y, t, K, pad, B, IH, IW, OH, OW, C);
    else
        hipLaunchKernelGGL(blur_kxk_kernel<bf16_t>, dim3(grid_for((size_t)B * OH * OW * C / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, t, K, pad, B, IH, IW, OH, OW, C);
    SGX_LAUNCH_CHECK("blur_kxk");
    return 0;
}

// ---------------------------------------------------------------- 2x2 pooling / nearest upsample.  C generic (RGB has C=3):
// scalar-per-lane variant used when C is not a multiple of the vector width.
template <typename T, bool VECT>
__global__ void pool2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, float scale) {
    constexpr int VE = VECT ? VecTraits<T>::VE : 1;
    const int cv = C / VE, OH = H / 2, OW = W / 2;
    const size_t n = (size_t)B * OH * OW * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        size_t p = i / cv;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int b = (int)(p / OH);
        const size_t base = ((((size_t)b * H + 2 * oh) * W + 2 * ow) * cv + c) * VE;
        if (VECT) {
            float a0[VecTraits<T>::VE], a1[VecTraits<T>::VE], a2[VecTraits<T>::VE], a3[VecTraits<T>::VE];
            VecTraits<T>::load(x + base, a0);
            VecTraits<T>::load(x + base + C, a1);
            VecTraits<T>::load(x + base + (size_t)W * C, a2);
            VecTraits<T>::load(x + base + (size_t)W * C + C, a3);
#pragma unroll
            for (int j = 0; j < VecTraits<T>::VE; ++j) a0[j] = scale * ((a0[j] + a1[j]) + (a2[j] + a3[j]));
            VecTraits<T>::store(y + i * VE, a0);
        } else {
            float s = (to_f(x[base]) + to_f(x[base + C])) + (to_f(x[base + (size_t)W * C]) + to_f(x[base + (size_t)W * C + C]));
            y[i] = from_f<T>(scale * s);
        }
    }
}
// fp32 RGB images (C = 3, W % 8 == 0): the scalar variant above moves 4 bytes per lane (4.4 TB/s on the five 1024^2 image pools of a batch-32
// step).  Here a lane owns FOUR output pixels: 2 x 6 float4 loads (eight input pixels of two rows), 3 float4 stores; the sums in the scalar
// variant's order (bit-identical).  One trip per thread (grid_all).
__global__ __launch_bounds__(256) void pool2_rgb_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, float scale) {
    const int OH = H / 2, OW = W / 2, q4 = OW / 4;
    const size_t n = (size_t)B * OH * q4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % q4);
        const size_t row = i / q4;                                 // b * OH + oh
        const int oh = (int)(row % OH);
        const size_t b = row / OH;
        const float4* r0 = reinterpret_cast<const float4*>(x + ((b * H + 2 * oh) * W + 8 * q) * 3);
        const float4* r1 = reinterpret_cast<const float4*>(x + ((b * H + 2 * oh + 1) * W + 8 * q) * 3);
        float a[24], c[24], o[12];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float4 t = r0[k], u = r1[k];
            a[4 * k] = t.x; a[4 * k + 1] = t.y; a[4 * k + 2] = t.z; a[4 * k + 3] = t.w;
            c[4 * k] = u.x; c[4 * k + 1] = u.y; c[4 * k + 2] = u.z; c[4 * k + 3] = u.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) o[3 * j + ch] = scale * ((a[6 * j + ch] + a[6 * j + 3 + ch]) + (c[6 * j + ch] + c[6 * j + 3 + ch]));
        float4* dst = reinterpret_cast<float4*>(y + (row * OW + 4 * q) * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) dst[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    }
}
extern "C" int sgx_pool2(const void* x, void* y, int B, int H, int W, int C, float scale, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_NOTE(0.0, 1.25 * (dtype == SGX_F32 ? 4.0 : 2.0) * B * H * W * C, "pool2 B%d %dx%d C%d", B, H, W, C);
    SGX_REQUIRE(H % 2 == 0 && W % 2 == 0, SGX_EINVAL, "pool2: odd size");
    const size_t nout = (size_t)B * (H / 2) * (W / 2) * C;
    if (dtype == SGX_F32) {
        if (C % 4 == 0) hipLaunchKernelGGL((pool2_kernel<float, true>), dim3(grid_for(nout / 4)), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C, scale);
        else if (C == 3 && W % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
            hipLaunchKernelGGL(pool2_rgb_kernel, dim3(grid_all(nout / 12)), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, scale);
        else hipLaunchKernelGGL((pool2_kernel<float, false>), dim3(grid_for(nout)), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C, scale);
    } else {
        if (C % 8 == 0) hipLaunchKernelGGL((pool2_kernel<bf16_t, true>), dim3(grid_for(nout / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, scale);
        else hipLaunchKernelGGL((pool2_kernel<bf16_t, false>), dim3(grid_for(nout)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, scale);
    }
    SGX_LAUNCH_CHECK("pool2");
    return 0;
}

// ---------------------------------------------------------------- real images at the current depth with the fade-in blend
// out = alpha * x + beta * nearest_up2(avgpool2(x)) on fp32 RGB images (models/GAN.py:575-586: ds_real_samples and its
// down-then-up-sampled "prior" blended by alpha) in one pass: a lane owns one 2x2 pixel block (two runs of 6 floats).
__global__ void downsample_fade_rgb_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, float alpha, float beta,
                                           const float* __restrict__ ab_dev) {
    if (ab_dev) { alpha = ab_dev[0]; beta = ab_dev[1]; }
    const int hw = W >> 1, hh = H >> 1;
    const size_t n = (size_t)B * hh * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int bx = (int)(i % hw);
        const size_t r = i / hw;                                  // b * hh + by
        const int by = (int)(r % hh);
        const size_t b = r / hh;
        const size_t o0 = ((b * H + 2 * by) * W + 2 * bx) * 3, o1 = o0 + (size_t)W * 3;
        float v0[6], v1[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float2 a2 = *reinterpret_cast<const float2*>(x + o0 + 2 * k), b2 = *reinterpret_cast<const float2*>(x + o1 + 2 * k);
            v0[2 * k] = a2.x; v0[2 * k + 1] = a2.y; v1[2 * k] = b2.x; v1[2 * k + 1] = b2.y;
        }
        float m[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) m[c] = beta * (0.25f * ((v0[c] + v0[3 + c]) + (v1[c] + v1[3 + c])));   // (the 2x2 mean as sgx_pool2 sums it)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            *reinterpret_cast<float2*>(out + o0 + 2 * k) = make_float2(alpha * v0[2 * k] + m[(2 * k) % 3], alpha * v0[2 * k + 1] + m[(2 * k + 1) % 3]);
            *reinterpret_cast<float2*>(out + o1 + 2 * k) = make_float2(alpha * v1[2 * k] + m[(2 * k) % 3], alpha * v1[2 * k + 1] + m[(2 * k + 1) % 3]);
        }
    }
}
extern "C" int sgx_downsample_fade_rgb(const float* x, float* out, int B, int H, int W, float alpha, float beta, const float* ab_dev, void* stream) {
    SGX_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, SGX_EINVAL, "downsample_fade_rgb: bad arguments (%dx%d)", H, W);
    SGX_NOTE(0.0, 2.0 * 4.0 * 3.0 * B * H * W, "downsample_fade B%d %dx%d", B, H, W);
    hipLaunchKernelGGL(downsample_fade_rgb_kernel, dim3(grid_for((size_t)B * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, x, out, B, H, W, alpha, beta, ab_dev);
    SGX_LAUNCH_CHECK("downsample_fade_rgb");
    return 0;
}

// ---------------------------------------------------------------- uint8 images -> normalised NHWC activations
// ToTensor + Normalize(0.5, 0.5) (+ RandomHorizontalFlip with host-drawn decisions) of the reference's input pipeline,
// on the device: the batch crosses PCIe as bytes (12.6 MB at B=4, 1024^2 instead of 50 MB of fp32).  HBM-bound: 3 B in,
// 12 B (fp32) or 6 B (bf16) out per pixel.  A lane converts 4 consecutive output pixels: 12 source bytes (three dword
// loads from an HWC row, or one dword from each plane of a CHW image) -> 48 / 24 contiguous output bytes.
// Arithmetic in the reference's order, fp32, IEEE division: (v / 255 - 0.5) / 0.5.
template <typename T, bool CHW>
__global__ void images_u8_kernel(const unsigned char* __restrict__ src, T* __restrict__ dst, const int* __restrict__ flip,
                                 int B, int H, int W) {
    const int wq = W / 4;
    const size_t n = (size_t)B * H * wq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int xq = (int)(i % wq);
        size_t p = i / wq;
        const int y = (int)(p % H), b = (int)(p / H);
        const bool fl = flip != nullptr && flip[b] != 0;
        const int sx = fl ? W - 4 - 4 * xq : 4 * xq;                 // first of the 4 source pixels (mirrored block if flipped)
        unsigned char v[4][3];                                       // [source pixel][channel]
        if (CHW) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned w = *reinterpret_cast<const unsigned*>(src + (((size_t)b * 3 + c) * H + y) * W + sx);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k][c] = (unsigned char)(w >> (8 * k));
            }
        } else {
            const unsigned* s3 = reinterpret_cast<const unsigned*>(src + (((size_t)b * H + y) * W + sx) * 3);
            const unsigned w0 = s3[0], w1 = s3[1], w2 = s3[2];
            const unsigned char bytes[12] = {(unsigned char)w0, (unsigned char)(w0 >> 8), (unsigned char)(w0 >> 16), (unsigned char)(w0 >> 24),
                                             (unsigned char)w1, (unsigned char)(w1 >> 8), (unsigned char)(w1 >> 16), (unsigned char)(w1 >> 24),
                                             (unsigned char)w2, (unsigned char)(w2 >> 8), (unsigned char)(w2 >> 16), (unsigned char)(w2 >> 24)};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) v[k][c] = bytes[3 * k + c];
        }
        float o[12];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) o[3 * k + c] = (__fdiv_rn((float)v[fl ? 3 - k : k][c], 255.0f) - 0.5f) * 2.0f;   // /0.5 == *2 exactly
        T* d = dst + (((size_t)b * H + y) * W + 4 * xq) * 3;
        if (sizeof(T) == 4) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                reinterpret_cast<float4*>(d)[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
        } else {
            uint2 lo, hi;
            lo.x = pack_bf16x2(o[0], o[1]); lo.y = pack_bf16x2(o[2], o[3]);
            hi.x = pack_bf16x2(o[4], o[5]); hi.y = pack_bf16x2(o[6], o[7]);
            reinterpret_cast<uint2*>(d)[0] = lo; reinterpret_cast<uint2*>(d)[1] = hi;
            uint2 t;
            t.x = pack_bf16x2(o[8], o[9]); t.y = pack_bf16x2(o[10], o[11]);
            reinterpret_cast<uint2*>(d)[2] = t;
        }
    }
}
extern "C" int sgx_images_u8_to_nhwc(const void* src, void* dst, const int* flip, int B, int H, int W, int src_chw, int dtype,
                                     void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(B > 0 && H > 0 && W > 0 && W % 4 == 0, SGX_EINVAL, "images_u8_to_nhwc: width must be a multiple of 4 (got %dx%d)", H, W);
    SGX_REQUIRE(dtype == SGX_F32 || dtype == SGX_BF16, SGX_EINVAL, "images_u8_to_nhwc: bad dtype");
    SGX_NOTE(0.0, (3.0 + 3.0 * (dtype == SGX_F32 ? 4.0 : 2.0)) * B * H * W, "images_u8 B%d %dx%d", B, H, W);
    const size_t n = (size_t)B * H * (W / 4);
    const auto* s = static_cast<const unsigned char*>(src);
    if (dtype == SGX_F32) {
        if (src_chw) hipLaunchKernelGGL((images_u8_kernel<float, true>), dim3(grid_for(n)), dim3(256), 0, st, s, (float*)dst, flip, B, H, W);
        else hipLaunchKernelGGL((images_u8_kernel<float, false>), dim3(grid_for(n)), dim3(256), 0, st, s, (float*)dst, flip, B, H, W);
    } else {
        if (src_chw) hipLaunchKernelGGL((images_u8_kernel<bf16_t, true>), dim3(grid_for(n)), dim3(256), 0, st, s, (bf16_t*)dst, flip, B, H, W);
        else hipLaunchKernelGGL((images_u8_kernel<bf16_t, false>), dim3(grid_for(n)), dim3(256), 0, st, s, (bf16_t*)dst, flip, B, H, W);
    }
    SGX_LAUNCH_CHECK("images_u8_kernel");
    return 0;
}

template <typename T, bool VECT>
__global__ void up2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, float scale) {
    constexpr int VE = VECT ? VecTraits<T>::VE : 1;
    const int cv = C / VE, OH = 2 * H, OW = 2 * W;
    const size_t n = (size_t)B * OH * OW * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        size_t p = i / cv;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int b = (int)(p / OH);
        const size_t src = ((((size_t)b * H + (oh >> 1)) * W + (ow >> 1)) * cv + c) * VE;
        if (VECT) {
            float v[VecTraits<T>::VE];
            VecTraits<T>::load(x + src, v);
#pragma unroll
            for (int j = 0; j < VecTraits<T>::VE; ++j) v[j] *= scale;
            VecTraits<T>::store(y + i * VE, v);
        } else {
            y[i] = from_f<T>(scale * to_f(x[src]));
        }
    }
}
// fp32 RGB images (C = 3): four consecutive floats of an output row per lane (16-byte stores; a row is 3*OW floats)
__global__ void up2_rgb_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, float scale) {
    const int OW = 2 * W, q4 = OW * 3 / 4;
    const size_t n = (size_t)B * 2 * H * q4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % q4);
        const size_t row = i / q4;                             // b * OH + oh
        const int oh = (int)(row % (2 * H));
        const size_t b = row / (2 * H);
        const float* src = x + ((b * H + (oh >> 1)) * W) * 3;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = q * 4 + j;
            v[j] = scale * src[((e / 3) >> 1) * 3 + e % 3];
        }
        *reinterpret_cast<float4*>(y + row * OW * 3 + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}
// the same with whole 16-byte accesses (W % 4 == 0): a lane owns four input pixels = 3 float4 loads and writes their eight output pixels to
// BOTH output rows = 12 float4 stores (the kernel above gathers four scalars per stored vector: 3.7 TB/s)
__global__ __launch_bounds__(256) void up2_rgb4_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, float scale) {
    const int q4 = W / 4, OW = 2 * W;
    const size_t n = (size_t)B * H * q4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % q4);
        const size_t row = i / q4;                                 // b * H + h
        const float4* src = reinterpret_cast<const float4*>(x + (row * W + 4 * q) * 3);
        float a[12], o[24];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float4 t = src[k]; a[4 * k] = t.x; a[4 * k + 1] = t.y; a[4 * k + 2] = t.z; a[4 * k + 3] = t.w; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { const float v = scale * a[3 * j + ch]; o[6 * j + ch] = v; o[6 * j + 3 + ch] = v; }
        float4* d0 = reinterpret_cast<float4*>(y + ((2 * row) * OW + 8 * q) * 3);
        float4* d1 = reinterpret_cast<float4*>(y + ((2 * row + 1) * OW + 8 * q) * 3);
#pragma unroll
        for (int k = 0; k < 6; ++k) { const float4 t = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]); d0[k] = t; d1[k] = t; }
    }
}
// y = a + scale * nearest-up(x) for fp32 RGB images (C = 3): the JOIN of the two gradients of an image that feeds both the full-resolution
// branch and, through a 2x2 average pool, the residual branch of the discriminator (models/GAN.py:423-427) -- the pool's adjoint and the sum
// of the two contributions in one pass (27 bytes per pixel instead of 15 + 36 for up2 + add).  Four consecutive floats of an output row per lane.
__global__ void up2_add_rgb_kernel(const float* __restrict__ x, const float* __restrict__ a, float* __restrict__ y, int B, int H, int W, float scale) {
    const int OW = 2 * W, q4 = OW * 3 / 4;
    const size_t n = (size_t)B * 2 * H * q4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % q4);
        const size_t row = i / q4;                             // b * OH + oh
        const int oh = (int)(row % (2 * H));
        const size_t b = row / (2 * H);
        const float* src = x + ((b * H + (oh >> 1)) * W) * 3;
        const float4 av = *reinterpret_cast<const float4*>(a + row * OW * 3 + q * 4);
        const float ae[4] = {av.x, av.y, av.z, av.w};
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = q * 4 + j;
            v[j] = ae[j] + scale * src[((e / 3) >> 1) * 3 + e % 3];
        }
        *reinterpret_cast<float4*>(y + row * OW * 3 + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}
template <typename T>
__global__ void up2_add_kernel(const T* __restrict__ x, const T* __restrict__ a, T* __restrict__ y, int B, int H, int W, int C, float scale) {
    const int OH = 2 * H, OW = 2 * W;
    const size_t n = (size_t)B * OH * OW * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t p = i / C;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int b = (int)(p / OH);
        y[i] = from_f<T>(to_f(a[i]) + scale * to_f(x[(((size_t)b * H + (oh >> 1)) * W + (ow >> 1)) * C + c]));
    }
}
extern "C" int sgx_up2_add(const void* x, const void* a, void* y, int B, int H, int W, int C, float scale, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(x && a && y && B > 0 && H > 0 && W > 0 && C > 0, SGX_EINVAL, "up2_add: bad arguments");
    SGX_REQUIRE(dtype == SGX_F32 || dtype == SGX_BF16, SGX_EINVAL, "up2_add: bad dtype");
    const double es = dtype == SGX_F32 ? 4.0 : 2.0;
    SGX_NOTE(0.0, 9.0 * es * B * H * W * C, "up2+add B%d %dx%d C%d", B, H, W, C);
    const size_t nout = (size_t)B * H * W * 4 * C;
    if (dtype == SGX_F32 && C == 3 && (2 * W * 3) % 4 == 0)
        hipLaunchKernelGGL(up2_add_rgb_kernel, dim3(grid_all(nout / 4)), dim3(256), 0, st, (const float*)x, (const float*)a, (float*)y, B, H, W, scale);
    else if (dtype == SGX_F32)
        hipLaunchKernelGGL(up2_add_kernel<float>, dim3(grid_for(nout)), dim3(256), 0, st, (const float*)x, (const float*)a, (float*)y, B, H, W, C, scale);
    else
        hipLaunchKernelGGL(up2_add_kernel<bf16_t>, dim3(grid_for(nout)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)a, (bf16_t*)y, B, H, W, C, scale);
    SGX_LAUNCH_CHECK("up2_add");
    return 0;
}
extern "C" int sgx_up2(const void* x, void* y, int B, int H, int W, int C, float scale, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_NOTE(0.0, 5.0 * (dtype == SGX_F32 ? 4.0 : 2.0) * B * H * W * C, "up2 B%d %dx%d C%d", B, H, W, C);
    const size_t nout = (size_t)B * H * W * 4 * C;
    if (dtype == SGX_F32) {
        if (C % 4 == 0) hipLaunchKernelGGL((up2_kernel<float, true>), dim3(grid_all(nout / 4)), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C, scale);
        else if (C == 3 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
            hipLaunchKernelGGL(up2_rgb4_kernel, dim3(grid_all((size_t)B * H * (W / 4))), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, scale);
        else if (C == 3 && (2 * W * 3) % 4 == 0) hipLaunchKernelGGL(up2_rgb_kernel, dim3(grid_all(nout / 4)), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, scale);
        else hipLaunchKernelGGL((up2_kernel<float, false>), dim3(grid_for(nout)), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C, scale);
    } else {
        if (C % 8 == 0) hipLaunchKernelGGL((up2_kernel<bf16_t, true>), dim3(grid_all(nout / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, scale);
        else hipLaunchKernelGGL((up2_kernel<bf16_t, false>), dim3(grid_for(nout)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, scale);
    }
    SGX_LAUNCH_CHECK("up2");
    return 0;
}

// ---------------------------------------------------------------- out[c] = sum_p x[p][c]
// stage 1: each block sums a pixel range into ws[block][C] (fp32 per-thread, double across the block);
// stage 2: one block sums the partials in double.  Deterministic.
#define COLSUM_BLOCKS 1024
template <typename T>
__global__ void colsum_stage1(const T* __restrict__ x, double* __restrict__ ws, size_t npix, int C) {
    // thread t handles channel (t % C') of pixel rows t / C' ... generic scalar layout: C <= 1024
    extern __shared__ double sh[];
    const int tpr = C < 256 ? C : 256;                 // threads per pixel row
    const int rows = 256 / tpr;                        // pixel rows per iteration
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const size_t per = (npix + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    for (int cb = 0; cb < C; cb += tpr) {                  // uniform trip count (barriers inside)
        const int c = cb + tc;
        double acc = 0.0;
        if (tr < rows && c < C) {
            float part = 0.f; int cnt = 0;
            for (size_t p = p0 + tr; p < p1; p += rows) {
                part += to_f(x[p * C + c]);
                if (++cnt == 64) { acc += (double)part; part = 0.f; cnt = 0; }
            }
            acc += (double)part;
        }
        sh[threadIdx.x] = acc;
        __syncthreads();
        if (tr == 0 && c < C) {
            double s = 0.0;
            for (int r = 0; r < rows; ++r) s += sh[r * tpr + tc];
            ws[(size_t)blockIdx.x * C + c] = s;
        }
        __syncthreads();
    }
}
// 32 outputs x 8 partial lanes per block: the chain over partial blocks is nblk/8 long, not nblk
__global__ __launch_bounds__(256) void colsum_stage2(const double* __restrict__ ws, float* __restrict__ out, int nblk, int C, float scale) {
    __shared__ double sh[64][5];                           // 4 outputs per block x 64 partial lanes
    const int cl = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cl;
    double s = 0.0;
    if (c < C)
        for (int b = pl; b < nblk; b += 64) s += ws[(size_t)b * C + c];
    sh[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && c < C) {
        double t = 0.0;
        for (int k = 0; k < 64; ++k) t += sh[k][cl];
        out[c] = (float)(t * (double)scale);
    }
}
// Vector variant (C % VE == 0, C/VE <= 256, power of two): one 16-byte load per lane per row; per-lane fp32 partials over
// <= 64 rows, then fp64.  NJ = 1: plain column sums.  NJ = 3: column sums weighted by the three RGB values of the pixel
// (the 1x1 RGB weight gradient), output ws[blk][j][C].
template <typename T, int NJ>
__global__ __launch_bounds__(256) void colsum_vec_stage1(const T* __restrict__ x, const float* __restrict__ img, double* __restrict__ ws,
                                                         size_t npix, int C) {
    constexpr int VE = VecTraits<T>::VE;
    extern __shared__ float shf[];                                   // [256][NJ*VE]
    const int cv = C / VE, rows = 256 / cv;
    const int tc = threadIdx.x % cv, tr = threadIdx.x / cv;
    const size_t per = (npix + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    // per-lane fp32 partials: a lane sums npix / (blocks * rows) values (tens to a few hundred); everything across lanes,
    // blocks and the final total is fp64
    float part[NJ * VE];
#pragma unroll
    for (int k = 0; k < NJ * VE; ++k) part[k] = 0.f;
    auto accum = [&](const float (&v)[VE], size_t p) {
        if (NJ == 1) {
#pragma unroll
            for (int k = 0; k < VE; ++k) part[k] += v[k];
        } else {
            const float r = img[p * 3], g = img[p * 3 + 1], b = img[p * 3 + 2];
#pragma unroll
            for (int k = 0; k < VE; ++k) { part[k] += v[k] * r; part[VE + k] += v[k] * g; part[2 * VE + k] += v[k] * b; }
        }
    };
    size_t p = p0 + tr;
    for (; p + 3 * (size_t)rows < p1; p += 4 * (size_t)rows) {          // four rows in flight per lane
        float v0[VE], v1[VE], v2[VE], v3[VE];
        VecTraits<T>::load(x + (p * cv + tc) * VE, v0);
        VecTraits<T>::load(x + ((p + rows) * cv + tc) * VE, v1);
        VecTraits<T>::load(x + ((p + 2 * (size_t)rows) * cv + tc) * VE, v2);
        VecTraits<T>::load(x + ((p + 3 * (size_t)rows) * cv + tc) * VE, v3);
        accum(v0, p); accum(v1, p + rows); accum(v2, p + 2 * (size_t)rows); accum(v3, p + 3 * (size_t)rows);
    }
    for (; p < p1; p += rows) {
        float v[VE];
        VecTraits<T>::load(x + (p * cv + tc) * VE, v);
        accum(v, p);
    }
#pragma unroll
    for (int k = 0; k < NJ * VE; ++k) shf[threadIdx.x * NJ * VE + k] = part[k];
    __syncthreads();
    for (int o = threadIdx.x; o < cv * NJ * VE; o += 256) {             // output (channel vector, k): sum over the rows
        const int c = o / (NJ * VE), k = o % (NJ * VE);
        double s = 0.0;
        for (int r = 0; r < rows; ++r) s += (double)shf[(r * cv + c) * NJ * VE + k];
        ws[((size_t)blockIdx.x * NJ + k / VE) * C + c * VE + (k % VE)] = s;
    }
}
static bool colsum_vec_ok(int C, int ve) { const int cv = C / ve; return C % ve == 0 && cv >= 1 && cv <= 256 && (cv & (cv - 1)) == 0; }

extern "C" size_t sgx_colsum_ws_bytes(size_t npix, int C) { (void)npix; return (size_t)COLSUM_BLOCKS * C * sizeof(double); }
extern "C" int sgx_colsum(const void* x, float* out, float scale, void* ws, size_t ws_bytes, size_t npix, int C, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(ws_bytes >= sgx_colsum_ws_bytes(npix, C), SGX_EWORKSPACE, "colsum: workspace");
    SGX_REQUIRE(C >= 1, SGX_EINVAL, "colsum: C=%d", C);     // scalar kernel: any C (C=3 for the RGB bias gradient)
    SGX_NOTE(0.0, (dtype == SGX_F32 ? 4.0 : 2.0) * npix * C, "colsum %zux%d", npix, C);
    int nblk = (int)((npix + 63) / 64);
    if (nblk > COLSUM_BLOCKS) nblk = COLSUM_BLOCKS;
    if (nblk < 1) nblk = 1;
    if (dtype == SGX_F32 && colsum_vec_ok(C, 4))
        hipLaunchKernelGGL((colsum_vec_stage1<float, 1>), dim3(nblk), dim3(256), 256 * 4 * sizeof(float), st, (const float*)x, (const float*)nullptr, (double*)ws, npix, C);
    else if (dtype == SGX_BF16 && colsum_vec_ok(C, 8))
        hipLaunchKernelGGL((colsum_vec_stage1<bf16_t, 1>), dim3(nblk), dim3(256), 256 * 8 * sizeof(float), st, (const bf16_t*)x, (const float*)nullptr, (double*)ws, npix, C);
    else if (dtype == SGX_F32) hipLaunchKernelGGL(colsum_stage1<float>, dim3(nblk), dim3(256), 256 * sizeof(double), st, (const float*)x, (double*)ws, npix, C);
    else hipLaunchKernelGGL(colsum_stage1<bf16_t>, dim3(nblk), dim3(256), 256 * sizeof(double), st, (const bf16_t*)x, (double*)ws, npix, C);
    SGX_LAUNCH_CHECK("colsum_stage1");
    hipLaunchKernelGGL(colsum_stage2, dim3((C + 3) / 4), dim3(256), 0, st, (const double*)ws, out, nblk, C, scale);
    SGX_LAUNCH_CHECK("colsum_stage2");
    return 0;
}

// ---------------------------------------------------------------- 1x1 RGB convolutions (3 <-> C), images fp32 [p][3]
template <typename T>
__global__ void rgb_in_kernel(const float* __restrict__ img, const float* __restrict__ w, int sj, int sc, float wscale,
                              const float* __restrict__ bias, T* __restrict__ y, size_t npix, int C, const T* __restrict__ add) {
    constexpr int VE = VecTraits<T>::VE;
    const int cv = C / VE;                                   // power of two <= 64, divides the grid stride
    const size_t nvec = npix * cv;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = (int)(i0 % cv) * VE;                      // the thread's channel group never changes: hoist its weights
    float wr[VE], wg[VE], wb[VE], bb[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) {
        wr[j] = wscale * w[(c0 + j) * sc]; wg[j] = wscale * w[sj + (c0 + j) * sc]; wb[j] = wscale * w[2 * sj + (c0 + j) * sc];
        bb[j] = bias ? bias[c0 + j] : 0.f;
    }
    for (size_t i = i0; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cv;
        const float r = img[p * 3], g = img[p * 3 + 1], b = img[p * 3 + 2];
        float v[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) v[j] = bb[j] + (r * wr[j] + g * wg[j] + b * wb[j]);
        if (add) {                                           // (sgx_rgb_in_add: the other gradient of a forked tensor joins here)
            float av[VE];
            VecTraits<T>::load(add + i * VE, av);
#pragma unroll
            for (int j = 0; j < VE; ++j) v[j] += av[j];
        }
        VecTraits<T>::store(y + i * VE, v);
    }
}
// The same as SHORT-LIVED blocks (round 6, tools/stream_probe.hip: a write stream runs at 4.6 TB/s through a capped grid-stride loop and at 6.8 as
// blocks that store one 16-byte vector per thread and exit; this pass is a write stream -- 12 bytes in, 2 C bytes out per pixel).  The hoisted
// weights that tied the kernel above to its loop (uncapped it lost 3x: 24 scalar loads per stored vector) go through LDS once per block: the
// same products `wscale * w`, the same sum order: bit-identical output.
template <typename T>
__global__ __launch_bounds__(256) void rgb_in1_kernel(const float* __restrict__ img, const float* __restrict__ w, int sj, int sc, float wscale,
                                                      const float* __restrict__ bias, T* __restrict__ y, unsigned nvec, int C, const T* __restrict__ add) {
    constexpr int VE = VecTraits<T>::VE, U = 4;              // four vectors per thread: with one, the block's lifetime is the latency of its table
                                                             // loads (1024^2 x 16 at batch 32: 294 -> 343 us; with four: see the launch function)
    extern __shared__ float tw[];                            // [4][C]: wscale * W[0..2][c], bias[c]
    for (int c = threadIdx.x; c < C; c += 256) {
        tw[c] = wscale * w[c * sc]; tw[C + c] = wscale * w[sj + c * sc]; tw[2 * C + c] = wscale * w[2 * sj + c * sc];
        tw[3 * C + c] = bias ? bias[c] : 0.f;
    }
    const unsigned i0 = blockIdx.x * (256u * U) + threadIdx.x, cv = (unsigned)(C / VE);
    float r[U], g[U], b[U];
    uint4 araw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned i = i0 + u * 256u;
        r[u] = g[u] = b[u] = 0.f; araw[u] = make_uint4(0u, 0u, 0u, 0u);
        if (i < nvec) {
            const size_t p = i / cv;
            r[u] = img[p * 3]; g[u] = img[p * 3 + 1]; b[u] = img[p * 3 + 2];
            if (add) araw[u] = *reinterpret_cast<const uint4*>(add + (size_t)i * VE);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned i = i0 + u * 256u;
        if (i >= nvec) continue;
        const int c0 = (int)(i % cv) * VE;
        float v[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) v[j] = tw[3 * C + c0 + j] + (r[u] * tw[c0 + j] + g[u] * tw[C + c0 + j] + b[u] * tw[2 * C + c0 + j]);
        if (add) {
            const unsigned w4[4] = {araw[u].x, araw[u].y, araw[u].z, araw[u].w};
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[(2 * k) % VE] += __uint_as_float(w4[k] << 16); v[(2 * k + 1) % VE] += __uint_as_float(w4[k] & 0xffff0000u); }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k % VE] += __uint_as_float(w4[k]);
            }
        }
        VecTraits<T>::store(y + (size_t)i * VE, v);
    }
}
template <typename T>
static bool rgb_in1_launch(const float* img, const float* w, int sj, int sc, float wscale, const float* bias, T* y, size_t npix, int C, const T* add, hipStream_t st) {
    static const int on = [] { const char* e = getenv("SGX_RGB_IN1"); return e ? atoi(e) : 1; }();           // A/B switch
    const size_t nvec = npix * (C / VecTraits<T>::VE);
    // measured inside the single-stream step (same box, SGX_RGB_IN1=0 vs 1): batch 32 rgb_in 1024^2 x 16 288 -> 275 us, rgb_in+add 512^2 x 32 221 -> 188;
    // batch 4 52.5 -> 39.4 and 52.0 -> 30.0 us.  (Staging the block's pixels through LDS with 16-byte loads instead of three scalar loads per
    // vector: no better -- 285 / 189 and 43 / 33 us.)
    if (!on || nvec < 65536 || nvec >= 0x7fffffffull || C > 2048) return false;                               // small launches keep the loop kernel
    hipLaunchKernelGGL(rgb_in1_kernel<T>, dim3((unsigned)((nvec + 1023) / 1024)), dim3(256), (size_t)4 * C * sizeof(float), st, img, w, sj, sc, wscale, bias, y,
                       (unsigned)nvec, C, add);
    return true;
}
// the kernel hoists its channel group's weights, so the grid stride (256 * blocks) must be a multiple of cv
static inline unsigned rgb_in_grid(size_t nvec, int cv) {
    unsigned g = grid_for(nvec);
    if (256 % cv != 0) g = (g + cv - 1) / cv * cv;
    return g;
}
extern "C" int sgx_rgb_in(const float* img, const float* w, int sj, int sc, float wscale, const float* bias, void* y, size_t npix, int C, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_NOTE(6.0 * npix * C, npix * (12.0 + (dtype == SGX_F32 ? 4.0 : 2.0) * C), "rgb_in %zux%d", npix, C);
    if (dtype == SGX_F32) {
        SGX_REQUIRE(C % 4 == 0, SGX_EUNSUPPORTED, "rgb_in: C %% 4");
        if (!rgb_in1_launch<float>(img, w, sj, sc, wscale, bias, (float*)y, npix, C, (const float*)nullptr, st))
        hipLaunchKernelGGL(rgb_in_kernel<float>, dim3(rgb_in_grid(npix * C / 4, C / 4)), dim3(256), 0, st, img, w, sj, sc, wscale, bias, (float*)y, npix, C, (const float*)nullptr);
    } else {
        SGX_REQUIRE(C % 8 == 0, SGX_EUNSUPPORTED, "rgb_in: C %% 8");
        if (!rgb_in1_launch<bf16_t>(img, w, sj, sc, wscale, bias, (bf16_t*)y, npix, C, (const bf16_t*)nullptr, st))
        hipLaunchKernelGGL(rgb_in_kernel<bf16_t>, dim3(rgb_in_grid(npix * C / 8, C / 8)), dim3(256), 0, st, img, w, sj, sc, wscale, bias, (bf16_t*)y, npix, C, (const bf16_t*)nullptr);
    }
    SGX_LAUNCH_CHECK("rgb_in");
    return 0;
}

// y = add + sgx_rgb_in(img) (no bias): to_rgb's data gradient written ON TOP of the gradient the forked activation received from its other
// consumer (generator under fade-in: a block's output feeds the next block and the previous resolution's to_rgb, models/GAN.py:199-202) --
// one rounding and one pass instead of sgx_rgb_in + an add pass (read 2, write 1)
extern "C" int sgx_rgb_in_add(const float* img, const float* w, int sj, int sc, float wscale, const void* add, void* y, size_t npix, int C, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(img && w && add && y && npix > 0, SGX_EINVAL, "rgb_in_add: null argument");
    SGX_NOTE(6.0 * npix * C, npix * (12.0 + 2.0 * (dtype == SGX_F32 ? 4.0 : 2.0) * C), "rgb_in+add %zux%d", npix, C);
    if (dtype == SGX_F32) {
        SGX_REQUIRE(C % 4 == 0, SGX_EUNSUPPORTED, "rgb_in_add: C %% 4");
        if (!rgb_in1_launch<float>(img, w, sj, sc, wscale, (const float*)nullptr, (float*)y, npix, C, (const float*)add, st))
        hipLaunchKernelGGL(rgb_in_kernel<float>, dim3(rgb_in_grid(npix * C / 4, C / 4)), dim3(256), 0, st, img, w, sj, sc, wscale, (const float*)nullptr, (float*)y, npix, C, (const float*)add);
    } else {
        SGX_REQUIRE(dtype == SGX_BF16 && C % 8 == 0, SGX_EUNSUPPORTED, "rgb_in_add: bf16 with C %% 8");
        if (!rgb_in1_launch<bf16_t>(img, w, sj, sc, wscale, (const float*)nullptr, (bf16_t*)y, npix, C, (const bf16_t*)add, st))
        hipLaunchKernelGGL(rgb_in_kernel<bf16_t>, dim3(rgb_in_grid(npix * C / 8, C / 8)), dim3(256), 0, st, img, w, sj, sc, wscale, (const float*)nullptr, (bf16_t*)y, npix, C, (const bf16_t*)add);
    }
    SGX_LAUNCH_CHECK("rgb_in_add");
    return 0;
}

// one pixel per group of LPP lanes (LPP = C/VE capped at 16); partial dot products reduced with shuffles.
// The 3 x C weights are staged once per block in LDS, pre-multiplied by wscale.
// ``low`` (optional): the fade-in lerp of the generator's output folded in (models/GAN.py:199-202) -- img = alpha * to_rgb(x) +
// beta * nearest_up2(low), low = the previous resolution's RGB image [B][H/2][W/2][3]; alpha / beta from the launch or, for a
// captured step graph, from device memory (ab_dev[0], ab_dev[1]).
template <typename T>
__global__ void rgb_out_kernel(const T* __restrict__ x, const float* __restrict__ w, int sj, int sc, float wscale,
                               const float* __restrict__ bias, float* __restrict__ img, size_t npix, int C, int lpp,
                               const float* __restrict__ low, int Himg, int Wimg, float alpha, float beta, const float* __restrict__ ab_dev) {
    constexpr int VE = VecTraits<T>::VE;
    extern __shared__ float sw[];                            // [3][C]
    if (ab_dev) { alpha = ab_dev[0]; beta = ab_dev[1]; }
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) sw[i] = (alpha * wscale) * w[(i / C) * sj + (i % C) * sc];
    __syncthreads();
    const float b0 = bias ? alpha * bias[0] : 0.f, b1 = bias ? alpha * bias[1] : 0.f, b2 = bias ? alpha * bias[2] : 0.f;
    const int cv = C / VE;
    const int sub = threadIdx.x % lpp;
    const size_t gid = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / lpp;
    const size_t gstride = ((size_t)gridDim.x * blockDim.x) / lpp;
    const size_t niter = (npix + gstride - 1) / gstride;
    for (size_t it = 0; it < niter; ++it) {
        const size_t p = gid + it * gstride;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (p < npix) {
            for (int v = sub; v < cv; v += lpp) {
                float t[VE];
                VecTraits<T>::load(x + (p * cv + v) * VE, t);
#pragma unroll
                for (int q = 0; q < VE / 4; ++q) {
                    const int c = v * VE + q * 4;
                    const float4 w0 = *reinterpret_cast<const float4*>(sw + c);
                    const float4 w1 = *reinterpret_cast<const float4*>(sw + C + c);
                    const float4 w2 = *reinterpret_cast<const float4*>(sw + 2 * C + c);
                    s0 += t[q * 4] * w0.x + t[q * 4 + 1] * w0.y + t[q * 4 + 2] * w0.z + t[q * 4 + 3] * w0.w;
                    s1 += t[q * 4] * w1.x + t[q * 4 + 1] * w1.y + t[q * 4 + 2] * w1.z + t[q * 4 + 3] * w1.w;
                    s2 += t[q * 4] * w2.x + t[q * 4 + 1] * w2.y + t[q * 4 + 2] * w2.z + t[q * 4 + 3] * w2.w;
                }
            }
        }
        for (int o = lpp >> 1; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64);
        }
        if (p < npix && sub == 0) {
            float o0 = s0 + b0, o1 = s1 + b1, o2 = s2 + b2;
            if (low) {
                const int xw = (int)(p % Wimg);
                const size_t row = p / Wimg;                       // b * Himg + y
                const int y = (int)(row % Himg);
                const size_t bimg = row / Himg;
                const float* l = low + ((bimg * (Himg >> 1) + (y >> 1)) * (Wimg >> 1) + (xw >> 1)) * 3;
                o0 += beta * l[0]; o1 += beta * l[1]; o2 += beta * l[2];
            }
            img[p * 3] = o0;
            img[p * 3 + 1] = o1;
            img[p * 3 + 2] = o2;
        }
    }
}
// The same with FOUR pixels per lane group and no loop (round 6, as rgb_in1): the loads of a lane's four vectors are requested before the block
// waits for its weight table, every block is 1024 / lpp pixels and exits.  Same products, same sum order per pixel: bit-identical image.
template <typename T>
__global__ __launch_bounds__(256) void rgb_out1_kernel(const T* __restrict__ x, const float* __restrict__ w, int sj, int sc, float wscale,
                                                       const float* __restrict__ bias, float* __restrict__ img, unsigned npix, int C, int lpp,
                                                       const float* __restrict__ low, int Himg, int Wimg, float alpha, float beta, const float* __restrict__ ab_dev) {
    constexpr int VE = VecTraits<T>::VE, U = 4;
    extern __shared__ float sw[];                            // [3][C]
    if (ab_dev) { alpha = ab_dev[0]; beta = ab_dev[1]; }
    const int sub = threadIdx.x % lpp, grp = threadIdx.x / lpp, ppb = 256 / lpp;     // lane in its pixel group, group, groups per block
    const unsigned p0 = blockIdx.x * (unsigned)(ppb * U) + grp;
    uint4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned p = p0 + u * ppb;
        raw[u] = make_uint4(0u, 0u, 0u, 0u);
        if (p < npix) raw[u] = *reinterpret_cast<const uint4*>(x + ((size_t)p * lpp + sub) * VE);           // (cv == lpp: one vector per lane, rgb_out1_launch)
    }
    for (int i = threadIdx.x; i < 3 * C; i += 256) sw[i] = (alpha * wscale) * w[(i / C) * sj + (i % C) * sc];
    __syncthreads();
    const float b0 = bias ? alpha * bias[0] : 0.f, b1 = bias ? alpha * bias[1] : 0.f, b2 = bias ? alpha * bias[2] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned p = p0 + u * ppb;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (p < npix) {
            float t[VE];
            const unsigned w4[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { t[(2 * k) % VE] = __uint_as_float(w4[k] << 16); t[(2 * k + 1) % VE] = __uint_as_float(w4[k] & 0xffff0000u); }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k % VE] = __uint_as_float(w4[k]);
            }
#pragma unroll
            for (int q = 0; q < VE / 4; ++q) {
                const int c = sub * VE + q * 4;
                const float4 w0 = *reinterpret_cast<const float4*>(sw + c);
                const float4 w1 = *reinterpret_cast<const float4*>(sw + C + c);
                const float4 w2 = *reinterpret_cast<const float4*>(sw + 2 * C + c);
                s0 += t[q * 4] * w0.x + t[q * 4 + 1] * w0.y + t[q * 4 + 2] * w0.z + t[q * 4 + 3] * w0.w;
                s1 += t[q * 4] * w1.x + t[q * 4 + 1] * w1.y + t[q * 4 + 2] * w1.z + t[q * 4 + 3] * w1.w;
                s2 += t[q * 4] * w2.x + t[q * 4 + 1] * w2.y + t[q * 4 + 2] * w2.z + t[q * 4 + 3] * w2.w;
            }
        }
        for (int o = lpp >> 1; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64);
        }
        if (p < npix && sub == 0) {
            float o0 = s0 + b0, o1 = s1 + b1, o2 = s2 + b2;
            if (low) {
                const int xw = (int)(p % (unsigned)Wimg);
                const unsigned row = p / (unsigned)Wimg;           // b * Himg + y
                const int y = (int)(row % (unsigned)Himg);
                const size_t bimg = row / (unsigned)Himg;
                const float* l = low + ((bimg * (Himg >> 1) + (y >> 1)) * (Wimg >> 1) + (xw >> 1)) * 3;
                o0 += beta * l[0]; o1 += beta * l[1]; o2 += beta * l[2];
            }
            img[(size_t)p * 3] = o0;
            img[(size_t)p * 3 + 1] = o1;
            img[(size_t)p * 3 + 2] = o2;
        }
    }
}
static int rgb_out_launch(const void* x, const float* w, int sj, int sc, float wscale, const float* bias, float* img, size_t npix, int C, int dtype,
                          const float* low, int H, int W, float alpha, float beta, const float* ab_dev, hipStream_t st) {
    const int ve = dtype == SGX_F32 ? 4 : 8;
    SGX_REQUIRE(C % ve == 0 && C <= 4096, SGX_EUNSUPPORTED, "rgb_out: C=%d", C);
    int lpp = C / ve;
    if (lpp > 16) lpp = 16;
    SGX_REQUIRE((lpp & (lpp - 1)) == 0, SGX_EUNSUPPORTED, "rgb_out: C=%d", C);
    const size_t sh = (size_t)3 * C * sizeof(float);
    {   // short-lived blocks where every lane of a pixel group holds exactly one vector (C / ve == lpp <= 16) and the launch is large enough
        static const int on = [] { const char* e = getenv("SGX_RGB_OUT1"); return e ? atoi(e) : 1; }();          // A/B switch
        if (on && C / ve == lpp && npix >= 65536 && npix < 0x7fffffffull) {
            const unsigned ppb4 = (unsigned)(256 / lpp) * 4, g1 = (unsigned)((npix + ppb4 - 1) / ppb4);
            if (dtype == SGX_F32) hipLaunchKernelGGL(rgb_out1_kernel<float>, dim3(g1), dim3(256), sh, st, (const float*)x, w, sj, sc, wscale, bias, img, (unsigned)npix, C, lpp, low, H, W, alpha, beta, ab_dev);
            else hipLaunchKernelGGL(rgb_out1_kernel<bf16_t>, dim3(g1), dim3(256), sh, st, (const bf16_t*)x, w, sj, sc, wscale, bias, img, (unsigned)npix, C, lpp, low, H, W, alpha, beta, ab_dev);
            SGX_LAUNCH_CHECK("rgb_out1");
            return 0;
        }
    }
    const unsigned g = grid_for(npix * lpp);
    if (dtype == SGX_F32) hipLaunchKernelGGL(rgb_out_kernel<float>, dim3(g), dim3(256), sh, st, (const float*)x, w, sj, sc, wscale, bias, img, npix, C, lpp, low, H, W, alpha, beta, ab_dev);
    else hipLaunchKernelGGL(rgb_out_kernel<bf16_t>, dim3(g), dim3(256), sh, st, (const bf16_t*)x, w, sj, sc, wscale, bias, img, npix, C, lpp, low, H, W, alpha, beta, ab_dev);
    SGX_LAUNCH_CHECK("rgb_out");
    return 0;
}
extern "C" int sgx_rgb_out(const void* x, const float* w, int sj, int sc, float wscale, const float* bias, float* img, size_t npix, int C, int dtype, void* stream) {
    SGX_NOTE(6.0 * npix * C, npix * (12.0 + (dtype == SGX_F32 ? 4.0 : 2.0) * C), "rgb_out %zux%d", npix, C);
    return rgb_out_launch(x, w, sj, sc, wscale, bias, img, npix, C, dtype, nullptr, 1, 1, 1.f, 0.f, nullptr, (hipStream_t)stream);
}
extern "C" int sgx_rgb_out_fade(const void* x, const float* w, int sj, int sc, float wscale, const float* bias, const float* low, float alpha,
                                float beta, const float* ab_dev, float* img, int B, int H, int W, int C, int dtype, void* stream) {
    SGX_REQUIRE(x && w && low && img && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, SGX_EINVAL, "rgb_out_fade: bad arguments (%dx%d)", H, W);
    const size_t npix = (size_t)B * H * W;
    SGX_NOTE(6.0 * npix * C, npix * (12.0 + 3.0 + (dtype == SGX_F32 ? 4.0 : 2.0) * C), "rgb_out+fade %zux%d", npix, C);
    return rgb_out_launch(x, w, sj, sc, wscale, bias, img, npix, C, dtype, low, H, W, alpha, beta, ab_dev, (hipStream_t)stream);
}

// ---------------------------------------------------------------- the generator's LAST layer epilogue inside to_rgb (round 4)
// The last LayerEpilogue of the synthesis network feeds only to_rgb (models/GAN.py:199-202 after models/Blocks.py:87-88): with
//   t = lrelu(y + ebias + nw * noise),  x2 = (t - mean) * rstd * (s0 + 1) + s1 = A t + S   per (image, channel)
// the image is  alpha * (wscale W x2 + rbias) + beta * up(low) = sum_c (alpha wscale W[j][c] A[c]) t[c] + const[j]: one pass over
// the convolution's output y instead of the epilogue's apply pass (read y, write x2) plus to_rgb's read of x2.  Blocks are image
// aligned (grid.y = image), the folded weights live in LDS.
template <typename T>
__global__ __launch_bounds__(256) void rgb_out_epi_kernel(const T* __restrict__ y, const float* __restrict__ ebias, const float* __restrict__ noise,
                                                          const float* __restrict__ nw, const float* __restrict__ style, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ w, int sj, int sc, float wscale,
                                                          const float* __restrict__ rbias, float* __restrict__ img, int HW, int C, int lpp,
                                                          const float* __restrict__ low, int Himg, int Wimg, float alpha, float beta,
                                                          const float* __restrict__ ab_dev) {
    constexpr int VE = VecTraits<T>::VE;
    extern __shared__ float sw[];                            // [3][C] folded weights, [C] bias, [C] noise weight, [3] constants, [256 * 3] scratch
    if (ab_dev) { alpha = ab_dev[0]; beta = ab_dev[1]; }     // (coefficients in device memory: a captured step graph)
    float* kb = sw + 3 * C; float* kw = kb + C; float* sb = kw + C; float* red = sb + 4;
    const int b = blockIdx.y;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;                      // this thread's share of sum_c W[j][c] S[c]
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float s0 = style ? style[(size_t)b * 2 * C + c] : 0.f, s1 = style ? style[(size_t)b * 2 * C + C + c] : 0.f;
        const float A = rstd[(size_t)b * C + c] * (s0 + 1.f), S = s1 - mean[(size_t)b * C + c] * A;
        const float w0 = (alpha * wscale) * w[c * sc], w1 = (alpha * wscale) * w[sj + c * sc], w2 = (alpha * wscale) * w[2 * sj + c * sc];
        sw[c] = w0 * A; sw[C + c] = w1 * A; sw[2 * C + c] = w2 * A;
        c0 += w0 * S; c1 += w1 * S; c2 += w2 * S;
        kb[c] = ebias ? ebias[c] : 0.f; kw[c] = nw[c];
    }
    red[threadIdx.x * 3] = c0; red[threadIdx.x * 3 + 1] = c1; red[threadIdx.x * 3 + 2] = c2;
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
        for (int k = 0; k < 256; ++k) t += red[k * 3 + threadIdx.x];
        sb[threadIdx.x] = t + (rbias ? alpha * rbias[threadIdx.x] : 0.f);
    }
    __syncthreads();
    const float b0 = sb[0], b1 = sb[1], b2 = sb[2];
    const int cv = C / VE;
    const int sub = threadIdx.x % lpp;
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / lpp, gstride = (gridDim.x * blockDim.x) / lpp;
    const T* yb = y + (size_t)b * HW * C;
    const float* nzb = noise + (size_t)b * HW;
    const int niter = (HW + gstride - 1) / gstride;          // (uniform trip count: the shuffles below need every lane)
    for (int it = 0; it < niter; ++it) {
        const int p = gid + it * gstride;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (p < HW) {
            const float nz = nzb[p];
            for (int v = sub; v < cv; v += lpp) {
                float t[VE];
                VecTraits<T>::load(yb + ((size_t)p * cv + v) * VE, t);
#pragma unroll
                for (int q = 0; q < VE; ++q) {
                    const int c = v * VE + q;
                    const float a = lrelu(t[q] + kb[c] + kw[c] * nz);
                    s0 += a * sw[c]; s1 += a * sw[C + c]; s2 += a * sw[2 * C + c];
                }
            }
        }
        for (int o = lpp >> 1; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64);
        }
        if (p < HW && sub == 0) {
            float o0 = s0 + b0, o1 = s1 + b1, o2 = s2 + b2;
            if (low) {
                const int xw = p % Wimg, yy = p / Wimg;
                const float* l = low + (((size_t)b * (Himg >> 1) + (yy >> 1)) * (Wimg >> 1) + (xw >> 1)) * 3;
                o0 += beta * l[0]; o1 += beta * l[1]; o2 += beta * l[2];
            }
            float* o = img + ((size_t)b * HW + p) * 3;
            o[0] = o0; o[1] = o1; o[2] = o2;
        }
    }
}
extern "C" int sgx_rgb_out_epi(const void* y, const float* ebias, const float* noise, const float* nw, const float* style, const float* mean,
                               const float* rstd, const float* w, int sj, int sc, float wscale, const float* rbias, const float* low, float alpha,
                               float beta, const float* ab_dev, float* img, int B, int H, int W, int C, int dtype, void* stream) {
    SGX_REQUIRE(y && noise && nw && mean && rstd && w && img && B > 0 && H > 0 && W > 0, SGX_EINVAL, "rgb_out_epi: bad arguments");
    SGX_REQUIRE(!low || (H % 2 == 0 && W % 2 == 0), SGX_EINVAL, "rgb_out_epi: odd size %dx%d with a low-resolution image", H, W);
    const int ve = dtype == SGX_F32 ? 4 : 8;
    SGX_REQUIRE((dtype == SGX_F32 || dtype == SGX_BF16) && C % ve == 0 && C <= 2048, SGX_EUNSUPPORTED, "rgb_out_epi: C=%d", C);
    int lpp = C / ve;
    if (lpp > 16) lpp = 16;
    SGX_REQUIRE((lpp & (lpp - 1)) == 0, SGX_EUNSUPPORTED, "rgb_out_epi: C=%d", C);
    const int HW = H * W;
    const double npix = (double)B * HW;
    SGX_NOTE(8.0 * npix * C, npix * (12.0 + 4.0 + (low ? 3.0 : 0.0) + (dtype == SGX_F32 ? 4.0 : 2.0) * C), "epi+rgb_out%s %.0fx%d", low ? "+fade" : "", npix, C);
    long bx = ((long)HW * lpp + 255) / 256;
    const long cap = 4096 / B > 1 ? 4096 / B : 1;
    if (bx > cap) bx = cap;
    const size_t sh = (size_t)(5 * C + 4 + 256 * 3) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SGX_F32)
        hipLaunchKernelGGL(rgb_out_epi_kernel<float>, dim3((unsigned)bx, (unsigned)B), dim3(256), sh, st, (const float*)y, ebias, noise, nw, style, mean, rstd, w, sj, sc,
                           wscale, rbias, img, HW, C, lpp, low, H, W, alpha, beta, ab_dev);
    else
        hipLaunchKernelGGL(rgb_out_epi_kernel<bf16_t>, dim3((unsigned)bx, (unsigned)B), dim3(256), sh, st, (const bf16_t*)y, ebias, noise, nw, style, mean, rstd, w, sj,
                           sc, wscale, rbias, img, HW, C, lpp, low, H, W, alpha, beta, ab_dev);
    SGX_LAUNCH_CHECK("rgb_out_epi");
    return 0;
}

// to_rgb's weight and bias gradient with the epilogue recomputed from y:  dW[j][c] = scale sum_b (A[b][c] G1[b][j][c] + S[b][c] G0[b][j]),
// G1[b][j][c] = sum_{p in b} g[p][j] t[p][c],  G0[b][j] = sum_{p in b} g[p][j];  db[j] = bscale sum_b G0[b][j].
// Stage 1: image-aligned blocks, per-block partials ws[b][blk][3 * C + 3] (fp32 per lane over <= a few hundred pixels, then fp64).
template <typename T>
__global__ __launch_bounds__(256) void rgb_wgrad_epi_stage1(const T* __restrict__ y, const float* __restrict__ g, const float* __restrict__ ebias,
                                                            const float* __restrict__ noise, const float* __restrict__ nw, double* __restrict__ ws, int HW, int C) {
    constexpr int VE = VecTraits<T>::VE;
    extern __shared__ float shf[];                                   // [256][3 * VE + 3]
    constexpr int NP = 3 * VE + 3;
    const int b = blockIdx.y, nblk = gridDim.x;
    const int cv = C / VE, rows = 256 / cv;
    const int tc = threadIdx.x % cv, tr = threadIdx.x / cv;
    const int per = (HW + nblk - 1) / nblk;
    const int p0 = blockIdx.x * per, p1 = p0 + per < HW ? p0 + per : HW;
    float kb[VE], kw[VE];
#pragma unroll
    for (int k = 0; k < VE; ++k) { kb[k] = ebias ? ebias[tc * VE + k] : 0.f; kw[k] = nw[tc * VE + k]; }
    float part[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) part[k] = 0.f;
    const T* yb = y + (size_t)b * HW * C;
    const float* gb = g + (size_t)b * HW * 3;
    const float* nzb = noise + (size_t)b * HW;
    if (tr < rows) {
        for (int p = p0 + tr; p < p1; p += rows) {
            float v[VE];
            VecTraits<T>::load(yb + ((size_t)p * cv + tc) * VE, v);
            const float nz = nzb[p], r = gb[p * 3], gg = gb[p * 3 + 1], bb = gb[p * 3 + 2];
#pragma unroll
            for (int k = 0; k < VE; ++k) {
                const float a = lrelu(v[k] + kb[k] + kw[k] * nz);
                part[k] += a * r; part[VE + k] += a * gg; part[2 * VE + k] += a * bb;
            }
            part[3 * VE] += r; part[3 * VE + 1] += gg; part[3 * VE + 2] += bb;
        }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) shf[threadIdx.x * NP + k] = part[k];
    __syncthreads();
    double* o = ws + ((size_t)b * nblk + blockIdx.x) * (3 * C + 3);
    for (int e = threadIdx.x; e < cv * 3 * VE + 3; e += 256) {
        double s = 0.0;
        if (e < cv * 3 * VE) {
            const int c = e / (3 * VE), k = e % (3 * VE);
            for (int r = 0; r < rows; ++r) s += (double)shf[(r * cv + c) * NP + k];
            o[(k / VE) * C + c * VE + (k % VE)] = s;
        } else {
            const int j = e - cv * 3 * VE;
            for (int r = 0; r < rows; ++r) s += (double)shf[(r * cv) * NP + 3 * VE + j];      // (every channel lane of a row summed the same pixels: take lane 0)
            o[3 * C + j] = s;
        }
    }
}
__global__ __launch_bounds__(256) void rgb_wgrad_epi_stage2(const double* __restrict__ ws, const float* __restrict__ style, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dw, float* __restrict__ db, int B, int nblk, int C,
                                                            int sj, int sc, float scale, float bscale) {
    // one block per output (e = j * C + c, then the three bias sums): its 256 threads stride over the B * nblk partials, fixed-order
    // tree in fp64 (deterministic)
    __shared__ double red[256];
    const int e = blockIdx.x, stride = 3 * C + 3, n = B * nblk;
    double t = 0.0;
    if (e < 3 * C) {
        const int j = e / C, c = e % C;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int b = i / nblk;
            const double* o = ws + (size_t)i * stride;
            const float s0 = style ? style[(size_t)b * 2 * C + c] : 0.f, s1 = style ? style[(size_t)b * 2 * C + C + c] : 0.f;
            const float A = rstd[(size_t)b * C + c] * (s0 + 1.f), S = s1 - mean[(size_t)b * C + c] * A;
            t += (double)A * o[j * C + c] + (double)S * o[3 * C + j];
        }
    } else {
        const int j = e - 3 * C;
        for (int i = threadIdx.x; i < n; i += 256) t += ws[(size_t)i * stride + 3 * C + j];
    }
    red[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (e < 3 * C) dw[(e / C) * sj + (e % C) * sc] = (float)(red[0] * scale);
        else if (db) db[e - 3 * C] = (float)(red[0] * bscale);
    }
}
static int rgb_wgrad_epi_blocks(int B, int HW) {
    int n = (HW + 255) / 256;
    const int cap = 1024 / B > 1 ? 1024 / B : 1;
    if (n > cap) n = cap;
    return n < 1 ? 1 : n;
}
extern "C" size_t sgx_rgb_wgrad_epi_ws_bytes(int B, int HW, int C) { return (size_t)B * rgb_wgrad_epi_blocks(B, HW) * (3 * C + 3) * sizeof(double); }
extern "C" int sgx_rgb_wgrad_epi(const void* y, const float* g, const float* ebias, const float* noise, const float* nw, const float* style, const float* mean,
                                 const float* rstd, float* dw, float* db, int sj, int sc, float scale, float bscale, void* ws, size_t ws_bytes, int B, int HW,
                                 int C, int dtype, void* stream) {
    SGX_REQUIRE(y && g && noise && nw && mean && rstd && dw && ws, SGX_EINVAL, "rgb_wgrad_epi: null argument");
    SGX_REQUIRE(ws_bytes >= sgx_rgb_wgrad_epi_ws_bytes(B, HW, C), SGX_EWORKSPACE, "rgb_wgrad_epi: workspace");
    const int ve = dtype == SGX_F32 ? 4 : 8;
    SGX_REQUIRE((dtype == SGX_F32 || dtype == SGX_BF16) && colsum_vec_ok(C, ve), SGX_EUNSUPPORTED, "rgb_wgrad_epi: C=%d", C);
    const int nblk = rgb_wgrad_epi_blocks(B, HW);
    SGX_NOTE(8.0 * (double)B * HW * C, (double)B * HW * (12.0 + 4.0 + (dtype == SGX_F32 ? 4.0 : 2.0) * C), "epi+rgb_wgrad %dx%d", B * HW, C);
    hipStream_t st = (hipStream_t)stream;
    const size_t sh = (size_t)256 * (3 * ve + 3) * sizeof(float);
    if (dtype == SGX_F32) hipLaunchKernelGGL(rgb_wgrad_epi_stage1<float>, dim3(nblk, B), dim3(256), sh, st, (const float*)y, g, ebias, noise, nw, (double*)ws, HW, C);
    else hipLaunchKernelGGL(rgb_wgrad_epi_stage1<bf16_t>, dim3(nblk, B), dim3(256), sh, st, (const bf16_t*)y, g, ebias, noise, nw, (double*)ws, HW, C);
    SGX_LAUNCH_CHECK("rgb_wgrad_epi_stage1");
    hipLaunchKernelGGL(rgb_wgrad_epi_stage2, dim3(3 * C + 3), dim3(256), 0, st, (const double*)ws, style, mean, rstd, dw, db, B, nblk, C, sj, sc, scale, bscale);
    SGX_LAUNCH_CHECK("rgb_wgrad_epi_stage2");
    return 0;
}

// dw[j][c] = sum_p img[p][j] * f[p][c]: stage 1 partials per block in double, stage 2 sums them.
template <typename T>
__global__ void rgb_wgrad_stage1(const float* __restrict__ img, const T* __restrict__ f, double* __restrict__ ws, size_t npix, int C) {
    extern __shared__ double sh[];                        // [256][3]
    const int tpr = C < 256 ? C : 256, rows = 256 / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const size_t per = (npix + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
    for (int cb = 0; cb < C; cb += tpr) {
        const int c = cb + tc;
        double a0 = 0, a1 = 0, a2 = 0;
        if (tr < rows && c < C) {
            float q0 = 0.f, q1 = 0.f, q2 = 0.f; int cnt = 0;
            for (size_t p = p0 + tr; p < p1; p += rows) {
                const float v = to_f(f[p * C + c]);
                q0 += v * img[p * 3]; q1 += v * img[p * 3 + 1]; q2 += v * img[p * 3 + 2];
                if (++cnt == 64) { a0 += q0; a1 += q1; a2 += q2; q0 = q1 = q2 = 0.f; cnt = 0; }
            }
            a0 += q0; a1 += q1; a2 += q2;
        }
        sh[threadIdx.x * 3] = a0; sh[threadIdx.x * 3 + 1] = a1; sh[threadIdx.x * 3 + 2] = a2;
        __syncthreads();
        if (tr == 0 && c < C) {
            double s0 = 0, s1 = 0, s2 = 0;
            for (int r = 0; r < rows; ++r) { s0 += sh[(r * tpr + tc) * 3]; s1 += sh[(r * tpr + tc) * 3 + 1]; s2 += sh[(r * tpr + tc) * 3 + 2]; }
            double* o = ws + (size_t)blockIdx.x * 3 * C;
            o[c] = s0; o[C + c] = s1; o[2 * C + c] = s2;
        }
        __syncthreads();
    }
}
extern "C" size_t sgx_rgb_wgrad_ws_bytes(size_t npix, int C) { (void)npix; return (size_t)COLSUM_BLOCKS * 3 * C * sizeof(double); }
// sums the per-block partials [blk][3][C] and scatters into the parameter layout dw[j*sj + c*sc]
__global__ __launch_bounds__(256) void rgb_wgrad_stage2(const double* __restrict__ ws, float* __restrict__ dw, int nblk, int C, int sj, int sc,
                                                        float wscale) {
    __shared__ double sh[64][5];
    const int el = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const int e = blockIdx.x * 4 + el;                       // e = j*C + c
    double s = 0.0;
    if (e < 3 * C)
        for (int b = pl; b < nblk; b += 64) s += ws[(size_t)b * 3 * C + e];
    sh[pl][el] = s;
    __syncthreads();
    if (pl == 0 && e < 3 * C) {
        double t = 0.0;
        for (int k = 0; k < 64; ++k) t += sh[k][el];
        dw[(e / C) * sj + (e % C) * sc] = (float)(t * (double)wscale);
    }
}
extern "C" int sgx_rgb_wgrad(const float* img, const void* f, float* dw, int sj, int sc, float wscale, void* ws, size_t ws_bytes, size_t npix, int C, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(ws_bytes >= sgx_rgb_wgrad_ws_bytes(npix, C), SGX_EWORKSPACE, "rgb_wgrad: workspace");
    SGX_REQUIRE(C <= 256 ? (256 % C == 0) : true, SGX_EUNSUPPORTED, "rgb_wgrad: C=%d", C);
    SGX_NOTE(6.0 * npix * C, npix * (12.0 + (dtype == SGX_F32 ? 4.0 : 2.0) * C), "rgb_wgrad %zux%d", npix, C);
    int nblk = (int)((npix + 63) / 64);
    if (nblk > COLSUM_BLOCKS) nblk = COLSUM_BLOCKS;
    if (nblk < 1) nblk = 1;
    if (dtype == SGX_F32 && colsum_vec_ok(C, 4))
        hipLaunchKernelGGL((colsum_vec_stage1<float, 3>), dim3(nblk), dim3(256), 256 * 12 * sizeof(float), st, (const float*)f, img, (double*)ws, npix, C);
    else if (dtype == SGX_BF16 && colsum_vec_ok(C, 8))
        hipLaunchKernelGGL((colsum_vec_stage1<bf16_t, 3>), dim3(nblk), dim3(256), 256 * 24 * sizeof(float), st, (const bf16_t*)f, img, (double*)ws, npix, C);
    else if (dtype == SGX_F32) hipLaunchKernelGGL(rgb_wgrad_stage1<float>, dim3(nblk), dim3(256), 768 * sizeof(double), st, img, (const float*)f, (double*)ws, npix, C);
    else hipLaunchKernelGGL(rgb_wgrad_stage1<bf16_t>, dim3(nblk), dim3(256), 768 * sizeof(double), st, img, (const bf16_t*)f, (double*)ws, npix, C);
    SGX_LAUNCH_CHECK("rgb_wgrad_stage1");
    hipLaunchKernelGGL(rgb_wgrad_stage2, dim3((3 * C + 3) / 4), dim3(256), 0, st, (const double*)ws, dw, nblk, C, sj, sc, wscale);
    SGX_LAUNCH_CHECK("rgb_wgrad_stage2");
    return 0;
}
