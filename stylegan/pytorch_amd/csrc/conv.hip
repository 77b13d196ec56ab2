// MFMA implicit-GEMM convolutions for gfx950 (CDNA4), NHWC.
//
// One kernel template covers the three geometries the StyleGAN path needs:
//   G3X3  : 3x3 stride 1 pad 1            (EqualizedConv2d plain path + its data gradient)
//   GDOWN : 4x4 stride 2 pad 1            (fused conv+downscale; data gradient of GUP)
//   GUP   : 4x4 stride 2 pad 1 transposed (fused upscale+conv; data gradient of GDOWN), computed as four
//           output-parity classes, each a 2x2 convolution over the coarse grid
// GEMM view: M = output channels (A operand = packed weights w[tap][n][k]), N = output pixels (B operand =
// activations), K = taps x input channels.  With channels on M the 16x16 accumulator holds 4 CONSECUTIVE channels
// per lane, so the NHWC store is a 16-byte (fp32) / 8-byte (bf16) vector per lane and a wave writes whole
// 64-byte channel rows.
//
// Per block: 256 threads (4 waves), BP output pixels (NI images x TH x TW) x BCO=16*CT output channels.  Per
// K-chunk of KC input channels the input patch (with halo, zero filled) and the weights of all taps are staged in
// LDS once and reused by every tap (9x / 16x / 4x reuse of the activation bytes).
//
// fp32: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain; parity configs).  bf16: v_mfma_f32_16x16x32_bf16, or
// 16x16x16 when the layer has only 16 input channels.  Accumulation is always fp32.
#include "common.h"

enum { G3X3 = 0, GDOWN = 1, GUP = 2 };

template <typename T, int KC> struct Frag;
template <> struct Frag<float, 16> {
    static constexpr int ROWB = 68;                 // 17 dwords: conflict-free ds_read_b32 across 16 pixels
    static constexpr int NK = 4;
    typedef float frag_t;
    __device__ static __forceinline__ frag_t load(const char* row, int kk, int q) {
        return *reinterpret_cast<const float*>(row + (kk * 4 + q) * 4);
    }
    __device__ static __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void stage(char* dst, const float* src, bool ok) {
        float4 v = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = reinterpret_cast<float*>(dst);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
};
template <> struct Frag<bf16_t, 32> {
    static constexpr int ROWB = 80;                 // 64 B data + 16 B pad: conflict-free ds_read_b128
    static constexpr int NK = 1;
    typedef bf16x8 frag_t;
    __device__ static __forceinline__ frag_t load(const char* row, int, int q) {
        return *reinterpret_cast<const bf16x8*>(row + q * 16);
    }
    __device__ static __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void stage(char* dst, const bf16_t* src, bool ok) {
        uint4 v = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(dst) = v;
    }
};
template <> struct Frag<bf16_t, 16> {
    static constexpr int ROWB = 48;                 // 32 B data + 16 B pad: conflict-free ds_read_b64
    static constexpr int NK = 1;
    typedef s16x4 frag_t;
    __device__ static __forceinline__ frag_t load(const char* row, int, int q) {
        return *reinterpret_cast<const s16x4*>(row + q * 8);
    }
    __device__ static __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ void stage(char* dst, const bf16_t* src, bool ok) {
        uint4 v = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(dst) = v;
    }
};

struct ConvArgs {
    const void* x; const void* w; const float* bias; void* y;
    int B, H, W, OH, OW, OHc, OWc, Cin, Cout, act, tiles_x, tiles_y;
};

template <int GEO> struct Geo;
template <> struct Geo<G3X3> { static constexpr int IS = 1, TK = 3, NCLS = 1; };
template <> struct Geo<GDOWN> { static constexpr int IS = 2, TK = 4, NCLS = 1; };
template <> struct Geo<GUP> { static constexpr int IS = 1, TK = 2, NCLS = 4; };

template <typename T, int KC, int GEO, int TH, int TW, int BP, int CT>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs a) {
    using F = Frag<T, KC>;
    constexpr int IS = Geo<GEO>::IS, TK = Geo<GEO>::TK, NT = TK * TK;
    constexpr int PH = (TH - 1) * IS + TK, PW = (TW - 1) * IS + TK;
    constexpr int NI = BP / (TH * TW), SPW = BP / 64, BCO = CT * 16;
    constexpr int VE = 16 / (int)sizeof(T), VPP = KC / VE;
    constexpr int IN_BYTES = (NI * PH * PW * F::ROWB + 15) / 16 * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* in_lds = smem;
    char* w_lds = smem + IN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, l15 = lane & 15;
    int bx = blockIdx.x;
    const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
    const int ty_i = bx % a.tiles_y;
    const int img0 = (bx / a.tiles_y) * NI;
    const int ty0 = ty_i * TH, tx0 = tx_i * TW;
    const int co0 = blockIdx.y * BCO;
    int py = 0, px = 0;
    if (GEO == GUP) { py = blockIdx.z >> 1; px = blockIdx.z & 1; }
    const int iy0 = ty0 * IS + (GEO == GUP ? py - 1 : -1);
    const int ix0 = tx0 * IS + (GEO == GUP ? px - 1 : -1);
    const T* __restrict__ xg = static_cast<const T*>(a.x);
    const T* __restrict__ wg = static_cast<const T*>(a.w);

    int pixoff[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int m = (wave * SPW + s) * 16 + l15;
        const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
        pixoff[s] = ((il * PH + r * IS) * PW + c * IS) * F::ROWB;
    }
    f32x4 acc[CT][SPW];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int s = 0; s < SPW; ++s) acc[ct][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < a.Cin; k0 += KC) {
        if (k0) __syncthreads();
        // ---- stage the input patch (zero filled outside the image / batch)
        for (int idx = tid; idx < NI * PH * PW * VPP; idx += 256) {
            const int v = idx % VPP, pixel = idx / VPP;
            const int il = pixel / (PH * PW), rem = pixel % (PH * PW);
            const int gy = iy0 + rem / PW, gx = ix0 + rem % PW, b = img0 + il;
            const bool ok = (b < a.B) && ((unsigned)gy < (unsigned)a.H) && ((unsigned)gx < (unsigned)a.W);
            const T* src = xg + (((size_t)b * a.H + gy) * a.W + gx) * a.Cin + k0 + v * VE;
            F::stage(in_lds + pixel * F::ROWB + v * 16, src, ok);
        }
        // ---- stage the weights of every tap for this channel chunk
        for (int idx = tid; idx < NT * BCO * VPP; idx += 256) {
            const int v = idx % VPP, row = idx / VPP;
            const int n = row % BCO, t = row / BCO;
            int tg = t;
            if (GEO == GUP) tg = (3 - py - 2 * (t >> 1)) * 4 + (3 - px - 2 * (t & 1));
            const T* src = wg + ((size_t)tg * a.Cout + co0 + n) * a.Cin + k0 + v * VE;
            F::stage(w_lds + row * F::ROWB + v * 16, src, true);
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int toff = ((t / TK) * PW + (t % TK)) * F::ROWB;
#pragma unroll
            for (int kk = 0; kk < F::NK; ++kk) {
                typename F::frag_t fa[CT], fb[SPW];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) fa[ct] = F::load(w_lds + (t * BCO + ct * 16 + l15) * F::ROWB, kk, q);
#pragma unroll
                for (int s = 0; s < SPW; ++s) fb[s] = F::load(in_lds + pixoff[s] + toff, kk, q);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int s = 0; s < SPW; ++s) acc[ct][s] = F::mma(fa[ct], fb[s], acc[ct][s]);
            }
        }
    }
    // ---- epilogue: bias, activation, NHWC vector store (4 consecutive channels per lane)
    T* __restrict__ yg = static_cast<T*>(a.y);
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int m = (wave * SPW + s) * 16 + l15;
        const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
        const int b = img0 + il, oyc = ty0 + r, oxc = tx0 + c;
        if (b >= a.B || oyc >= a.OHc || oxc >= a.OWc) continue;
        const int oy = (GEO == GUP) ? 2 * oyc + py : oyc, ox = (GEO == GUP) ? 2 * oxc + px : oxc;
        T* dst = yg + (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + q * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            float v[4] = {acc[ct][s][0], acc[ct][s][1], acc[ct][s][2], acc[ct][s][3]};
            if (a.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + co0 + ct * 16 + q * 4);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if (a.act == SGX_ACT_LRELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = lrelu(v[i]);
            }
            if (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(dst + ct * 16) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                uint2 o;
                o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(dst + ct * 16) = o;
            }
        }
    }
}

template <typename T, int KC, int GEO, int TH, int TW, int BP, int CT>
static int launch_conv(const ConvArgs& a, int ngroups, hipStream_t st) {
    using F = Frag<T, KC>;
    constexpr int IS = Geo<GEO>::IS, TK = Geo<GEO>::TK;
    constexpr int PH = (TH - 1) * IS + TK, PW = (TW - 1) * IS + TK, NI = BP / (TH * TW);
    constexpr int IN_BYTES = (NI * PH * PW * F::ROWB + 15) / 16 * 16;
    constexpr int LDS = IN_BYTES + TK * TK * CT * 16 * F::ROWB;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv_kernel<T, KC, GEO, TH, TW, BP, CT>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)attr;
    dim3 grid((unsigned)(ngroups * a.tiles_y * a.tiles_x), (unsigned)(a.Cout / (CT * 16)), Geo<GEO>::NCLS);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, st, a);
    SGX_LAUNCH_CHECK("conv_kernel");
    return 0;
}

template <typename T, int KC, int GEO, int BP, int CT>
static int dispatch_tile(ConvArgs& a, hipStream_t st) {
    // tile shape by class-grid size; NI images per block fill the pixel tile at low resolution
    if (a.OHc >= 16 && a.OWc >= 16) {
        constexpr int TH = BP / 16, TW = 16;
        a.tiles_y = (a.OHc + TH - 1) / TH; a.tiles_x = (a.OWc + TW - 1) / TW;
        return launch_conv<T, KC, GEO, TH, TW, BP, CT>(a, a.B, st);
    } else if (a.OHc >= 8 && a.OWc >= 8) {
        constexpr int NI = BP / 64;
        a.tiles_y = (a.OHc + 7) / 8; a.tiles_x = (a.OWc + 7) / 8;
        return launch_conv<T, KC, GEO, 8, 8, BP, CT>(a, (a.B + NI - 1) / NI, st);
    } else {
        constexpr int NI = BP / 16;
        a.tiles_y = (a.OHc + 3) / 4; a.tiles_x = (a.OWc + 3) / 4;
        return launch_conv<T, KC, GEO, 4, 4, BP, CT>(a, (a.B + NI - 1) / NI, st);
    }
}

template <typename T, int KC, int GEO, int BP>
static int dispatch_ct(ConvArgs& a, int max_ct, hipStream_t st) {
    if (max_ct >= 4 && a.Cout % 64 == 0) return dispatch_tile<T, KC, GEO, BP, 4>(a, st);
    if (max_ct >= 2 && a.Cout % 32 == 0) return dispatch_tile<T, KC, GEO, BP, 2>(a, st);
    return dispatch_tile<T, KC, GEO, BP, 1>(a, st);
}

template <int GEO, int BP>
static int dispatch_conv(ConvArgs& a, int dtype, int max_ct, hipStream_t st) {
    SGX_REQUIRE(a.Cout % 16 == 0 && a.Cin % 16 == 0, SGX_EUNSUPPORTED, "conv: channels must be multiples of 16 (Cin=%d Cout=%d)", a.Cin, a.Cout);
    SGX_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0, SGX_EINVAL, "conv: bad shape");
    if (dtype == SGX_F32) return dispatch_ct<float, 16, GEO, BP>(a, max_ct, st);
    if (dtype == SGX_BF16) {
        if (a.Cin % 32 == 0) return dispatch_ct<bf16_t, 32, GEO, BP>(a, max_ct, st);
        return dispatch_ct<bf16_t, 16, GEO, BP>(a, max_ct, st);
    }
    SGX_REQUIRE(false, SGX_EINVAL, "conv: bad dtype %d", dtype);
}

extern "C" int sgx_conv3x3(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin,
                           int Cout, int act, int dtype, void* stream) {
    ConvArgs a{x, w, bias, y, B, H, W, H, W, H, W, Cin, Cout, act, 0, 0};
    return dispatch_conv<G3X3, 256>(a, dtype, 4, (hipStream_t)stream);
}

extern "C" int sgx_conv4x4s2_down(const void* x, const void* w, const float* bias, void* y, int B, int H, int W,
                                  int Cin, int Cout, int act, int dtype, void* stream) {
    SGX_REQUIRE(H % 2 == 0 && W % 2 == 0, SGX_EINVAL, "conv4x4s2_down: odd input size");
    ConvArgs a{x, w, bias, y, B, H, W, H / 2, W / 2, H / 2, W / 2, Cin, Cout, act, 0, 0};
    return dispatch_conv<GDOWN, 128>(a, dtype, 2, (hipStream_t)stream);
}

extern "C" int sgx_conv4x4s2_up(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout,
                                int dtype, void* stream) {
    ConvArgs a{x, w, nullptr, y, B, H, W, 2 * H, 2 * W, H, W, Cin, Cout, SGX_ACT_NONE, 0, 0};
    return dispatch_conv<GUP, 256>(a, dtype, 4, (hipStream_t)stream);
}

// =====================================================================================================
// Weight gradients.  GEMM view: M = "n" channels (dy / coarse side, A operand), N = "k" channels (x / fine side,
// B operand), reduction K = pixels.  Each block owns a (16*NSUB) x (16*KSUB) channel pair and walks a strided
// subset of the pixel tiles (persistent loop), keeping all taps' accumulators in registers; the per-split
// partials are then summed by reduce_splits (deterministic: no float atomics).
// =====================================================================================================
struct WgradArgs {
    const void* kside; const void* nside; float* out;     // out: [nsplit][NT][Cn][Ck] partials (or dw when nsplit==1)
    int B, Hk, Wk, Hn, Wn, Ck, Cn, tiles_x, tiles_y, ntiles;
};

template <typename T> struct WFrag;
template <> struct WFrag<float> {
    static constexpr int KPS = 4;                               // pixels consumed per MFMA
    static constexpr int row_bytes(int ch) { return (ch == 16 ? 16 : ch + 16) * 4; }   // == 16 dwords (mod 32)
    typedef float frag_t;
    // lane (l15, q): element [pixel q of this k-step][channel l15]
    __device__ static __forceinline__ frag_t load(const char* base, const int (&poff)[4], int q, int choff) {
        return *reinterpret_cast<const float*>(base + poff[0] + choff * 4);
    }
    __device__ static __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};
template <> struct WFrag<bf16_t> {
    static constexpr int KPS = 16;
    static constexpr int row_bytes(int ch) { return ch * 2 + 16; }
    typedef s16x4 frag_t;
    // lane (l15, q): elements [pixels 4q..4q+3 of this k-step][channel l15]  (scalar LDS reads: first version)
    __device__ static __forceinline__ frag_t load(const char* base, const int (&poff)[4], int q, int choff) {
        s16x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const short*>(base + poff[j] + choff * 2);
        return r;
    }
    __device__ static __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    }
};

template <typename T, int GEO, int TH, int TW, int BP, int NSUB, int KSUB>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    using F = WFrag<T>;
    constexpr int IS = Geo<GEO>::IS, TK = Geo<GEO>::TK, NT = TK * TK;
    constexpr int PH = (TH - 1) * IS + TK, PW = (TW - 1) * IS + TK;
    constexpr int NI = BP / (TH * TW);
    constexpr int NCH = NSUB * 16, KCH = KSUB * 16;
    constexpr int NROW = F::row_bytes(NCH), KROW = F::row_bytes(KCH);
    constexpr int VE = 16 / (int)sizeof(T);
    constexpr int TPW = (NT + 3) / 4;                          // taps per wave
    constexpr int PPL = F::KPS / 4;                            // pixels per lane per k-step (1 fp32, 4 bf16)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* n_lds = smem;
    char* k_lds = smem + BP * NROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, l15 = lane & 15;
    const int kblocks = a.Ck / KCH;
    const int n0 = (blockIdx.x / kblocks) * NCH, kc0 = (blockIdx.x % kblocks) * KCH;
    const T* __restrict__ kg = static_cast<const T*>(a.kside);
    const T* __restrict__ ng = static_cast<const T*>(a.nside);

    f32x4 acc[TPW][NSUB][KSUB];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int k = 0; k < KSUB; ++k) acc[i][j][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.y; tile < a.ntiles; tile += gridDim.y) {
        int bx = tile;
        const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
        const int ty_i = bx % a.tiles_y;
        const int img0 = (bx / a.tiles_y) * NI;
        const int ty0 = ty_i * TH, tx0 = tx_i * TW;
        const int iy0 = ty0 * IS - 1, ix0 = tx0 * IS - 1;
        __syncthreads();
        // n-side tile: BP pixels x NCH channels
        for (int idx = tid; idx < BP * (NCH / VE); idx += 256) {
            const int v = idx % (NCH / VE), m = idx / (NCH / VE);
            const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
            const int b = img0 + il, gy = ty0 + r, gx = tx0 + c;
            const bool ok = (b < a.B) && (gy < a.Hn) && (gx < a.Wn);
            const T* src = ng + (((size_t)b * a.Hn + gy) * a.Wn + gx) * a.Cn + n0 + v * VE;
            uint4 val = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(n_lds + m * NROW + v * 16) = val;
        }
        // k-side patch with halo: NI*PH*PW pixels x KCH channels
        for (int idx = tid; idx < NI * PH * PW * (KCH / VE); idx += 256) {
            const int v = idx % (KCH / VE), pixel = idx / (KCH / VE);
            const int il = pixel / (PH * PW), rem = pixel % (PH * PW);
            const int gy = iy0 + rem / PW, gx = ix0 + rem % PW, b = img0 + il;
            const bool ok = (b < a.B) && ((unsigned)gy < (unsigned)a.Hk) && ((unsigned)gx < (unsigned)a.Wk);
            const T* src = kg + (((size_t)b * a.Hk + gy) * a.Wk + gx) * a.Ck + kc0 + v * VE;
            uint4 val = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(k_lds + pixel * KROW + v * 16) = val;
        }
        __syncthreads();
        for (int ks = 0; ks < BP / F::KPS; ++ks) {
            int noff[4], koff[4];
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const int m = ks * F::KPS + q * PPL + j;
                const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
                noff[j] = m * NROW;
                koff[j] = ((il * PH + r * IS) * PW + c * IS) * KROW;
            }
            typename F::frag_t fa[NSUB];
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) fa[ns] = F::load(n_lds, noff, q, ns * 16 + l15);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int t = wave + 4 * i;
                if (t < NT) {
                    const int toff = ((t / TK) * PW + (t % TK)) * KROW;
#pragma unroll
                    for (int kk = 0; kk < KSUB; ++kk) {
                        typename F::frag_t fb = F::load(k_lds + toff, koff, q, kk * 16 + l15);
#pragma unroll
                        for (int ns = 0; ns < NSUB; ++ns) acc[i][ns][kk] = F::mma(fa[ns], fb, acc[i][ns][kk]);
                    }
                }
            }
        }
    }
    // partial store: out[split][t][n][k]; lane holds rows n = q*4+r, column k = l15
    float* out = a.out + (size_t)blockIdx.y * NT * a.Cn * a.Ck;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + 4 * i;
        if (t < NT) {
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
                for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out[((size_t)t * a.Cn + n0 + ns * 16 + q * 4 + r) * a.Ck + kc0 + kk * 16 + l15] = acc[i][ns][kk][r];
        }
    }
}

// ---- finishing pass: sum the split partials AND map the packed gradient back to the parameter layout
// dW[O][I][3][3] (adjoint of sgx_pack_weight), in one kernel.  Partial element (t, o, i) lives at
//   ws[s*total + ((tt*A + a)*Bd + b)],  tt = flip_t ? 8-t : t,  (a,b,A,Bd) = transposed ? (i,o,Ip,O) : (o,i,O,Ip)
// mode S: dW[o][i][y][x] = scale * P[y*3+x];  4x4 modes: dW[o][i][y][x] = scale * sum_{a,b in {0,1}} P[(y+a)*4 + (x+b)]
// (x0.25 for the down kernel; spatial flip of (y,x) for the non-fused-up semantics) -- the transposes of
// reference models/CustomLayers.py:146-150 and :159-162.
struct FinishArgs { const float* ws; float* dw; int nsplit, O, I, Ip, mode, transposed, flip_t; float scale; };

__global__ __launch_bounds__(256) void wgrad_finish_kernel(FinishArgs f) {
    __shared__ float red[16][16][9];
    const int pl = threadIdx.x & 15, sl = threadIdx.x >> 4;          // pair lane (consecutive i), split lane
    const int pair = blockIdx.x * 16 + pl;
    const int npairs = f.O * f.I;
    const int o = pair / f.I, i = pair % f.I;
    const int taps = f.mode == SGX_PACK_S ? 9 : 16;
    const size_t total = (size_t)taps * f.O * f.Ip;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    if (pair < npairs) {
        const size_t tstride = (size_t)f.O * f.Ip;
        const size_t eoff = f.transposed ? (size_t)i * f.O + o : (size_t)o * f.Ip + i;
        for (int s = sl; s < f.nsplit; s += 16) {
            const float* p = f.ws + (size_t)s * total + eoff;
            if (f.mode == SGX_PACK_S) {
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] += p[(size_t)(f.flip_t ? 8 - t : t) * tstride];
            } else {
                float v[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = p[(size_t)t * tstride];
#pragma unroll
                for (int y = 0; y < 3; ++y)
#pragma unroll
                    for (int x = 0; x < 3; ++x)
                        acc[y * 3 + x] += (v[y * 4 + x] + v[y * 4 + x + 1]) + (v[(y + 1) * 4 + x] + v[(y + 1) * 4 + x + 1]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[sl][pl][k] = acc[k];
    __syncthreads();
    if (sl == 0 && pair < npairs) {
        const float c = f.scale * (f.mode == SGX_PACK_D ? 0.25f : 1.f);
        float* d = f.dw + (size_t)pair * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float sum = 0.f;
#pragma unroll
            for (int l = 0; l < 16; ++l) sum += red[l][pl][k];
            d[f.mode == SGX_PACK_UF ? 8 - k : k] = c * sum;
        }
    }
}

static int wgrad_nsplit(int pairs, int ntiles) {
    int want = (1024 + pairs - 1) / pairs;
    if (want > 512) want = 512;
    if (want > ntiles) want = ntiles;
    if (want < 1) want = 1;
    return want;
}

template <typename T, int GEO, int TH, int TW, int BP, int NSUB, int KSUB>
static int launch_wgrad(WgradArgs& a, void* ws, size_t ws_bytes, int* nsplit_out, hipStream_t st) {
    using F = WFrag<T>;
    constexpr int IS = Geo<GEO>::IS, TK = Geo<GEO>::TK, NT = TK * TK;
    constexpr int PH = (TH - 1) * IS + TK, PW = (TW - 1) * IS + TK, NI = BP / (TH * TW);
    constexpr int LDS = BP * F::row_bytes(NSUB * 16) + NI * PH * PW * F::row_bytes(KSUB * 16);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    a.tiles_y = (a.Hn + TH - 1) / TH; a.tiles_x = (a.Wn + TW - 1) / TW;
    a.ntiles = ((a.B + NI - 1) / NI) * a.tiles_y * a.tiles_x;
    const int pairs = (a.Cn / (NSUB * 16)) * (a.Ck / (KSUB * 16));
    int nsplit = wgrad_nsplit(pairs, a.ntiles);
    const size_t total = (size_t)NT * a.Cn * a.Ck;
    size_t fit = ws_bytes / (total * sizeof(float));
    if ((size_t)nsplit > fit) nsplit = (int)fit;
    SGX_REQUIRE(nsplit >= 1, SGX_EWORKSPACE, "wgrad: workspace too small (%zu bytes, need >= %zu)", ws_bytes, total * sizeof(float));
    a.out = static_cast<float*>(ws);
    auto kern = wgrad_kernel<T, GEO, TH, TW, BP, NSUB, KSUB>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)attr;
    hipLaunchKernelGGL(kern, dim3(pairs, nsplit), dim3(256), LDS, st, a);
    SGX_LAUNCH_CHECK("wgrad_kernel");
    *nsplit_out = nsplit;
    return 0;
}

template <typename T, int GEO, int BP, int NSUB, int KSUB>
static int wgrad_tile(WgradArgs& a, void* ws, size_t wsb, int* ns, hipStream_t st) {
    if (a.Hn >= 16 && a.Wn >= 16) return launch_wgrad<T, GEO, BP / 16, 16, BP, NSUB, KSUB>(a, ws, wsb, ns, st);
    if (a.Hn >= 8 && a.Wn >= 8) return launch_wgrad<T, GEO, (BP >= 64 ? 8 : BP / 8), 8, BP, NSUB, KSUB>(a, ws, wsb, ns, st);
    return launch_wgrad<T, GEO, (BP >= 16 ? 4 : BP / 4), 4, BP, NSUB, KSUB>(a, ws, wsb, ns, st);
}

template <typename T, int GEO, int BP>
static int wgrad_ch(WgradArgs& a, void* ws, size_t wsb, int* ns, hipStream_t st) {
    SGX_REQUIRE(a.Cn % 16 == 0 && a.Ck % 16 == 0, SGX_EUNSUPPORTED, "wgrad: channels must be multiples of 16");
    const bool n32 = a.Cn % 32 == 0, k32 = a.Ck % 32 == 0;
    if (GEO == GDOWN) {                                       // fine patch is 4x larger: keep the k side at 16 channels
        if (n32) return wgrad_tile<T, GEO, BP, 2, 1>(a, ws, wsb, ns, st);
        return wgrad_tile<T, GEO, BP, 1, 1>(a, ws, wsb, ns, st);
    }
    if (n32 && k32) return wgrad_tile<T, GEO, BP, 2, 2>(a, ws, wsb, ns, st);
    if (n32) return wgrad_tile<T, GEO, BP, 2, 1>(a, ws, wsb, ns, st);
    if (k32) return wgrad_tile<T, GEO, BP, 1, 2>(a, ws, wsb, ns, st);
    return wgrad_tile<T, GEO, BP, 1, 1>(a, ws, wsb, ns, st);
}

static int wgrad_finish(const void* ws, float* dw, int nsplit, int O, int I, int Ip, int mode, int transposed, int flip_t,
                        float scale, hipStream_t st) {
    FinishArgs f{static_cast<const float*>(ws), dw, nsplit, O, I, Ip, mode, transposed, flip_t, scale};
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((O * I + 15) / 16)), dim3(256), 0, st, f);
    SGX_LAUNCH_CHECK("wgrad_finish_kernel");
    return 0;
}

extern "C" size_t sgx_wgrad_ws_bytes(int taps, int B, int H, int W, int Ck, int Cn) {
    size_t total = (size_t)taps * Ck * Cn * sizeof(float);
    int pairs = (Ck / 32 > 0 ? Ck / 32 : 1) * (Cn / 32 > 0 ? Cn / 32 : 1);
    size_t ntiles = (size_t)B * ((H + 3) / 4) * ((W + 3) / 4);
    size_t ns = (size_t)wgrad_nsplit(pairs, ntiles > (1u << 30) ? (1 << 30) : (int)ntiles);
    return total * ns;
}

extern "C" int sgx_wgrad3x3_param(const void* x, const void* dy, float* dW, void* ws, size_t ws_bytes, int B, int H, int W,
                                  int Cx, int Cdy, int adjoint, float scale, int O, int I, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int Ip = adjoint ? Cdy : Cx;
    SGX_REQUIRE((adjoint ? Cx : Cdy) == O && Ip >= I, SGX_EINVAL, "wgrad3x3_param: channel mismatch (Cx=%d Cdy=%d O=%d I=%d adj=%d)", Cx, Cdy, O, I, adjoint);
    WgradArgs a{x, dy, nullptr, B, H, W, H, W, Cx, Cdy, 0, 0, 0};
    int ns = 0, rc;
    if (dtype == SGX_F32) rc = wgrad_ch<float, G3X3, 128>(a, ws, ws_bytes, &ns, st);
    else if (dtype == SGX_BF16) rc = wgrad_ch<bf16_t, G3X3, 128>(a, ws, ws_bytes, &ns, st);
    else { SGX_REQUIRE(false, SGX_EINVAL, "wgrad3x3_param: bad dtype"); }
    if (rc) return rc;
    return wgrad_finish(ws, dW, ns, O, I, Ip, SGX_PACK_S, adjoint, adjoint, scale, st);
}

extern "C" int sgx_wgrad4x4s2_param(const void* fine, const void* coarse, float* dW, void* ws, size_t ws_bytes, int B, int H,
                                    int W, int Cfine, int Ccoarse, int mode, float scale, int O, int I, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(H % 2 == 0 && W % 2 == 0, SGX_EINVAL, "wgrad4x4s2_param: odd fine size");
    SGX_REQUIRE(mode == SGX_PACK_D || mode == SGX_PACK_U || mode == SGX_PACK_UF, SGX_EINVAL, "wgrad4x4s2_param: bad mode %d", mode);
    const int transposed = mode != SGX_PACK_D;                 // kernel output is [t][coarse ch][fine ch]
    SGX_REQUIRE(transposed ? (Ccoarse == I && Cfine == O) : (Ccoarse == O && Cfine == I), SGX_EINVAL,
                "wgrad4x4s2_param: channel mismatch (fine %d coarse %d O %d I %d mode %d)", Cfine, Ccoarse, O, I, mode);
    WgradArgs a{fine, coarse, nullptr, B, H, W, H / 2, W / 2, Cfine, Ccoarse, 0, 0, 0};
    int ns = 0, rc;
    if (dtype == SGX_F32) rc = wgrad_ch<float, GDOWN, 64>(a, ws, ws_bytes, &ns, st);
    else if (dtype == SGX_BF16) rc = wgrad_ch<bf16_t, GDOWN, 64>(a, ws, ws_bytes, &ns, st);
    else { SGX_REQUIRE(false, SGX_EINVAL, "wgrad4x4s2_param: bad dtype"); }
    if (rc) return rc;
    return wgrad_finish(ws, dW, ns, O, I, I, mode, transposed, 0, scale, st);
}

// =====================================================================================================
// Weight packing: parameter [O][I][3][3] fp32 -> the two MFMA operand packs of a layer, in the activation
// dtype, in one launch: fwd[t][O][Ip] for the layer's own convolution and adj[t'][Ip][O] for its data gradient
// (t' = 8-t for the 3x3 kernel, whose adjoint is spatially flipped; t' = t for the 4x4 stride-2 pair).
//   S : v = scale * w[o][i][ty][tx]                                         (models/CustomLayers.py:170-171)
//   D : v = 0.25*scale * sum_{a,b} w[o][i][ky-a][kx-b]                      (:159-162)
//   U : v = scale * sum_{a,b} w[o][i][ky-a][kx-b];  UF: same on the flipped 3x3 kernel   (:146-150)
// =====================================================================================================
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ fwd, T* __restrict__ adj, int O, int I, int Ip,
                                   int mode, float scale) {
    const int taps = mode == SGX_PACK_S ? 9 : 16;
    const size_t n = (size_t)taps * O * Ip;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % Ip);
        const int o = (int)((e / Ip) % O);
        const int t = (int)(e / ((size_t)Ip * O));
        float v = 0.f;
        if (i < I) {
            const float* wp = w + ((size_t)o * I + i) * 9;
            if (mode == SGX_PACK_S) {
                v = scale * wp[t];
            } else {
                const int ky = t >> 2, kx = t & 3;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int y = ky - a, x = kx - b;
                        if (y >= 0 && y < 3 && x >= 0 && x < 3) v += wp[mode == SGX_PACK_UF ? (2 - y) * 3 + (2 - x) : y * 3 + x];
                    }
                v *= scale * (mode == SGX_PACK_D ? 0.25f : 1.f);
            }
        }
        fwd[e] = from_f<T>(v);
        const int ta = mode == SGX_PACK_S ? 8 - t : t;
        adj[((size_t)ta * Ip + i) * O + o] = from_f<T>(v);
    }
}

extern "C" int sgx_pack_weight(const float* w, void* fwd, void* adj, int O, int I, int Ipad, int mode, float scale, int dtype,
                               void* stream) {
    SGX_REQUIRE(mode >= SGX_PACK_S && mode <= SGX_PACK_UF && Ipad >= I && O > 0 && I > 0, SGX_EINVAL, "pack_weight: bad args");
    const size_t n = (size_t)(mode == SGX_PACK_S ? 9 : 16) * O * Ipad;
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (dtype == SGX_F32) hipLaunchKernelGGL(pack_weight_kernel<float>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, (float*)fwd, (float*)adj, O, I, Ipad, mode, scale);
    else if (dtype == SGX_BF16) hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)fwd, (bf16_t*)adj, O, I, Ipad, mode, scale);
    else { SGX_REQUIRE(false, SGX_EINVAL, "pack_weight: bad dtype"); }
    SGX_LAUNCH_CHECK("pack_weight_kernel");
    return 0;
}
