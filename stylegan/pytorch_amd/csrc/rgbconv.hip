// The discriminator's FIRST layer pair at the current resolution as ONE 3-channel convolution (round 4).
//
// Reference: Discriminator.forward (models/GAN.py:413-427) feeds the image through from_rgb -- a 1x1 EqualizedConv2d WITHOUT an
// activation (models/GAN.py:353) -- and then through DiscriminatorBlock.conv0, a 3x3 EqualizedConv2d, LeakyReLU and the blur
// (models/Blocks.py:137-142).  Two linear maps in a row compose:
//     conv0(from_rgb(img))[p][o] = b0[o] + sum_{tap valid at p} sum_j W'[o][j][tap] img[p + tap][j] + sum_{tap valid at p} T[o][tap]
//     W'[o][j][tap] = s0 sr sum_i W0[o][i][tap] Wr[i][j],     T[o][tap] = s0 sum_i W0[o][i][tap] br[i]
// (from_rgb's bias reaches conv0 only through taps that fall INSIDE the image: conv0 pads from_rgb's output with zeros, not with
// its bias.)  T rides in a 4th input channel that is 1 inside the image and 0 outside, so the border needs no special case.
// What this removes per discriminator pass at 1024^2: the from_rgb pass (12 B/pixel read, 32 written), conv0's 32 B/pixel read, the
// write + read of the pre-activation around the blur, and in the backward conv0's data gradient, from_rgb's weight-gradient pass
// and (where the image needs a gradient: R1, generator step) the to-RGB adjoint -- ~120 B/pixel forward become 46.
//
// Kernels (bf16 activations; fp32 images, parameters, gradients):
//   rgbconv_pack_kernel        W0, Wr, br -> the two MFMA operand packs (forward: [ky][o][kx*4+j], fp16; adjoint: [ky'][kx'*4+j][o], bf16)
//   rgbconv_fwd_kernel<CB,EPI> EPI 1: xb = blur(lrelu(conv + b0)) + sign bits of the pre-activation; EPI 0: the plain convolution
//                              (the adjoint's own backward under the R1 double backward).  Image tile with halo -> LDS as bf16
//                              (r,g,b,1); 16 pixels x 16 channels per v_mfma_f32_16x16x16_bf16 triple (one per kernel row, K =
//                              3 pixels x 4 channels + 4 padding); activated tile (+1 halo) -> LDS; separable blur from LDS
//   rgbconv_fwdblur_kernel<CB> the same result as EPI 1 from a row-streaming wave (no LDS tile, no barrier): the default
//   rgbconv_dgrad_kernel<CB>   image gradient  gi[q][j] = sum_{tap,o} gz[q - tap][o] W'[o][j][tap], row-streaming: M = (kernel column, colour),
//                              K = (kernel row, channel): 3 MFMAs per 16 pixels, the three kernel columns summed across lanes
//   rgbconv_wgrad_kernel<CB>   dW'[o][tap][j] = sum_p gz[p][o] img1[p + tap][j]: persistent blocks over 8x64-pixel tiles, operands
//                              transposed into planar LDS images so that 4 consecutive PIXELS are one aligned 8-byte fragment read
//   rgbconv_wfinish_kernel     deterministic sum of the block partials; rgbconv_chain_kernel: dW' -> dW0, db0, dWr, dbr (chain rule
//                              through the composition), written or accumulated in the parameters' own layouts
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// The FORWARD kernels take the image and the composed weights as fp16, not bf16: the image is the network's input, and rounding it
// to bf16 (2^-9) raised the discriminator's gradient error against fp64 from 0.075 to 0.10 median rel-L2 -- every activation
// downstream inherits the input's error, where the unfused path rounds from_rgb's 16-channel OUTPUT, whose errors average out over
// conv0's 144 inputs.  fp16 (2^-11; images are in [-1, 1], the weights O(1)) costs no instruction more; carrying the image as two
// bf16 terms (hi + lo, 6 MFMAs per row) was measured too: same accuracy as the unfused path, +12 % time.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {   // round to nearest even
    const h16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned, v);
}
static __device__ __forceinline__ f32x4_t mma16h(s16x4 a, s16x4 b, f32x4_t c) {   // same layouts as mma16, fp16 operands
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
}
static __device__ __forceinline__ f32x4_t mma16(s16x4 a, s16x4 b, f32x4_t c) {
    // A[i = lane & 15][k = 4 (lane >> 4) + 0..3], B[k][j = lane & 15], D[i = 4 (lane >> 4) + reg][j = lane & 15]
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------ packs
// wf: [3 ky][C o][16 k], k = kx * 4 + j (j = 3: the bias channel), k 12..15 zero.   wd: [3 ky'][16 i][C o], i = kx' * 4 + j (j = 3 and
// kx' = 3 zero): tap (ky', kx') reads gz at q + (ky' - 1, kx' - 1), i.e. it is forward tap (2 - ky', 2 - kx').
__global__ void rgbconv_pack_kernel(const float* __restrict__ w0, float s0, const float* __restrict__ wr, float sr,
                                    const float* __restrict__ br, float bscale, bf16_t* __restrict__ wf, bf16_t* __restrict__ wd, int C) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x, nth = blockDim.x * gridDim.x;
    for (int e = tid; e < 3 * C * 16; e += nth) {
        const int ky = e / (C * 16), o = (e / 16) % C, k = e % 16, kx = k >> 2, j = k & 3;
        float v = 0.f;
        if (kx < 3) {
            for (int i = 0; i < C; ++i) {
                const float w = w0[((o * C + i) * 3 + ky) * 3 + kx];
                v += w * (j < 3 ? wr[i * 3 + j] : (br ? br[i] * bscale : 0.f));
            }
            v *= j < 3 ? s0 * sr : s0;
        }
        const _Float16 hv = (_Float16)v;                        // the forward pack is fp16 (see mma16h)
        wf[e] = __builtin_bit_cast(bf16_t, hv);
    }
    for (int e = tid; e < 3 * 16 * C; e += nth) {
        const int kyp = e / (16 * C), i = (e / C) % 16, o = e % C, kxp = i >> 2, j = i & 3, ky = 2 - kyp, kx = 2 - kxp;
        float v = 0.f;
        if (j < 3 && kxp < 3) {
            for (int ii = 0; ii < C; ++ii) v += w0[((o * C + ii) * 3 + ky) * 3 + kx] * wr[ii * 3 + j];
            v *= s0 * sr;
        }
        wd[e] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------------------------ forward
template <int CB, int EPI> struct RcFwd {
    static constexpr int C = 16 * CB, R = EPI ? 2 : 1;
    static constexpr int TH = EPI ? (CB == 1 ? 16 : 8) : 16, TW = 64;
    static constexpr int IH = TH + 2 * R, IW = TW + 2 * R;                 // staged image region
    static constexpr int ZH = TH + 2 * (R - 1), ZW = TW + 2 * (R - 1);     // convolution outputs the block computes
    static constexpr int NPX = ZH * ZW, NG = (NPX + 15) / 16;
    static constexpr int IPX = IH * IW + 4;                                // + 4 pixels of padding: the k-group of the last pixel reads 3 beyond
    static constexpr int IMG_BYTES = (IPX * 8 + 15) / 16 * 16;
    static constexpr int LDS = IMG_BYTES + (EPI ? NPX * C * 2 : 16);          // EPI 0: + the four waves' input maxima (power-of-two prescale)
};

// Tile schedule of the persistent kernels below: the grid is a multiple of 8 blocks; block b runs on XCD b & 7 (round-robin
// dispatch) and walks every (grid / 8)-th tile of that XCD's contiguous eighth of the tile raster, so tiles that share halo rows
// are processed by neighbouring blocks of ONE XCD close in time (their re-reads hit that XCD's L2).
struct RcSched { int first, end, stride; };
static __device__ __forceinline__ RcSched rc_sched(int ntiles) {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, per = gridDim.x >> 3, band = (ntiles + 7) >> 3;
    RcSched s;
    s.first = xcd * band + slot;
    s.end = (xcd + 1) * band < ntiles ? (xcd + 1) * band : ntiles;
    s.stride = per;
    return s;
}
struct rgb3 { float r, g, b; };                            // 12 bytes, 4-byte aligned: one global_load_dwordx3

template <int CB, int EPI>
__global__ __launch_bounds__(256) void rgbconv_fwd_kernel(const float* __restrict__ img, const bf16_t* __restrict__ wf, const float* __restrict__ b0,
                                                          bf16_t* __restrict__ y, unsigned char* __restrict__ bits, int B, int H, int W, int ones,
                                                          int tiles_x, int tiles_y, int ntiles) {
    using G = RcFwd<CB, EPI>;
    constexpr int C = G::C, R = G::R, TH = G::TH, TW = G::TW, IH = G::IH, IW = G::IW, ZW = G::ZW, NPX = G::NPX, NG = G::NG, IPX = G::IPX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2* imgl = reinterpret_cast<uint2*>(smem);
    char* zl = smem + G::IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const RcSched sc = rc_sched(ntiles);
    if (sc.first >= sc.end) return;
    // weights of this lane: A[i = o][k] for the three kernel rows
    s16x4 wfr[CB][3];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) wfr[cb][ky] = *reinterpret_cast<const s16x4*>(wf + ((ky * C + cb * 16 + l15) * 16 + 4 * l4));
    float4 bias[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
        bias[cb] = (EPI && b0) ? *reinterpret_cast<const float4*>(b0 + cb * 16 + 4 * l4) : make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- image region (+halo) of a tile -> registers (the NEXT tile's loads are in flight while this one is computed)
    constexpr int NIT = (IPX + 255) / 256;
    rgb3 pv[NIT];
    unsigned okm = 0;
    auto tile_at = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        b = t / tiles_y; ty0 = ty * TH; tx0 = tx * TW;
    };
    auto load_tile = [&](int t) {
        int b, ty0, tx0;
        tile_at(t, b, ty0, tx0);
        okm = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid, r = idx / IW, c = idx - r * IW;
            const int gy = ty0 - R + r, gx = tx0 - R + c;
            const bool ok = idx < IH * IW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            pv[it] = rgb3{0.f, 0.f, 0.f};
            if (ok) {
                pv[it] = *reinterpret_cast<const rgb3*>(img + (((size_t)b * H + gy) * W + gx) * 3);
                okm |= 1u << it;
            }
        }
    };
    load_tile(sc.first);
    for (int t = sc.first; t < sc.end; t += sc.stride) {
        int b, ty0, tx0;
        tile_at(t, b, ty0, tx0);
        // EPI 0 (the plain convolution: the adjoint's own backward under the R1 double backward) is fed a GRADIENT, not an image:
        // gamma / B * dD/dimg per pixel is 1e-6..1e-5 at 1024^2, below fp16's normal range (6.1e-5).  The operation is linear, so the
        // tile is brought into fp16 range by a power of two taken from its own largest magnitude (exact), and the accumulators
        // are multiplied by the inverse power (exact) before they are rounded to bf16: full fp16 precision relative to the tile's
        // maximum whatever the magnitude of the input -- what fp32 operands would give up to 2^-11 of the largest neighbour
        float in_scale = 1.f, out_scale = 1.f;
        if constexpr (EPI == 0) {
            float am = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) am = fmaxf(fmaxf(am, fabsf(pv[it].r)), fmaxf(fabsf(pv[it].g), fabsf(pv[it].b)));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
            float* amx = reinterpret_cast<float*>(smem + G::IMG_BYTES);
            if (lane == 0) amx[wave] = am;
            __syncthreads();                               // (every thread read the previous tile's maxima before that tile's staging barrier)
            am = fmaxf(fmaxf(amx[0], amx[1]), fmaxf(amx[2], amx[3]));
            const int e = (int)((__float_as_uint(am) >> 23) & 0xffu);          // am in [2^(e-127), 2^(e-126))
            int se = e == 0 ? 127 : 127 + 14 - (e - 127);                       // scale = 2^(14 - (e - 127)): am * scale in [2^14, 2^15)
            se = se < 1 ? 1 : (se > 253 ? 253 : se);
            in_scale = __uint_as_float((unsigned)se << 23);
            out_scale = __uint_as_float((unsigned)(254 - se) << 23);
        }
        // ---- phase A: registers -> LDS as fp16 (r, g, b, 1 | 0); zero outside the image (the convolution's padding)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid;
            if (idx < IPX) imgl[idx] = make_uint2(pack_f16x2(pv[it].r * in_scale, pv[it].g * in_scale),
                                                  pack_f16x2(pv[it].b * in_scale, (((okm >> it) & 1u) && ones) ? 1.f : 0.f));
        }
        __syncthreads();                                   // the region is staged; everybody is done with the previous tile's blur
        if (t + sc.stride < sc.end) load_tile(t + sc.stride);

        // ---- phase B: 16 pixels x 16 channels per MFMA triple
        for (int g = wave; g < NG; g += 4) {
            const int px = g * 16 + l15, pxc = px < NPX ? px : NPX - 1;
            const int zr = pxc / ZW, zc = pxc - zr * ZW;
            f32x4_t acc[CB];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const uint2 raw = imgl[(zr + ky) * IW + zc + l4];
                s16x4 bf;
                bf[0] = (short)(raw.x & 0xffffu); bf[1] = (short)(raw.x >> 16); bf[2] = (short)(raw.y & 0xffffu); bf[3] = (short)(raw.y >> 16);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[cb] = mma16h(wfr[cb][ky], bf, acc[cb]);
            }
            const int gy = ty0 - (R - 1) + zr, gx = tx0 - (R - 1) + zc;
            const bool inimg = px < NPX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                if constexpr (EPI == 0) {
                    if (inimg)
                        *reinterpret_cast<uint2*>(y + (((size_t)b * H + gy) * W + gx) * C + cb * 16 + 4 * l4) =
                            make_uint2(pack_bf16x2(acc[cb][0] * out_scale, acc[cb][1] * out_scale), pack_bf16x2(acc[cb][2] * out_scale, acc[cb][3] * out_scale));
                } else {
                    // pre-activation + bias -> LeakyReLU; positions outside the image are the BLUR's zero padding
                    const float a0 = inimg ? lrelu(acc[cb][0] + bias[cb].x) : 0.f, a1 = inimg ? lrelu(acc[cb][1] + bias[cb].y) : 0.f;
                    const float a2 = inimg ? lrelu(acc[cb][2] + bias[cb].z) : 0.f, a3 = inimg ? lrelu(acc[cb][3] + bias[cb].w) : 0.f;
                    if (px < NPX) *reinterpret_cast<uint2*>(zl + ((size_t)px * C + cb * 16 + 4 * l4) * 2) = make_uint2(pack_bf16x2(a0, a1), pack_bf16x2(a2, a3));
                }
            }
        }
        __syncthreads();                                   // the activated tile is complete; the image region may be overwritten
        if constexpr (EPI == 1) {
            // ---- phase C: separable [1,2,1]x[1,2,1]/16 down a column strip: thread = (column, 8-channel vector[, row half])
            constexpr int VPP = C / 8, NSTRIP = TW * VPP, RSPLIT = 256 / NSTRIP, RPT = TH / RSPLIT;
            static_assert(256 % NSTRIP == 0 && TH % RSPLIT == 0, "strip geometry");
            const int s = tid % NSTRIP, half = tid / NSTRIP, c = s / VPP, v = s % VPP, r0 = half * RPT;
            float h0[8], h1[8];
            uint4 cprev = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int q = 0; q < 8; ++q) { h0[q] = 0.f; h1[q] = 0.f; }
#pragma unroll
            for (int rr = 0; rr < RPT + 2; ++rr) {
                const char* rowp = zl + ((size_t)((r0 + rr) * ZW + c) * C + v * 8) * 2;
                const uint4 Lq = *reinterpret_cast<const uint4*>(rowp), Mq = *reinterpret_cast<const uint4*>(rowp + C * 2),
                            Rq = *reinterpret_cast<const uint4*>(rowp + 2 * C * 2);
                const unsigned lw[4] = {Lq.x, Lq.y, Lq.z, Lq.w}, mw[4] = {Mq.x, Mq.y, Mq.z, Mq.w}, rw[4] = {Rq.x, Rq.y, Rq.z, Rq.w};
                float h[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h[2 * q] = __uint_as_float(lw[q] << 16) + 2.f * __uint_as_float(mw[q] << 16) + __uint_as_float(rw[q] << 16);
                    h[2 * q + 1] = __uint_as_float(lw[q] & 0xffff0000u) + 2.f * __uint_as_float(mw[q] & 0xffff0000u) + __uint_as_float(rw[q] & 0xffff0000u);
                }
                if (rr >= 2) {
                    const int gy = ty0 + r0 + rr - 2, gx = tx0 + c;
                    if (gy < H && gx < W) {
                        unsigned ow[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            ow[q] = pack_bf16x2((h0[2 * q] + 2.f * h1[2 * q] + h[2 * q]) * 0.0625f, (h0[2 * q + 1] + 2.f * h1[2 * q + 1] + h[2 * q + 1]) * 0.0625f);
                        const size_t pix = ((size_t)b * H + gy) * W + gx;
                        *reinterpret_cast<uint4*>(y + pix * C + v * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                        if (bits) {
                            // sign bits of the pre-activation = of the stored activation (LeakyReLU keeps the sign): bit j = channel 8v + j
                            const unsigned cw[4] = {cprev.x, cprev.y, cprev.z, cprev.w};
                            unsigned bb = 0;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                bb |= ((short)(cw[q] & 0xffffu) > 0 ? 1u : 0u) << (2 * q);
                                bb |= ((short)(cw[q] >> 16) > 0 ? 1u : 0u) << (2 * q + 1);
                            }
                            bits[pix * VPP + v] = (unsigned char)bb;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) { h0[q] = h1[q]; h1[q] = h[q]; }
                cprev = Mq;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ row-streaming kernels
// The forward with the blur (the hot one: three launches per step) and the image gradient as ROW-STREAMING kernels: a wave owns a
// strip of 14 output columns (16 MFMA columns: one neighbour each side for the horizontal taps) and walks down a block of rows,
// keeping the three input rows a 3x3 window needs as MFMA B fragments in registers -- each new row is one small global load per lane
// (prefetched two rows ahead).  No LDS tile, no barrier: the horizontal [1,2,1] of the blur is a DPP row shift (an MFMA result row
// of 16 lanes = 16 consecutive pixels), the vertical one a two-row history in registers.  First version of these kernels (tile in
// LDS, activated tile back through LDS, three phases between barriers): 500-608 us at batch 32, 1024^2 against 276 us for the same
// convolution without the blur -- at three resident blocks per CU the phases did not overlap.
static __device__ __forceinline__ float dpp_row_shr1(float v) {     // lane l <- lane l - 1 inside each row of 16 lanes (0 into lane 0)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x111, 0xf, 0xf, true));
}
static __device__ __forceinline__ float dpp_row_shl1(float v) {     // lane l <- lane l + 1 (0 into lane 15)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x101, 0xf, 0xf, true));
}
enum { RC_FWD_DEFAULT = 0 };
enum { RC_STRIP = 14, RC_ROWS = 32, RC_PF = 3 };     // RC_PF: rows in flight per wave (8 measured slower: 110 registers, 4 waves per SIMD)

// The forward kernel is bound by its vector instructions, not by HBM (first version: ~180 VALU instructions per 448-byte output row of a
// wave = 0.67 ms of issue time at batch 32 against 0.3 ms of streaming time).  What the second version does about it:
//   * a wave whose strip, rows and halo lie inside the image (91 % of them at 1024^2) walks a loop without a single border test; the
//     others walk the same loop with per-row masks (the BORDER instantiation) -- a wave-uniform branch at the top, two loop bodies;
//   * the strip / row bookkeeping lives in scalar registers (the wave's item index through readfirstlane): row bases are scalar
//     pointers, the per-lane part a 32-bit offset; loads are unconditional from a clamped address (a conditional load merges with a
//     zero and the compiler waits for it on the spot: no prefetch) and a row block is a whole number of 6-row steps (stores predicated,
//     nothing else): rings and histories are indexed statically, nothing is moved between registers;
//   * conv0's bias is the MFMA's C operand; the blur's 1/16 is folded into the LeakyReLU's two slopes (exact: a power of two) and the
//     activation is max(x/16, 0.2 x/16) -- outside the image the two slopes are zero, which is the blur's zero padding;
//   * the [1,2,1] taps are two neighbour sums per direction ((a[l-1] + a[l]) + (a[l] + a[l+1])): two DPP adds per value horizontally
//     (this file is compiled without the SLP vectorizer so that the DPP operand folds into the add), vertically the pair sum of the
//     previous step is kept instead of two rows;
//   * a row's sign nibble is built when the row is computed (med3(bits, 0, 1) per channel) and carried one step as one register; the
//     two nibbles of a byte meet through v_permlane16_swap.
// The MFMA results stay in VGPRs (-mllvm -amdgpu-mfma-vgpr-form for this file: 6 waves per SIMD instead of 4, no accvgpr traffic).
// Workgroups go round-robin over the 8 XCDs, each with its own L2: with the natural order the four strips of a block and the four of
// the next block -- which share image columns (halo loads) and 128-byte lines of the output and of the sign bits (14-pixel strips are
// not line-aligned) -- sit on different XCDs, and their partial lines cannot merge in an L2.  Logical block = (XCD, index within the
// XCD): neighbours in memory run on one XCD, about at the same time.  (the grid is a multiple of 8 blocks; surplus items exit)
static __device__ __forceinline__ unsigned rc_xcd_block() { return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3); }
template <int CB, bool BORDER>
static __device__ __forceinline__ void fwdblur_walk(const float* __restrict__ ibase, const bf16_t* __restrict__ wf, const float* __restrict__ b0,
                                                    bf16_t* __restrict__ ybase, unsigned char* __restrict__ bbase, int H, int W, int ones, int sx,
                                                    int r_begin, int r_end, int nit, int dbg) {
    constexpr int C = 16 * CB;
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int zc = sx * RC_STRIP - 1 + l15;                // image column of this lane's convolution output
    const int pc = zc - 1 + (l4 < 3 ? l4 : 2);             // image column of this lane's B-operand pixel (kernel column l4; l4 = 3: the
    //                                                        weights' k 12..15 are zero, the lane re-reads kernel column 2's pixel)
    const bool pc_ok = (unsigned)pc < (unsigned)W;
    const int pca = pc < 0 ? 0 : (pc < W ? pc : W - 1);
    const unsigned ioff = (unsigned)pca * 3u;
    s16x4 wfr[CB][3];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) wfr[cb][ky] = *reinterpret_cast<const s16x4*>(wf + ((ky * C + cb * 16 + l15) * 16 + 4 * l4));
    f32x4_t biasv[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const float4 t = b0 ? *reinterpret_cast<const float4*>(b0 + cb * 16 + 4 * l4) : make_float4(0.f, 0.f, 0.f, 0.f);
        biasv[cb] = f32x4_t{t.x, t.y, t.z, t.w};
    }
    const unsigned lmask = (!BORDER || pc_ok) ? 0xffffffffu : 0u;       // BORDER: this lane's pixel column exists
    const unsigned one_hi = ones ? (0x3c000000u & lmask) : 0u;          // fp16 1.0 in the high half: the bias channel of an in-image pixel
    auto load_row = [&](int gy) -> rgb3 {                  // this lane's pixel of image row gy (row clamped into the image: masked at use)
        const int gyc = gy < 0 ? 0 : (gy < H ? gy : H - 1);
        return *reinterpret_cast<const rgb3*>(ibase + (size_t)gyc * W * 3 + ioff);
    };
    auto frag_of = [&](const rgb3& v, int gy) -> s16x4 {   // fp16 (r, g, b, 1 inside the image | 0)
        unsigned p01 = pack_f16x2(v.r, v.g);
        unsigned p23 = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v.b) | one_hi;
        if (BORDER) {
            const unsigned m = (unsigned)gy < (unsigned)H ? lmask : 0u;
            p01 &= m; p23 &= m;
        }
        s16x4 f;
        f[0] = (short)(p01 & 0xffffu); f[1] = (short)(p01 >> 16); f[2] = (short)(p23 & 0xffffu); f[3] = (short)(p23 >> 16);
        return f;
    };
    const int z0 = r_begin - 1;
    s16x4 fr[3];                                           // fragments of rows zrow - 1, zrow, zrow + 1 at slots s, s + 1, s + 2 (mod 3)
    fr[0] = frag_of(load_row(z0 - 1), z0 - 1); fr[1] = frag_of(load_row(z0), z0); fr[2] = fr[1];
    rgb3 ring[3];                                          // rows zrow + 1 .. zrow + 3 in flight
#pragma unroll
    for (int u = 0; u < 3; ++u) ring[u] = load_row(z0 + 1 + u);
    float h1[CB][4];                                       // horizontal sums of the previous row
    f32x2 vp[CB][2];                                       // pair sum (row - 2) + (row - 1)
    unsigned pbyte[CB];                                    // sign byte of the previous row (both nibbles, on every lane of the pair)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) h1[cb][i] = 0.f;
        vp[cb][0] = 0.f; vp[cb][1] = 0.f; pbyte[cb] = 0u;
    }
    const bool col_in = (unsigned)zc < (unsigned)W;
    const bool col_out = l15 >= 1 && l15 <= RC_STRIP && zc < W;       // this lane stores an output column (zc >= 0 there)
    const bool bit_out = col_out && !(l4 & 1) && bbase && !(dbg & 8);
    const bool y_out = col_out && !(dbg & 4);
    const unsigned yoff = col_out ? (unsigned)zc * C + 4u * l4 : 0u, boff = col_out ? (unsigned)zc * (C / 8) + (unsigned)(l4 >> 1) : 0u;
    const float sp_l = (!BORDER || col_in) ? 0.0625f : 0.f, sn_l = (!BORDER || col_in) ? SGX_LRELU * 0.0625f : 0.f;
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int zrow = z0 + 6 * it + u, s = u % 3;
            fr[(s + 2) % 3] = frag_of(ring[s], zrow + 1);
            ring[s] = load_row(zrow + 4);
            float sp = sp_l, sn = sn_l;
            if (BORDER) {                                              // rows outside the image: the blur's zero padding
                const bool row_in = (unsigned)zrow < (unsigned)H;
                sp = row_in ? sp_l : 0.f; sn = row_in ? sn_l : 0.f;
            }
            const f32x2 sp2 = {sp, sp}, sn2 = {sn, sn};
            const int orow = zrow - 1;                                   // centre row of the three horizontal sums now complete
            const bool store_row = orow >= r_begin && orow < r_end;      // (wave-uniform)
            bf16_t* yrow = ybase + (size_t)orow * W * C;
            unsigned char* brow = bbase + (size_t)orow * W * (C / 8);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f32x4_t acc = mma16h(wfr[cb][0], fr[s], biasv[cb]);
                acc = mma16h(wfr[cb][1], fr[(s + 1) % 3], acc);
                acc = mma16h(wfr[cb][2], fr[(s + 2) % 3], acc);
                float a[4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x2 x = {acc[2 * q], acc[2 * q + 1]};
                    const f32x2 p = x * sp2, n = x * sn2;            // lrelu(x) / 16 = max(x / 16, 0.2 x / 16)
                    a[2 * q] = fmaxf(p.x, n.x); a[2 * q + 1] = fmaxf(p.y, n.y);
                }
                unsigned nib = 0u;
#pragma unroll
                for (int i = 0; i < 4; ++i)                              // a > 0 <=> its bits, as a signed integer, are >= 1
                    nib |= (unsigned)min(max(__float_as_int(a[i]), 0), 1) << i;
                const auto sw = __builtin_amdgcn_permlane16_swap(nib, nib, false, false);       // [0]: the even row's nibble, [1]: the odd row's
                float h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = a[i] + dpp_row_shr1(a[i]);
                    h[i] = t + dpp_row_shl1(t);
                }
                const f32x2 vc0 = f32x2{h1[cb][0], h1[cb][1]} + f32x2{h[0], h[1]}, vc1 = f32x2{h1[cb][2], h1[cb][3]} + f32x2{h[2], h[3]};
                if (store_row) {
                    if (y_out) {
                        const f32x2 o0 = vp[cb][0] + vc0, o1 = vp[cb][1] + vc1;
                        *reinterpret_cast<uint2*>(yrow + yoff + cb * 16) = make_uint2(pack_bf16x2(o0.x, o0.y), pack_bf16x2(o1.x, o1.y));
                    }
                    if (bit_out) brow[boff + cb * 2] = (unsigned char)pbyte[cb];
                }
                vp[cb][0] = vc0; vp[cb][1] = vc1;
#pragma unroll
                for (int i = 0; i < 4; ++i) h1[cb][i] = h[i];
                pbyte[cb] = sw[0] | (sw[1] << 4);
            }
        }
    }
}
template <int CB>
__global__ __launch_bounds__(256) void rgbconv_fwdblur_kernel(const float* __restrict__ img, const bf16_t* __restrict__ wf, const float* __restrict__ b0,
                                                              bf16_t* __restrict__ y, unsigned char* __restrict__ bits, int B, int H, int W, int ones,
                                                              int nstrips, int nrb, int nit, int dbg) {
    // nit: 6-row steps per row block (a block stores 6 nit - 2 rows).  dbg (SGX_RGBCONV_DBG, profiling ablations -- wrong results by
    // design): 4 no output stores, 8 no sign bits
    constexpr int C = 16 * CB;
    int item = __builtin_amdgcn_readfirstlane((int)(rc_xcd_block() * 4 + (threadIdx.x >> 6)));
    const int sx = item % nstrips; item /= nstrips;
    const int rbk = item % nrb, b = item / nrb;
    if (b >= B) return;
    const int rb = 6 * nit - 2;
    const int r_begin = rbk * rb, r_end = r_begin + rb < H ? r_begin + rb : H;
    const float* ibase = img + (size_t)b * H * W * 3;
    bf16_t* ybase = y + (size_t)b * H * W * C;
    unsigned char* bbase = bits ? bits + (size_t)b * H * W * (C / 8) : nullptr;
    // every pixel the walk touches is an image pixel: columns sx * 14 - 2 .. sx * 14 + 15, rows r_begin - 2 .. r_begin + rb + 1
    const bool inside = sx * RC_STRIP - 2 >= 0 && sx * RC_STRIP + 15 < W && r_begin - 2 >= 0 && r_begin + rb + 1 < H;
    if (inside) fwdblur_walk<CB, false>(ibase, wf, b0, ybase, bbase, H, W, ones, sx, r_begin, r_end, nit, dbg);
    else fwdblur_walk<CB, true>(ibase, wf, b0, ybase, bbase, H, W, ones, sx, r_begin, r_end, nit, dbg);
}

template <int CB>
__global__ __launch_bounds__(256) void rgbconv_dgrad_kernel(const bf16_t* __restrict__ gz, const bf16_t* __restrict__ wd, float* __restrict__ gi,
                                                            int B, int H, int W, int nstrips, int nrb) {
    constexpr int C = 16 * CB;
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    int item = (int)rc_xcd_block() * 4 + (threadIdx.x >> 6);
    const int sx = item % nstrips; item /= nstrips;
    const int rbk = item % nrb, b = item / nrb;
    if (b >= B) return;
    const int r_begin = rbk * RC_ROWS, r_end = r_begin + RC_ROWS < H ? r_begin + RC_ROWS : H;
    const int pc = sx * RC_STRIP - 1 + l15;                // column of this lane's gz pixel AND of the output it assembles
    const bool pc_ok = (unsigned)pc < (unsigned)W;
    // A[i = kx' * 4 + j][k = channel] per kernel row ky': E[kx'][j][p] = sum_{ky', o} wd[ky'][kx' * 4 + j][o] gz[r + ky' - 1][p][o]
    s16x4 wfr[3][CB];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) wfr[ky][cb] = *reinterpret_cast<const s16x4*>(wd + ((ky * 16 + l15) * C + cb * 16 + 4 * l4));
    const bf16_t* gbase = gz + (size_t)b * H * W * C;
    struct Row { uint2 v[CB]; };
    auto load_row = [&](int gy) -> Row {                   // 4 channels per channel block of this lane's pixel in row gy (zeros outside)
        Row r;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            r.v[cb] = make_uint2(0u, 0u);
            if (pc_ok && (unsigned)gy < (unsigned)H) r.v[cb] = *reinterpret_cast<const uint2*>(gbase + ((size_t)gy * W + pc) * C + cb * 16 + 4 * l4);
        }
        return r;
    };
    auto frag_of = [&](const uint2& u) -> s16x4 {
        s16x4 f;
        f[0] = (short)(u.x & 0xffffu); f[1] = (short)(u.x >> 16); f[2] = (short)(u.y & 0xffffu); f[3] = (short)(u.y >> 16);
        return f;
    };
    Row g0 = load_row(r_begin - 1), g1 = load_row(r_begin);
    Row ring[RC_PF];                                       // rows r + 1 .. r + RC_PF in flight (see the forward kernel)
#pragma unroll
    for (int u = 0; u < RC_PF; ++u) ring[u] = load_row(r_begin + 1 + u);
    const bool col_out = l4 == 1 && l15 >= 1 && l15 <= RC_STRIP && pc < W;
    for (int rb = r_begin; rb < r_end; rb += RC_PF) {
#pragma unroll
        for (int u = 0; u < RC_PF; ++u) {
            const int r = rb + u;
            if (r < r_end) {                                             // (wave-uniform)
                const Row g2 = ring[u];
                ring[u] = load_row(r + 1 + RC_PF);
                f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    acc = mma16(wfr[0][cb], frag_of(g0.v[cb]), acc);
                    acc = mma16(wfr[1][cb], frag_of(g1.v[cb]), acc);
                    acc = mma16(wfr[2][cb], frag_of(g2.v[cb]), acc);
                }
                // this lane (kernel column l4, pixel l15) holds E[l4][j = reg]; the output pixel q sums E[0][q - 1] + E[1][q] + E[2][q + 1]
                float o3[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) o3[j] = __shfl(acc[j], l15 - 1, 64) + acc[j] + __shfl(acc[j], 32 + l15 + 1, 64);
                if (col_out) *reinterpret_cast<rgb3*>(gi + (((size_t)b * H + r) * W + pc) * 3) = rgb3{o3[0], o3[1], o3[2]};
                g0 = g1; g1 = g2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ weight gradient
template <int CB> struct RcWg {
    static constexpr int C = 16 * CB, TH = 8, TW = 64, TP = TH * TW, RH = TH + 2;
    // plane strides == 16 bytes mod 256: the 16 rows (channels / (kx, j) planes) a fragment read touches fall on 16 different
    // 16-byte bank slots (ds_read_b64: 32 lanes per LDS cycle over 64 banks -- unpadded planes were a 16-way conflict)
    static constexpr int TPS = TP + 8, IPS = RH * TW + 8;
    static constexpr int GZT = C * TPS * 2;                   // gzT[C][TPS] bf16
    static constexpr int IMT = 12 * IPS * 2;                  // imgT[3 kx][4 j][IPS] bf16, row rr at rr * TW
    static constexpr int RED = 4 * CB * 3 * 256 * 4;          // cross-wave reduction of the accumulators (reuses the stage area)
    static constexpr int LDS = (GZT + IMT > RED ? GZT + IMT : RED);
    static constexpr int NOUT = C * 48;                       // dW'[o][ky][16 n], n = kx * 4 + j (12..15 unused)
};

template <int CB>
__global__ __launch_bounds__(256) void rgbconv_wgrad_kernel(const float* __restrict__ img, const bf16_t* __restrict__ gz, float* __restrict__ part,
                                                            int B, int H, int W, int ones, int tiles_x, int tiles_y, int ntiles) {
    using G = RcWg<CB>;
    constexpr int C = G::C, TH = G::TH, TW = G::TW, TP = G::TP, TPS = G::TPS, IPS = G::IPS, RH = G::RH, VPP = C / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* gzT = reinterpret_cast<bf16_t*>(smem);
    bf16_t* imgT = reinterpret_cast<bf16_t*>(smem + G::GZT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    f32x4_t acc[CB][3];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc[cb][ky] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // B operand of this lane: column n = lane & 15 -> (kx, j); columns 12..15 are padding (any valid address: their results are dropped)
    const int nkx = (l15 >> 2) < 3 ? (l15 >> 2) : 0, nj = l15 & 3;
    const RcSched sc = rc_sched(ntiles);
    constexpr int NGV = TP * VPP / 256;
    constexpr int NIP = RH * (TW + 2), NII = (NIP + 255) / 256;
    uint4 gv[NGV];
    rgb3 iv[NII];
    unsigned okm = 0;
    auto tile_at = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        b = t / tiles_y; ty0 = ty * TH; tx0 = tx * TW;
    };
    auto load_tile = [&](int t) {
        int b, ty0, tx0;
        tile_at(t, b, ty0, tx0);
#pragma unroll
        for (int it = 0; it < NGV; ++it) {
            const int idx = it * 256 + tid, p = idx / VPP, v = idx - p * VPP, r = p / TW, c = p - r * TW;
            const int gy = ty0 + r, gx = tx0 + c;
            gv[it] = (gy < H && gx < W) ? *reinterpret_cast<const uint4*>(gz + (((size_t)b * H + gy) * W + gx) * C + v * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
        okm = 0;
#pragma unroll
        for (int it = 0; it < NII; ++it) {
            const int idx = it * 256 + tid, rr = idx / (TW + 2), cr = idx - rr * (TW + 2);
            const int gy = ty0 - 1 + rr, gx = tx0 - 1 + cr;
            const bool ok = idx < NIP && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            iv[it] = rgb3{0.f, 0.f, 0.f};
            if (ok) {
                iv[it] = *reinterpret_cast<const rgb3*>(img + (((size_t)b * H + gy) * W + gx) * 3);
                okm |= 1u << it;
            }
        }
    };
    if (sc.first < sc.end) load_tile(sc.first);
    for (int t = sc.first; t < sc.end; t += sc.stride) {
        // gz tile -> planar gzT[channel][pixel]
#pragma unroll
        for (int it = 0; it < NGV; ++it) {
            const int idx = it * 256 + tid, p = idx / VPP, v = idx - p * VPP;
            const unsigned w[4] = {gv[it].x, gv[it].y, gv[it].z, gv[it].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gzT[(v * 8 + 2 * q) * TPS + p] = (bf16_t)(w[q] & 0xffffu);
                gzT[(v * 8 + 2 * q + 1) * TPS + p] = (bf16_t)(w[q] >> 16);
            }
        }
        // image region (one halo pixel) as bf16 (r, g, b, 1 | 0), one pre-shifted copy per kernel column
#pragma unroll
        for (int it = 0; it < NII; ++it) {
            const int idx = it * 256 + tid, rr = idx / (TW + 2), cr = idx - rr * (TW + 2);
            if (idx < NIP) {
                const unsigned p01 = pack_bf16x2(iv[it].r, iv[it].g), p23 = pack_bf16x2(iv[it].b, (((okm >> it) & 1u) && ones) ? 1.f : 0.f);
                const bf16_t vj[4] = {(bf16_t)(p01 & 0xffffu), (bf16_t)(p01 >> 16), (bf16_t)(p23 & 0xffffu), (bf16_t)(p23 >> 16)};
                // imgT[kx][j][rr][c] = image(row ty0 - 1 + rr, column tx0 + c + kx - 1): region column cr = c + kx
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int c = cr - kx;
                    if (c >= 0 && c < TW) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) imgT[(kx * 4 + j) * IPS + rr * TW + c] = vj[j];
                    }
                }
            }
        }
        __syncthreads();
        if (t + sc.stride < sc.end) load_tile(t + sc.stride);
        for (int g = wave; g < TH * 4; g += 4) {
            const int r = g >> 2, c0 = (g & 3) * 16 + 4 * l4;
            s16x4 af[CB];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) af[cb] = *reinterpret_cast<const s16x4*>(gzT + (cb * 16 + l15) * TPS + r * TW + c0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const s16x4 bf = *reinterpret_cast<const s16x4*>(imgT + (nkx * 4 + nj) * IPS + (r + ky) * TW + c0);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[cb][ky] = mma16(af[cb], bf, acc[cb][ky]);
            }
        }
        __syncthreads();                                   // the fragment reads are done: the next tile may be staged
    }
    // ---- the four waves' accumulators -> one partial per block: part[block][(o * 3 + ky) * 16 + n]
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) red[((wave * CB + cb) * 3 + ky) * 256 + rg * 64 + lane] = acc[cb][ky][rg];
    __syncthreads();
    for (int e = tid; e < CB * 3 * 256; e += 256) {
        const int cbky = e >> 8, rem = e & 255, rg = rem >> 6, ln = rem & 63, cb = cbky / 3, ky = cbky - cb * 3;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * CB * 3 + cbky) * 256 + rem];
        const int o = cb * 16 + 4 * (ln >> 4) + rg, n = ln & 15;
        part[(size_t)blockIdx.x * G::NOUT + (o * 3 + ky) * 16 + n] = s;
    }
}

// sum of the block partials, deterministic: block = 8 outputs x 32 slices of the partial list
__global__ __launch_bounds__(256) void rgbconv_wfinish_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int nout) {
    __shared__ float red[256];
    const int tid = threadIdx.x, el = tid & 7, sl = tid >> 3, e = blockIdx.x * 8 + el;
    float s = 0.f;
    if (e < nout)
        for (int k = sl; k < nblk; k += 32) s += part[(size_t)k * nout + e];
    red[tid] = s;
    __syncthreads();
    if (tid < 8 && e < nout) {
        float tsum = 0.f;
        for (int k = 0; k < 32; ++k) tsum += red[k * 8 + tid];
        out[e] = tsum;
    }
}

// chain rule through W' = s0 sr W0 (x) Wr, T = s0 W0 br:   dwp[(o * 3 + ky) * 16 + kx * 4 + j]
//   dW0[o][i][ky][kx] = s0 sr sum_j dW'[o][j][ky][kx] Wr[i][j] + s0 dW'[o][3][ky][kx] br[i] bscale
//   dWr[i][j] = s0 sr sum_{o,ky,kx} dW'[o][j][ky][kx] W0[o][i][ky][kx];   dbr[i] = s0 bscale sum_{o,ky,kx} dW'[o][3][ky][kx] W0[o][i][ky][kx]
//   db0[o] = dW'[o][3][1][1]  (the bias channel under the centre tap counts every pixel of the image once)
// acc bit 0: dW0, 1: db0, 2: dWr, 3: dbr accumulate into the existing gradient instead of overwriting it.  NULL outputs are skipped.
__global__ __launch_bounds__(256) void rgbconv_chain_kernel(const float* __restrict__ dwp, const float* __restrict__ w0, float s0, const float* __restrict__ wr,
                                                            float sr, const float* __restrict__ br, float bscale, float* __restrict__ dw0, float* __restrict__ db0,
                                                            float* __restrict__ dwr, float* __restrict__ dbr, int acc, int C) {
    const int tid = threadIdx.x;
    if (dw0) {
        for (int e = tid; e < C * C * 9; e += 256) {
            const int o = e / (C * 9), i = (e / 9) % C, tap = e % 9, ky = tap / 3, kx = tap % 3;
            const float* d = dwp + (o * 3 + ky) * 16 + kx * 4;
            float v = s0 * sr * (d[0] * wr[i * 3] + d[1] * wr[i * 3 + 1] + d[2] * wr[i * 3 + 2]);
            if (br) v += s0 * d[3] * br[i] * bscale;
            dw0[e] = (acc & 1) ? dw0[e] + v : v;
        }
    }
    if (db0)
        for (int o = tid; o < C; o += 256) {
            const float v = dwp[(o * 3 + 1) * 16 + 4 + 3];
            db0[o] = (acc & 2) ? db0[o] + v : v;
        }
    for (int e = tid; e < C * 4; e += 256) {
        const int i = e >> 2, j = e & 3;
        float* dst = j < 3 ? (dwr ? dwr + i * 3 + j : nullptr) : (dbr ? dbr + i : nullptr);
        if (!dst) continue;
        float v = 0.f;
        for (int o = 0; o < C; ++o)
            for (int tap = 0; tap < 9; ++tap) v += dwp[(o * 3 + tap / 3) * 16 + (tap % 3) * 4 + j] * w0[(o * C + i) * 9 + tap];
        v *= j < 3 ? s0 * sr : s0 * bscale;
        const bool a = j < 3 ? (acc & 4) : (acc & 8);
        *dst = a ? *dst + v : v;
    }
}

// ------------------------------------------------------------------------------------------------------------ host side
static bool rgbconv_shape_ok(int B, int H, int W, int C, int dtype) {
    static const int on = [] { const char* e = getenv("SGX_RGBCONV"); return e ? atoi(e) : 1; }();     // A/B switch
    return on && dtype == SGX_BF16 && (C == 16 || C == 32) && B >= 1 && H >= 16 && W >= 64 && H % 16 == 0 && W % 64 == 0;
}
extern "C" int sgx_rgbconv_ok(int B, int H, int W, int C, int dtype) { return rgbconv_shape_ok(B, H, W, C, dtype) ? 1 : 0; }

extern "C" int sgx_rgbconv_pack(const float* w0, float s0, const float* wr, float sr, const float* br, float bscale, void* wf, void* wd, int C,
                                void* stream) {
    SGX_REQUIRE(w0 && wr && wf && wd, SGX_EINVAL, "rgbconv_pack: null argument");
    SGX_REQUIRE(C == 16 || C == 32, SGX_EUNSUPPORTED, "rgbconv_pack: %d channels (16 or 32)", C);
    SGX_NOTE(0.0, 0.0, "rgbconv_pack C%d", C);
    hipLaunchKernelGGL(rgbconv_pack_kernel, dim3(C == 16 ? 12 : 24), dim3(256), 0, (hipStream_t)stream, w0, s0, wr, sr, br, bscale,
                       static_cast<bf16_t*>(wf), static_cast<bf16_t*>(wd), C);
    SGX_LAUNCH_CHECK("rgbconv_pack_kernel");
    return 0;
}

// persistent grid: as many blocks as fit the chip at once (by LDS; at most 4 per CU), a multiple of 8 (one share per XCD), not
// more than there are tiles
static int rc_ncu() { return sgx_ncu(); }
template <typename K>
static int rc_per_cu(K kern, int lds_bytes) {              // resident 256-thread blocks per CU (registers AND LDS), at most 4
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, (size_t)lds_bytes) != hipSuccess || n < 1) n = 1;
    return n > 4 ? 4 : n;
}
static int rc_grid(int ntiles, int per_cu) {
    long g = (long)rc_ncu() * per_cu;
    if (g > ntiles) g = ntiles;
    g = (g + 7) / 8 * 8;
    return (int)g;
}

template <int CB, int EPI>
static int launch_rgbconv_fwd(const float* img, const bf16_t* wf, const float* b0, bf16_t* y, unsigned char* bits, int B, int H, int W, int ones,
                              hipStream_t st, bool one_shot = false) {
    using G = RcFwd<CB, EPI>;
    const int tiles_x = W / G::TW, tiles_y = H / G::TH, ntiles = B * tiles_x * tiles_y;
    sgx_lds_opt_in<rgbconv_fwd_kernel<CB, EPI>>(G::LDS);
    static const int per_cu = rc_per_cu(rgbconv_fwd_kernel<CB, EPI>, G::LDS);
    hipLaunchKernelGGL((rgbconv_fwd_kernel<CB, EPI>), dim3((unsigned)(one_shot ? (ntiles + 7) / 8 * 8 : rc_grid(ntiles, per_cu))), dim3(256), G::LDS, st, img, wf, b0, y, bits, B, H, W, ones,
                       tiles_x, tiles_y, ntiles);
    SGX_LAUNCH_CHECK("rgbconv_fwd_kernel");
    return 0;
}
// A/B and probe state of the forward kernel.  Read from the environment ONCE per process (SGX_RGBCONV_FWD: 0 row-streaming kernel,
// 1 LDS-tile kernel with persistent blocks, 2 LDS-tile kernel with one tile per block; SGX_RGBCONV_NIT: 6-row steps per row block,
// 1..8), changed afterwards only through sgx_rgbconv_tune (tests, tools/rgbconv_probe.py): no getenv on the launch path, and the
// profiling ablations (`dbg`: 4 no output stores, 8 no sign bits -- WRONG results by design) cannot be switched on by a stray
// environment variable.
struct RcTune { int fwd, nit, dbg; };
static RcTune& rc_tune() {
    static RcTune t = [] {
        RcTune r{RC_FWD_DEFAULT, 0, 0};
        const char* ve = getenv("SGX_RGBCONV_FWD");
        if (ve && atoi(ve) >= 0 && atoi(ve) <= 2) r.fwd = atoi(ve);
        const char* ne = getenv("SGX_RGBCONV_NIT");
        if (ne && atoi(ne) >= 1 && atoi(ne) <= 8) r.nit = atoi(ne);
        return r;
    }();
    return t;
}
extern "C" int sgx_rgbconv_tune(int fwd_variant, int nit, int dbg) {
    SGX_REQUIRE(fwd_variant >= -1 && fwd_variant <= 2, SGX_EINVAL, "rgbconv_tune: forward variant %d (0..2, -1 = the default)", fwd_variant);
    SGX_REQUIRE(nit >= 0 && nit <= 8, SGX_EINVAL, "rgbconv_tune: %d six-row steps per row block (1..8, 0 = by launch size)", nit);
    SGX_REQUIRE(dbg == 0 || dbg == 4 || dbg == 8 || dbg == 12, SGX_EINVAL, "rgbconv_tune: ablation mask %d", dbg);
    RcTune& t = rc_tune();
    t.fwd = fwd_variant < 0 ? RC_FWD_DEFAULT : fwd_variant; t.nit = nit; t.dbg = dbg;
    return 0;
}

extern "C" int sgx_rgbconv_fwd(const float* img, const void* wf, const float* b0, void* y, void* bits, int B, int H, int W, int C, int epi, int ones,
                               int dtype, void* stream) {
    SGX_REQUIRE(img && wf && y, SGX_EINVAL, "rgbconv_fwd: null argument");
    SGX_REQUIRE(rgbconv_shape_ok(B, H, W, C, dtype), SGX_EUNSUPPORTED, "rgbconv_fwd: shape B%d %dx%d C%d dtype %d (sgx_rgbconv_ok == 0)", B, H, W, C, dtype);
    SGX_REQUIRE(epi == 0 || epi == 1, SGX_EINVAL, "rgbconv_fwd: epilogue %d", epi);
    SGX_REQUIRE(epi == 1 || !ones, SGX_EINVAL, "rgbconv_fwd: the plain convolution prescales its input (a gradient) by a power of two; it has no bias channel");
    const double px = (double)B * H * W;
    SGX_NOTE(2.0 * 27 * C * px, px * (12.0 + 2.0 * C + (epi && bits ? C / 8.0 : 0.0)), "rgbconv%s B%d %dx%d 3->%d", epi ? "+act+blur" : "", B, H, W, C);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* w = static_cast<const bf16_t*>(wf);
    bf16_t* out = static_cast<bf16_t*>(y);
    unsigned char* bt = static_cast<unsigned char*>(bits);
    if (epi) {
        // A/B (read per launch: tools/rgbconv_probe.py): 0 = row-streaming kernel, 1 = LDS-tile kernel with persistent blocks, 2 = LDS-tile
        // kernel with one tile per block
        const int variant = rc_tune().fwd;
        if (variant) return C == 16 ? launch_rgbconv_fwd<1, 1>(img, w, b0, out, bt, B, H, W, ones, st, variant == 2)
                                    : launch_rgbconv_fwd<2, 1>(img, w, b0, out, bt, B, H, W, ones, st, variant == 2);
        // rows per block 6 nit - 2 (two halo rows per block are recomputed): 34 where that still leaves > 3 waves per SIMD slot of the
        // chip, 22 / 16 for small launches (batch 4 at 1024^2: 14k / 19k waves)          SGX_RGBCONV_NIT overrides (probe)
        const int nstrips = (W + RC_STRIP - 1) / RC_STRIP;
        const int nit_env = rc_tune().nit;                  // 1..8 overrides the row-block heuristic (sgx_rgbconv_tune)
        int nit = nit_env ? nit_env : 6;
        if (!nit_env) {
            while (nit > 3 && (long)B * nstrips * ((H + 6 * nit - 3) / (6 * nit - 2)) < 3L * 256 * 24) --nit;
        }
        const int rb = 6 * nit - 2, nrb = (H + rb - 1) / rb;
        const unsigned grid = (unsigned)((((long)B * nstrips * nrb + 3) / 4 + 7) / 8 * 8);      // (rc_xcd_block)
        const int dbg = rc_tune().dbg;                      // profiling ablations: only through sgx_rgbconv_tune, never from the environment
        if (C == 16) hipLaunchKernelGGL((rgbconv_fwdblur_kernel<1>), dim3(grid), dim3(256), 0, st, img, w, b0, out, bt, B, H, W, ones, nstrips, nrb, nit, dbg);
        else hipLaunchKernelGGL((rgbconv_fwdblur_kernel<2>), dim3(grid), dim3(256), 0, st, img, w, b0, out, bt, B, H, W, ones, nstrips, nrb, nit, dbg);
        SGX_LAUNCH_CHECK("rgbconv_fwdblur_kernel");
        return 0;
    }
    return C == 16 ? launch_rgbconv_fwd<1, 0>(img, w, b0, out, bt, B, H, W, ones, st) : launch_rgbconv_fwd<2, 0>(img, w, b0, out, bt, B, H, W, ones, st);
}

extern "C" int sgx_rgbconv_dgrad(const void* gz, const void* wd, float* gi, int B, int H, int W, int C, int dtype, void* stream) {
    SGX_REQUIRE(gz && wd && gi, SGX_EINVAL, "rgbconv_dgrad: null argument");
    SGX_REQUIRE(rgbconv_shape_ok(B, H, W, C, dtype), SGX_EUNSUPPORTED, "rgbconv_dgrad: shape B%d %dx%d C%d dtype %d (sgx_rgbconv_ok == 0)", B, H, W, C, dtype);
    const double px = (double)B * H * W;
    SGX_NOTE(2.0 * 27 * C * px, px * (12.0 + 2.0 * C), "rgbconv_dgrad B%d %dx%d %d->3", B, H, W, C);
    hipStream_t st = (hipStream_t)stream;
    const int nstrips = (W + RC_STRIP - 1) / RC_STRIP, nrb = (H + RC_ROWS - 1) / RC_ROWS;
    const unsigned grid = (unsigned)((((long)B * nstrips * nrb + 3) / 4 + 7) / 8 * 8);          // (rc_xcd_block)
    if (C == 16) hipLaunchKernelGGL((rgbconv_dgrad_kernel<1>), dim3(grid), dim3(256), 0, st, static_cast<const bf16_t*>(gz), static_cast<const bf16_t*>(wd), gi, B, H, W, nstrips, nrb);
    else hipLaunchKernelGGL((rgbconv_dgrad_kernel<2>), dim3(grid), dim3(256), 0, st, static_cast<const bf16_t*>(gz), static_cast<const bf16_t*>(wd), gi, B, H, W, nstrips, nrb);
    SGX_LAUNCH_CHECK("rgbconv_dgrad_kernel");
    return 0;
}

static int rgbconv_wgrad_blocks(int B, int H, int W) {
    const long ntiles = (long)B * (H / 8) * (W / 64);
    return rc_grid((int)(ntiles < (1 << 30) ? ntiles : (1 << 30)), 3);     // 3 resident blocks per CU for both channel counts (registers / LDS)
}
extern "C" size_t sgx_rgbconv_wgrad_ws_bytes(int B, int H, int W, int C) {
    if (B < 1 || H < 8 || W < 64 || (C != 16 && C != 32)) return 0;
    return ((size_t)rgbconv_wgrad_blocks(B, H, W) + 1) * C * 48 * sizeof(float);
}
extern "C" int sgx_rgbconv_wgrad(const float* img, const void* gz, int ones, const float* w0, float s0, const float* wr, float sr, const float* br,
                                 float bscale, float* dw0, float* db0, float* dwr, float* dbr, int acc, void* ws, size_t ws_bytes, int B, int H, int W,
                                 int C, int dtype, void* stream) {
    SGX_REQUIRE(img && gz && w0 && wr && ws, SGX_EINVAL, "rgbconv_wgrad: null argument");
    SGX_REQUIRE(rgbconv_shape_ok(B, H, W, C, dtype), SGX_EUNSUPPORTED, "rgbconv_wgrad: shape B%d %dx%d C%d dtype %d (sgx_rgbconv_ok == 0)", B, H, W, C, dtype);
    SGX_REQUIRE(ws_bytes >= sgx_rgbconv_wgrad_ws_bytes(B, H, W, C), SGX_EWORKSPACE, "rgbconv_wgrad: workspace %zu < %zu", ws_bytes,
                sgx_rgbconv_wgrad_ws_bytes(B, H, W, C));
    SGX_REQUIRE(ones || (!db0 && !dbr), SGX_EINVAL, "rgbconv_wgrad: bias gradients need the bias channel (ones = 1)");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = rgbconv_wgrad_blocks(B, H, W), nout = C * 48;
    const int tiles_x = W / 64, tiles_y = H / 8, ntiles = B * tiles_x * tiles_y;
    float* part = static_cast<float*>(ws);
    float* dwp = part + (size_t)nblk * nout;
    const double px = (double)B * H * W;
    SGX_NOTE(2.0 * 36 * C * px, px * (12.0 + 2.0 * C), "rgbconv_wgrad B%d %dx%d 3x%d", B, H, W, C);
    if (C == 16) {
        sgx_lds_opt_in<rgbconv_wgrad_kernel<1>>(RcWg<1>::LDS);
        hipLaunchKernelGGL((rgbconv_wgrad_kernel<1>), dim3((unsigned)nblk), dim3(256), RcWg<1>::LDS, st, img, static_cast<const bf16_t*>(gz), part, B, H, W,
                           ones, tiles_x, tiles_y, ntiles);
    } else {
        sgx_lds_opt_in<rgbconv_wgrad_kernel<2>>(RcWg<2>::LDS);
        hipLaunchKernelGGL((rgbconv_wgrad_kernel<2>), dim3((unsigned)nblk), dim3(256), RcWg<2>::LDS, st, img, static_cast<const bf16_t*>(gz), part, B, H, W,
                           ones, tiles_x, tiles_y, ntiles);
    }
    SGX_LAUNCH_CHECK("rgbconv_wgrad_kernel");
    hipLaunchKernelGGL(rgbconv_wfinish_kernel, dim3((unsigned)((nout + 7) / 8)), dim3(256), 0, st, (const float*)part, dwp, nblk, nout);
    SGX_LAUNCH_CHECK("rgbconv_wfinish_kernel");
    hipLaunchKernelGGL(rgbconv_chain_kernel, dim3(1), dim3(256), 0, st, (const float*)dwp, w0, s0, wr, sr, br, bscale, dw0, db0, dwr, dbr, acc, C);
    SGX_LAUNCH_CHECK("rgbconv_chain_kernel");
    return 0;
}
