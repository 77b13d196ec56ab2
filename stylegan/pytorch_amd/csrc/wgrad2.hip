// Second-generation bf16 weight-gradient kernels for gfx950: v_mfma_f32_32x32x16_bf16 over pixel-major (NHWC) operand
// tiles that are staged by LDS-DMA (global_load_lds_dwordx4) into two LDS stages and read K-major with
// ds_read_b64_tr_b16 -- one pass over both activations per 64 x 64 channel tile with ALL taps accumulated in registers.
//
// Why (round-2 counters, profiles/r02_b_pmc_mfma_bf16_b4.json + r02_pmc_traffic_bf16_b4.json): the first-generation kernel
// (conv.hip, wgrad_kernel) owns a 32 x 16/32 channel pair per block, so a layer with C channels re-reads each activation
// C/16..C/32 times (249 MB of HBM traffic per launch against 66 MB algorithmic), its transpose reads run at 40-50 % bank
// conflicts (80-byte pixel pitch) and its MFMA pipe is 9-19 % busy.  Here a block is 8 waves on a 64 (dy side, "n") x 64 (x
// side, "k") channel tile; the operands of a pixel tile land in LDS once and every tap reuses them.
//
// GEMM view: dW[tap][n][k] = sum over pixels p of nside[p][n] * kside[p + tap][k]:  M = n channels (A operand), N = k
// channels (B operand), reduction K = 16 consecutive pixels of one image row per MFMA.
//
// LDS image of a stage: per operand and 32-channel plane an array of pixels, 64 bytes (32 channels) each, pixels of a tile /
// patch row contiguous.  A 32x32x16 operand fragment = 16 consecutive pixels x 32 channels: lane l of the wave belongs to
// the 16-lane group g = l / 16 with channel half cb = g & 1 and pixel half kg = g / 2, and issues two transpose reads
// (pixels +0..3 and +4..7 of its half); inside a read the 32 lanes of a half-wave touch 4 pixels x 64 B = 256 contiguous
// bytes: conflict-free for any starting pixel (so a tap's column shift costs nothing).
//
// Two geometries:
//   W2_S : 3x3 stride 1.  Tile 8 rows x 32 pixels; k-side patch 10 x 34 (halo 1).  Waves = 4 channel pairs (n-block, k-block
//          of 32) x 2 column segments of 16 pixels; a wave walks the patch rows once, keeps the last three n-side fragments
//          and feeds 9 accumulators (row R of the patch meets n-side rows R, R-1, R-2).  The two segment waves of a pair are
//          summed through LDS at the end.
//   W2_D : 4x4 stride 2 (fine k side, coarse n side) as four polyphase 2x2 problems.  Tile 4 x 16 coarse pixels; per phase a
//          5 x 17 patch of the phase's sub-image.  Waves = 4 phases x 2 n-blocks, each with 4 taps x KBW k-blocks of
//          accumulators; no cross-wave reduction.  tap (ky,kx) = (2e+1-py, 2f+1-px) for patch offsets e,f in {0,1}.
// Output: the per-split partials of wgrad_finish_kernel (conv.hip), out[split][tap][n][k] (+ the n side's column sums = the
// bias gradient), deterministic (no atomics).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __attribute__((aligned(64))) const unsigned wg2_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ void wg2_glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ s16x4 wg2_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}
// 16 pixels x 32 channels starting at `p` (the lane's own address inside the fragment, see file header): pixels +0..3 of the
// lane's half from the first read, +4..7 from the second (4 pixels = 256 bytes further)
__device__ __forceinline__ bf16x8 wg2_frag(const char* p) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x4 lo = wg2_tr16(p), hi = wg2_tr16(p + 256);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 wg2_ones() {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
    return __builtin_bit_cast(bf16x8, v);
}

struct Wg2Args {
    const bf16_t* kside; const bf16_t* nside; float* out;
    int B, Hk, Wk, Hn, Wn, Ck, Cn;            // k side: x (S) / the fine tensor (D);  n side: dy (S) / the coarse tensor (D)
    int tiles_x, tiles_y, ntiles, nsplit;     // tiles of the n-side grid; pixel splits (a multiple of 8)
    int nct_n, nct_k;                         // channel tiles: Cn / 64, Ck / (32 * KBW)
    int want_bias;
    long split_stride;                        // floats per split: NT * Cn * Ck + Cn
};

// store one 32 x 32 accumulator block: rows n (8g + 4hi + j), column k = l31
__device__ __forceinline__ void wg2_store_block(float* out_t, const f32x16& v, int Ck, int nrow0, int kcol, int hi) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) out_t[(size_t)(nrow0 + 8 * g + 4 * hi + j) * Ck + kcol] = v[4 * g + j];
}

// ------------------------------------------------------------------------------------------------------------- W2_S
template <int TH>
__global__ __launch_bounds__(512, 2) void wgrad2_s_kernel(Wg2Args a) {
    constexpr int TW = 32, PH = TH + 2, PW = TW + 2;
    constexpr int A_PLANE = TH * TW * 64, A_INSTR = A_PLANE / 1024;
    constexpr int B_PIX = PH * PW, B_INSTR = (B_PIX * 64 + 1023) / 1024, B_PLANE = B_INSTR * 1024;
    constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE, NINSTR = 2 * A_INSTR + 2 * B_INSTR, NPI = (NINSTR + 7) / 8;
    static_assert(A_PLANE % 1024 == 0, "whole DMA instructions per plane");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blocks of one XCD (id % 8) that are neighbours in id / 8 share a pixel split and differ in the channel tile: the tiles
    // they both read are served by that XCD's L2
    const int bid = blockIdx.x, xcd = bid & 7, j8 = bid >> 3;
    const int nct = a.nct_n * a.nct_k;
    const int ct = j8 % nct, split = (j8 / nct) * 8 + xcd;
    if (split >= a.nsplit) return;
    const int n0 = (ct / a.nct_k) * 64, k0 = (ct % a.nct_k) * 64;
    const int nsteps = (a.ntiles - split + a.nsplit - 1) / a.nsplit;

    // ---- per-lane DMA descriptors (tile independent).  Instruction ii of a stage fills LDS bytes [ii*1024, +1024):
    // 16 pixels x 64 bytes; lane -> (pixel ii*16 + lane/4 of its plane, 16-byte chunk lane%4)
    int rel[NPI], pos[NPI];
#pragma unroll
    for (int jj = 0; jj < NPI; ++jj) {
        const int ii = jj * 8 + wave, c16 = (lane & 3) * 8;
        rel[jj] = 0; pos[jj] = -1;
        if (ii < 2 * A_INSTR) {
            const int plane = ii / A_INSTR, pix = (ii % A_INSTR) * 16 + (lane >> 2), r = pix / TW, c = pix % TW;
            rel[jj] = (r * a.Wn + c) * a.Cn + plane * 32 + c16;
            pos[jj] = (r << 8) | c;
        } else if (ii < NINSTR) {
            const int j = ii - 2 * A_INSTR, plane = j / B_INSTR, pix = (j % B_INSTR) * 16 + (lane >> 2);
            if (pix < B_PIX) {
                const int pr = pix / PW, pc = pix % PW;
                rel[jj] = (pr * a.Wk + pc) * a.Ck + plane * 32 + c16;
                pos[jj] = (pr << 8) | pc;
            }
        }
    }
    const unsigned long long zaddr = reinterpret_cast<unsigned long long>(wg2_zero_page) + (lane & 3) * 16;
    auto tile_coords = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx_i = t % a.tiles_x; t /= a.tiles_x;
        const int ty_i = t % a.tiles_y;
        b = t / a.tiles_y; ty0 = ty_i * TH; tx0 = tx_i * TW;
    };
    auto issue = [&](int step, char* buf) {
        int b, ty0, tx0;
        tile_coords(split + step * a.nsplit, b, ty0, tx0);
        const bf16_t* abase = a.nside + (((long)b * a.Hn + ty0) * a.Wn + tx0) * a.Cn + n0;
        const bf16_t* bbase = a.kside + (((long)b * a.Hk + ty0 - 1) * a.Wk + tx0 - 1) * a.Ck + k0;
#pragma unroll
        for (int jj = 0; jj < NPI; ++jj) {
            const int ii = jj * 8 + wave;
            if (ii < NINSTR) {
                const bool isb = ii >= 2 * A_INSTR;                           // wave uniform
                const int pp = pos[jj];
                const int gy = (isb ? ty0 - 1 : ty0) + (pp >> 8), gx = (isb ? tx0 - 1 : tx0) + (pp & 255);
                const unsigned long long ok = ((pp >= 0) & ((unsigned)gy < (unsigned)a.Hn) & ((unsigned)gx < (unsigned)a.Wn)) ? ~0ull : 0ull;
                const unsigned long long pa = reinterpret_cast<unsigned long long>((isb ? bbase : abase) + rel[jj]);
                wg2_glds16(reinterpret_cast<const void*>(zaddr + ((pa - zaddr) & ok)), buf + ii * 1024);
            }
        }
    };

    // ---- this wave's part: channel pair (n-block, k-block), column segment
    const int pair = wave & 3, nb = pair >> 1, kb = pair & 1, seg = wave >> 2;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int lane_off = (((g16 >> 1) * 8 + (i16 >> 2)) * 64) + (g16 & 1) * 32 + (i16 & 3) * 8;
    const int a_off = nb * A_PLANE + seg * 16 * 64 + lane_off;
    const int b_off = 2 * A_PLANE + kb * B_PLANE + seg * 16 * 64 + lane_off;
    const bool do_bias = a.want_bias && k0 == 0 && kb == 0;

    f32x16 acc[9], bacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bacc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t][r] = 0.f;
    }
    const bf16x8 ones = wg2_ones();

    if (nsteps > 0) issue(0, smem);
    for (int step = 0; step < nsteps; ++step) {
        const char* cur = smem + (step & 1) * STAGE;
        __syncthreads();                         // stage `step` landed (vmcnt(0) precedes the barrier); step-1 is consumed
        if (step + 1 < nsteps) issue(step + 1, smem + ((step + 1) & 1) * STAGE);
        bf16x8 af[3];
#pragma unroll
        for (int R = 0; R < PH; ++R) {
            bf16x8 bf[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) bf[dx] = wg2_frag(cur + b_off + (R * PW + dx) * 64);
            if (R < TH) {
                af[R % 3] = wg2_frag(cur + a_off + (R * TW) * 64);
                if (do_bias) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[R % 3], ones, bacc, 0, 0, 0);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = R - dy;                                        // n-side row that meets patch row R under tap row dy
                if (r >= 0 && r < TH) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[r % 3], bf[dx], acc[dy * 3 + dx], 0, 0, 0);
                }
            }
        }
    }

    // ---- sum the two column-segment waves of each channel pair through LDS (the stages are dead), 16 floats per lane a round
    __syncthreads();
    // (lane-contiguous float4 slots: the [lane][16 floats] layout this replaced put 8 / 16 lanes of a ds_write_b128 / ds_read_b128 group on 2 / 4
    // banks' worth of addresses -- all of this kernel's 15 % LDS bank conflicts, profiles/r06_pmc_mfma_bf16_b32.json)
    float4* red = reinterpret_cast<float4*>(smem) + pair * 256 + lane;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        f32x16& v = t < 9 ? acc[t] : bacc;
        if (seg == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q * 64] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
        __syncthreads();
        if (seg == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 o = red[q * 64];
                v[4 * q] += o.x; v[4 * q + 1] += o.y; v[4 * q + 2] += o.z; v[4 * q + 3] += o.w;
            }
        }
        __syncthreads();
    }
    if (seg != 0) return;
    float* out = a.out + (size_t)split * a.split_stride;
#pragma unroll
    for (int t = 0; t < 9; ++t) wg2_store_block(out + (size_t)t * a.Cn * a.Ck, acc[t], a.Ck, n0 + nb * 32, k0 + kb * 32 + l31, hi);
    if (do_bias && l31 == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(size_t)9 * a.Cn * a.Ck + n0 + nb * 32 + 8 * g + 4 * hi + j] = bacc[4 * g + j];
    }
}

// ------------------------------------------------------------------------------------------------------------- W2_D
template <int KBW>
__global__ __launch_bounds__(512, 2) void wgrad2_d_kernel(Wg2Args a) {
    constexpr int TH = 4, TW = 16, PH = TH + 1, PW = TW + 1;
    constexpr int A_PLANE = TH * TW * 64, A_INSTR = A_PLANE / 1024;                                  // 4 KB, 4 instructions
    constexpr int B_PIX = PH * PW, B_INSTR = (B_PIX * 64 + 1023) / 1024, B_PLANE = B_INSTR * 1024;     // 85 pixels -> 6 KB
    constexpr int STAGE = 2 * A_PLANE + 4 * KBW * B_PLANE, NINSTR = 2 * A_INSTR + 4 * KBW * B_INSTR, NPI = (NINSTR + 7) / 8;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, j8 = bid >> 3;
    const int nct = a.nct_n * a.nct_k;
    const int ct = j8 % nct, split = (j8 / nct) * 8 + xcd;
    if (split >= a.nsplit) return;
    const int n0 = (ct / a.nct_k) * 64, k0 = (ct % a.nct_k) * (32 * KBW);
    const int nsteps = (a.ntiles - split + a.nsplit - 1) / a.nsplit;

    // DMA descriptors.  n side (coarse): pixel (r, c) of the tile.  k side (fine): phase (py, px), patch position (pr, pc) ->
    // fine pixel (2*(ty0 + pr) - py, 2*(tx0 + pc) - px): py = 0 reads coarse rows oy, oy+1 of its phase image, py = 1 rows
    // oy-1, oy (likewise in x), which is what the taps ky = 2e+1-py of a 4x4 stride-2 pad-1 kernel touch.
    int rel[NPI], posy[NPI], posx[NPI];
#pragma unroll
    for (int jj = 0; jj < NPI; ++jj) {
        const int ii = jj * 8 + wave, c16 = (lane & 3) * 8;
        rel[jj] = 0; posy[jj] = -1000; posx[jj] = 0;
        if (ii < 2 * A_INSTR) {
            const int plane = ii / A_INSTR, pix = (ii % A_INSTR) * 16 + (lane >> 2), r = pix / TW, c = pix % TW;
            rel[jj] = (r * a.Wn + c) * a.Cn + plane * 32 + c16;
            posy[jj] = r; posx[jj] = c;
        } else if (ii < NINSTR) {
            const int j = ii - 2 * A_INSTR, ph = j / (KBW * B_INSTR), plane = (j / B_INSTR) % KBW, pix = (j % B_INSTR) * 16 + (lane >> 2);
            if (pix < B_PIX) {
                const int pr = pix / PW, pc = pix % PW, fy = 2 * pr - (ph >> 1), fx = 2 * pc - (ph & 1);
                rel[jj] = (fy * a.Wk + fx) * a.Ck + plane * 32 + c16;
                posy[jj] = fy; posx[jj] = fx;
            }
        }
    }
    const unsigned long long zaddr = reinterpret_cast<unsigned long long>(wg2_zero_page) + (lane & 3) * 16;
    auto tile_coords = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx_i = t % a.tiles_x; t /= a.tiles_x;
        const int ty_i = t % a.tiles_y;
        b = t / a.tiles_y; ty0 = ty_i * TH; tx0 = tx_i * TW;
    };
    auto issue = [&](int step, char* buf) {
        int b, ty0, tx0;
        tile_coords(split + step * a.nsplit, b, ty0, tx0);
        const bf16_t* abase = a.nside + (((long)b * a.Hn + ty0) * a.Wn + tx0) * a.Cn + n0;
        const bf16_t* bbase = a.kside + (((long)b * a.Hk + 2 * ty0) * a.Wk + 2 * tx0) * a.Ck + k0;
#pragma unroll
        for (int jj = 0; jj < NPI; ++jj) {
            const int ii = jj * 8 + wave;
            if (ii < NINSTR) {
                const bool isb = ii >= 2 * A_INSTR;                           // wave uniform
                const int gy = (isb ? 2 * ty0 : ty0) + posy[jj], gx = (isb ? 2 * tx0 : tx0) + posx[jj];
                const int H = isb ? a.Hk : a.Hn, W = isb ? a.Wk : a.Wn;
                const unsigned long long ok = (((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W)) ? ~0ull : 0ull;
                const unsigned long long pa = reinterpret_cast<unsigned long long>((isb ? bbase : abase) + rel[jj]);
                wg2_glds16(reinterpret_cast<const void*>(zaddr + ((pa - zaddr) & ok)), buf + ii * 1024);
            }
        }
    };

    const int ph = wave & 3, nb = wave >> 2, py = ph >> 1, px = ph & 1;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int lane_off = (((g16 >> 1) * 8 + (i16 >> 2)) * 64) + (g16 & 1) * 32 + (i16 & 3) * 8;
    const int a_off = nb * A_PLANE + lane_off;
    const int b_off = 2 * A_PLANE + ph * KBW * B_PLANE + lane_off;
    const bool do_bias = a.want_bias && k0 == 0 && ph == 0;

    f32x16 acc[2][2][KBW], bacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bacc[r] = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int k = 0; k < KBW; ++k) acc[e][f][k][r] = 0.f;
    }
    const bf16x8 ones = wg2_ones();

    if (nsteps > 0) issue(0, smem);
    for (int step = 0; step < nsteps; ++step) {
        const char* cur = smem + (step & 1) * STAGE;
        __syncthreads();
        if (step + 1 < nsteps) issue(step + 1, smem + ((step + 1) & 1) * STAGE);
        bf16x8 af[2];
#pragma unroll
        for (int R = 0; R < PH; ++R) {
            bf16x8 bf[2][KBW];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int k = 0; k < KBW; ++k) bf[f][k] = wg2_frag(cur + b_off + k * B_PLANE + (R * PW + f) * 64);
            if (R < TH) {
                af[R & 1] = wg2_frag(cur + a_off + (R * TW) * 64);
                if (do_bias) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[R & 1], ones, bacc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = R - e;                                         // coarse row that meets patch row R at offset e
                if (r >= 0 && r < TH) {
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int k = 0; k < KBW; ++k)
                            acc[e][f][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[r & 1], bf[f][k], acc[e][f][k], 0, 0, 0);
                }
            }
        }
    }

    float* out = a.out + (size_t)split * a.split_stride;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int t = (2 * e + 1 - py) * 4 + (2 * f + 1 - px);
#pragma unroll
            for (int k = 0; k < KBW; ++k)
                wg2_store_block(out + (size_t)t * a.Cn * a.Ck, acc[e][f][k], a.Ck, n0 + nb * 32, k0 + k * 32 + l31, hi);
        }
    if (do_bias && l31 == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(size_t)16 * a.Cn * a.Ck + n0 + nb * 32 + 8 * g + 4 * hi + j] = bacc[4 * g + j];
    }
}

// ------------------------------------------------------------------------------------------------------------- W2_S16
// The 16 x 16-channel weights of the 1024x1024 level (3x3, both networks; round 3): dW[tap][n][k] over ALL pixels is one
// 16 x 16 tile per tap, so the whole problem is bandwidth -- both activations exactly once.  v_mfma_f32_16x16x32_bf16: M = the 16
// dy channels, N = the 16 x channels, K = the 32 pixels of one tile row: ONE MFMA per (row, tap).  Tile 16 rows x 32 pixels, k-side
// patch 18 x 34; a wave owns two tile rows (four patch rows, read once: 12 K-major fragments via ds_read_b64_tr_b16) and keeps
// all nine taps' 16 x 16 accumulators (+ the bias column sums) in 40 registers; the eight waves are summed through LDS once, at
// the end.  Operands pixel-major with a 32-byte pixel pitch (16 channels): a 16-lane group's transpose read touches 4 pixels =
// 128 contiguous bytes, and the K index <-> pixel mapping (group g: pixels 4g..4g+3 and 16+4g..16+4g+3 of the row) makes the two
// groups of a half-wave read adjacent 128-byte runs: all 64 banks once.  36 KB per stage, two stages, two blocks per CU; each XCD
// walks one contiguous eighth of the tile raster (halo rows / columns shared through its L2).
typedef __attribute__((ext_vector_type(4))) float f32x4w;
__device__ __forceinline__ bf16x8 wg16_frag(const char* p) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x4 lo = wg2_tr16(p), hi = wg2_tr16(p + 512);               // pixels +0..3 and +16..19 of the lane's group
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__global__ __launch_bounds__(512, 2) void wgrad16_s_kernel(Wg2Args a) {
    constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2;
    constexpr int A_BYTES = TH * TW * 32, A_INSTR = A_BYTES / 1024;                         // 16 KB
    constexpr int B_PIX = PH * PW, B_INSTR = (B_PIX * 32 + 1023) / 1024, B_BYTES = B_INSTR * 1024;   // 612 pixels -> 20 KB
    constexpr int STAGE = A_BYTES + B_BYTES, NINSTR = A_INSTR + B_INSTR, NPI = (NINSTR + 7) / 8;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, lslot = bid >> 3, per = a.nsplit >> 3;
    const int split = lslot * 8 + xcd;
    if (split >= a.nsplit) return;
    // XCD `xcd` owns tiles [xcd*band, (xcd+1)*band) of the raster; its `per` blocks walk them in order
    const int band = (a.ntiles + 7) >> 3;
    const int band_len = a.ntiles - xcd * band < band ? a.ntiles - xcd * band : band;
    const int nsteps = lslot < band_len ? (band_len - lslot + per - 1) / per : 0;
    const int tile0 = xcd * band + lslot;

    // per-lane DMA descriptors: instruction ii fills LDS bytes [ii*1024, +1024) = 32 pixels x 32 bytes; lane -> pixel lane/2, chunk lane%2
    int rel[NPI], pos[NPI];
#pragma unroll
    for (int jj = 0; jj < NPI; ++jj) {
        const int ii = jj * 8 + wave, c8 = (lane & 1) * 8;
        rel[jj] = 0; pos[jj] = -1;
        if (ii < A_INSTR) {
            const int r = ii, c = lane >> 1;
            rel[jj] = (r * a.Wn + c) * 16 + c8;
            pos[jj] = (r << 8) | c;
        } else if (ii < NINSTR) {
            const int pix = (ii - A_INSTR) * 32 + (lane >> 1);
            if (pix < B_PIX) {
                const int pr = pix / PW, pc = pix % PW;
                rel[jj] = (pr * a.Wk + pc) * 16 + c8;
                pos[jj] = (pr << 8) | pc;
            }
        }
    }
    const unsigned long long zaddr = reinterpret_cast<unsigned long long>(wg2_zero_page) + (lane & 3) * 16;
    auto issue = [&](int step, char* buf) {
        int t = tile0 + step * per;
        const int tx_i = t % a.tiles_x; t /= a.tiles_x;
        const int ty_i = t % a.tiles_y, b = t / a.tiles_y;
        const int ty0 = ty_i * TH, tx0 = tx_i * TW;
        const bf16_t* abase = a.nside + (((long)b * a.Hn + ty0) * a.Wn + tx0) * 16;
        const bf16_t* bbase = a.kside + (((long)b * a.Hk + ty0 - 1) * a.Wk + tx0 - 1) * 16;
#pragma unroll
        for (int jj = 0; jj < NPI; ++jj) {
            const int ii = jj * 8 + wave;
            if (ii < NINSTR) {
                const bool isb = ii >= A_INSTR;                              // wave uniform
                const int pp = pos[jj];
                const int gy = (isb ? ty0 - 1 : ty0) + (pp >> 8), gx = (isb ? tx0 - 1 : tx0) + (pp & 255);
                const unsigned long long ok = ((pp >= 0) & ((unsigned)gy < (unsigned)a.Hn) & ((unsigned)gx < (unsigned)a.Wn)) ? ~0ull : 0ull;
                const unsigned long long pa = reinterpret_cast<unsigned long long>((isb ? bbase : abase) + rel[jj]);
                wg2_glds16(reinterpret_cast<const void*>(zaddr + ((pa - zaddr) & ok)), buf + ii * 1024);
            }
        }
    };

    const int g16 = lane >> 4, i16 = lane & 15;
    const int lane_off = (g16 * 4 + (i16 >> 2)) * 32 + (i16 & 3) * 8;
    const int r0 = 2 * wave;
    f32x4w acc[9], bacc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bacc[q] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t][q] = 0.f;
    }
    const bf16x8 ones = wg2_ones();

    if (nsteps > 0) issue(0, smem);
    for (int step = 0; step < nsteps; ++step) {
        const char* cur = smem + (step & 1) * STAGE;
        __syncthreads();                         // stage `step` landed (vmcnt(0) precedes the barrier); step-1 is consumed
        if (step + 1 < nsteps) issue(step + 1, smem + ((step + 1) & 1) * STAGE);
        bf16x8 bf[4][3];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) bf[pr][dx] = wg16_frag(cur + A_BYTES + ((r0 + pr) * PW + dx) * 32 + lane_off);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const bf16x8 af = wg16_frag(cur + (r0 + rr) * TW * 32 + lane_off);
            if (a.want_bias) bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ones, bacc, 0, 0, 0);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[rr + dy][dx], acc[dy * 3 + dx], 0, 0, 0);
        }
    }

    // ---- sum the eight waves through LDS (the stages are dead): 4 -> 2 -> 1, fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float* dst = red + ((wave - half) * 64 + lane) * 40;
#pragma unroll
            for (int t = 0; t < 9; ++t) *reinterpret_cast<float4*>(dst + 4 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
            *reinterpret_cast<float4*>(dst + 36) = make_float4(bacc[0], bacc[1], bacc[2], bacc[3]);
        }
        __syncthreads();
        if (wave < half) {
            const float* src = red + (wave * 64 + lane) * 40;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 o = *reinterpret_cast<const float4*>(src + 4 * t);
                acc[t][0] += o.x; acc[t][1] += o.y; acc[t][2] += o.z; acc[t][3] += o.w;
            }
            const float4 o = *reinterpret_cast<const float4*>(src + 36);
            bacc[0] += o.x; bacc[1] += o.y; bacc[2] += o.z; bacc[3] += o.w;
        }
        __syncthreads();
    }
    if (wave != 0) return;
    // C/D layout of the 16x16 MFMA: column (x channel k) = lane & 15, row (dy channel n) = 4 * (lane >> 4) + register
    float* out = a.out + (size_t)split * a.split_stride;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[(size_t)t * 256 + (size_t)(4 * g16 + q) * 16 + i16] = acc[t][q];
    if (a.want_bias && i16 == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) out[(size_t)9 * 256 + 4 * g16 + q] = bacc[q];
    }
}

// ------------------------------------------------------------------------------------------------------------- host side
static int wg2_ncu() { return sgx_ncu(); }
static int wg2_switch() {                        // SGX_WGRAD2: bit 0 = the 3x3 geometry, bit 1 = the 4x4 stride-2 geometry, bit 2 = 3x3 16x16 channels (A/B)
    static const int on = [] { const char* e = getenv("SGX_WGRAD2"); return e ? atoi(e) : 7; }();
    return on;
}
static int wg2_max_ct() { static const int v = [] { const char* e = getenv("SGX_WGRAD2_MAXCT"); return e && atoi(e) > 0 ? atoi(e) : 32; }(); return v; }

// Which layers take these kernels, and with how many pixel splits.  geo 0: 3x3 (H, W = the common size); 1: 4x4 stride 2
// (H, W = the COARSE size).  Returns 0 if the shape stays with the first generation.
int sgx_wgrad2_plan(int geo, int B, int H, int W, int Ck, int Cn, int* nct_n, int* nct_k, int* kbw, int* ntiles) {
    if (!((wg2_switch() >> geo) & 1)) return 0;
    int th, tw, kt;
    if (geo == 0 && Ck == 16 && Cn == 16) {           // the 16 x 16-channel weights: wgrad16_s_kernel (bit 2 of SGX_WGRAD2)
        if (!(wg2_switch() & 4) || W % 32) return 0;
        *nct_n = 1; *nct_k = 1; *kbw = 16;
        *ntiles = B * ((H + 15) / 16) * (W / 32);
        int ns = 2 * wg2_ncu() / 8 * 8;                // two blocks per CU
        if (*ntiles < 2 * ns) ns = *ntiles / 2 / 8 * 8;
        return ns >= 8 ? ns : 0;
    }
    if (geo == 0) {
        if (Ck % 64 || Cn % 64 || W % 32 || H % 8) return 0;
        th = 8; tw = 32; kt = 64; *kbw = 2;
    } else {
        if (Cn % 64 || Ck % 32 || W % 16 || H % 4) return 0;
        th = 4; tw = 16; *kbw = Ck % 64 == 0 ? 2 : 1; kt = 32 * *kbw;
    }
    *nct_n = Cn / 64; *nct_k = Ck / kt;
    *ntiles = B * (H / th) * (W / tw);
    const int nct = *nct_n * *nct_k;
    if (nct > 2 * wg2_max_ct()) return 0;
    int ns = wg2_ncu() / nct / 8 * 8;              // ~ one block per CU, a multiple of 8 (one split per XCD slot)
    if (ns < 8) ns = 8;
    // big weights at low resolution (512 x 512 channels: 64 tiles, 9-17 MB of partials per split): only when every block
    // walks >= 8 pixel tiles, i.e. the operand traffic outweighs the partials (probe at 32^2: batch 32 255 -> 160 us and
    // 166 -> 107 us, batch 4 no gain / 31 -> 57 us: there the first generation's single-split no-partials path stays)
    if (nct > wg2_max_ct() && *ntiles < 8 * ns) return 0;
    if (*ntiles < 2 * ns) ns = *ntiles / 2 / 8 * 8;  // small problems: fewer splits, every block still pipelines >= 2 tiles
    return ns >= 8 ? ns : 0;                        // too few pixel tiles: first generation
}

int sgx_wgrad2_launch(int geo, const void* kside, const void* nside, float* ws, size_t ws_bytes, int B, int H, int W, int Ck, int Cn,
                      int want_bias, hipStream_t st, int* nsplit_out) {
    int nct_n, nct_k, kbw, ntiles;
    int ns = sgx_wgrad2_plan(geo, B, H, W, Ck, Cn, &nct_n, &nct_k, &kbw, &ntiles);
    *nsplit_out = 0;
    if (!ns) return 0;
    const int NT = geo == 0 ? 9 : 16;
    const size_t total = (size_t)NT * Cn * Ck + Cn;
    if ((size_t)ns * total * sizeof(float) > ws_bytes) return 0;
    Wg2Args a{static_cast<const bf16_t*>(kside), static_cast<const bf16_t*>(nside), ws, B, geo == 0 ? H : 2 * H, geo == 0 ? W : 2 * W, H, W, Ck, Cn,
              W / (geo == 0 ? 32 : 16), H / (geo == 0 ? 8 : 4), ntiles, ns, nct_n, nct_k, want_bias, (long)total};
    const dim3 grid((unsigned)(nct_n * nct_k * ns)), block(512);
    if (geo == 0 && kbw == 16) {
        a.tiles_y = (H + 15) / 16;
        constexpr int LDS = 2 * (16 * 32 * 32 + ((18 * 34 * 32 + 1023) / 1024) * 1024);
        static_assert(LDS >= 4 * 64 * 40 * 4, "the final reduction's scratch fits the stages");
        sgx_lds_opt_in<wgrad16_s_kernel>(LDS);
        hipLaunchKernelGGL(wgrad16_s_kernel, grid, block, LDS, st, a);
    } else if (geo == 0) {
        constexpr int LDS = 2 * (2 * 8 * 32 * 64 + 2 * ((10 * 34 * 64 + 1023) / 1024) * 1024);
        sgx_lds_opt_in<wgrad2_s_kernel<8>>(LDS);
        hipLaunchKernelGGL(wgrad2_s_kernel<8>, grid, block, LDS, st, a);
    } else if (kbw == 2) {
        constexpr int LDS = 2 * (2 * 4096 + 8 * 6144);
        sgx_lds_opt_in<wgrad2_d_kernel<2>>(LDS);
        hipLaunchKernelGGL(wgrad2_d_kernel<2>, grid, block, LDS, st, a);
    } else {
        constexpr int LDS = 2 * (2 * 4096 + 4 * 6144);
        sgx_lds_opt_in<wgrad2_d_kernel<1>>(LDS);
        hipLaunchKernelGGL(wgrad2_d_kernel<1>, grid, block, LDS, st, a);
    }
    SGX_LAUNCH_CHECK("wgrad2_kernel");
    *nsplit_out = ns;
    return 0;
}
