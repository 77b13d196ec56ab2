// Second-generation bf16 convolution kernels for gfx950: v_mfma_f32_32x32x16_bf16, operands staged by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) into TWO LDS stages, one barrier per K-step.
//
// Why (round-1 profile, profiles/r01_*): the first-generation kernel (conv.hip) runs one wave per SIMD with a single LDS
// stage -- global load -> ds_write -> barrier -> MFMA -> barrier serialise, and at 64^2..256^2 (the MFMA-bound layers) it
// reaches 12-24 % of the matrix peak while neither LDS bandwidth (6 %) nor HBM is the limit.  Here a block is 8 waves
// (two per SIMD; 4 for small problems) on a (2*NW) x 32 pixel tile; while the waves run the MFMAs of K-step s out of
// stage s&1, the DMA engine fills stage (s+1)&1 with the next step (or the next tile's first: the block is persistent).
//
// GEMM view as in conv.hip: M = output channels (A = packed weights w[tap][n][k]), N = pixels (B = activations),
// K = taps x input channels in K-steps of 32 channels.  A wave owns two rows of 32 pixels x MF*32 channels.
//
// Three geometries share the kernel:
//   C2_S : 3x3 stride 1 pad 1.  Patch (TH+2) x 34 pixels with halo, 9 taps per K-step.
//   C2_D : 4x4 stride 2 pad 1, as FOUR 2x2 stride-1 convolutions over the polyphase components of the input: a K-step is
//          (phase (py,px), 32 channels); its patch is the (TH+1) x 33 block of the phase's sub-image, its weights the four
//          taps (ky,kx) = (2a+1-py, 2b+1-px).  An input pixel is used by 4 taps only, so nothing is gained by staging the
//          whole 4x-larger stride-2 patch at once; this way a stage is 52 KB instead of 140.
//   C2_U : 4x4 stride 2 pad 1 TRANSPOSED: the four output-parity classes (py,px) are 2x2 convolutions over the coarse
//          grid; all four accumulate in one block from ONE 3x3-halo patch, class (py,px) using patch offset (dy,dx) with
//          tap (ky,kx) = (3-py-2(dy-py), 3-px-2(dx-px)) when dy-py, dx-px are 0 or 1.  The block stores whole fine rows.
//
// LDS image of one stage: [patch rows][weight rows], every row = 32 channels = 64 bytes = four 16-byte slots.  The DMA
// writes lane-linear (wave-uniform base + lane*16), so slot s of an instruction holds (row s/4, chunk (s%4) ^ swz(row)):
// the swizzle is applied on the per-lane SOURCE address and again on the fragment read (cdna guide rule 21).
//   patch row (pr, pc): swz = (pc >> 2) & 3   -- a fragment's 32 lanes read 32 consecutive pc of one patch row
//   weight row (tap, n): swz = (n >> 2) & 3   -- 32 consecutive n
// which makes every ds_read_b128 conflict-free (16-lane service groups {0-3,12-15,20-27}/{4-11,16-19,28-31} hit 16
// distinct 16-byte bank groups for any starting column).  Halo / out-of-image lanes read a 64-byte page of zeros.
//
// 16-channel layers (round 3; the 1024x1024 layers of both networks: 3x3 16->16, stride-2 16->32, transposed 32->16):
//   KC = 16: a K-step is 16 input channels = ONE 32x32x16 MFMA per tap; rows are 32 bytes = two 16-byte slots, stored
//            PLANAR (slot = chunk * rows + row): a fragment's 32 lanes read 32 consecutive slots of one plane, conflict-free
//            for any column shift without a swizzle.
//   CO16:    16 output channels in a 32-channel block: weight rows 16..31 are DMA'd from the page of zeros (half of each
//            MFMA is padding -- these layers are HBM-bound by a factor of 3), the store writes 32 bytes per pixel.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __attribute__((aligned(64))) const unsigned sgx_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the same with the nontemporal cache policy (aux = 2): the patch stream of a tensor far larger than the caches (Conv2Args.nt bit 1, probe)
__device__ __forceinline__ void glds16_nt(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

enum { C2_S = 0, C2_D = 1, C2_U = 2 };

struct Conv2Args {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
    const bf16_t* mask;                       // C2_S only: y *= slope(mask) in the store (shaped like y), see sgx_conv3x3
    int B, H, W, OH, OW, Cin, Cout, act;      // H, W: input; OH, OW: output
    int tiles_x, tiles_y, ntiles;             // tiles of the tile grid (S: the image, D: the output, U: the coarse input)
    int ncb, nslots;                          // channel blocks; persistent stride over tiles
    int bands;                                // tile order: 0 = slot, slot + nslots, ...; 1 = each XCD walks one contiguous eighth of the raster
    // EPI_STATS (C2_S): the LayerEpilogue that consumes y takes its instance-norm statistics from this store: per (image, tile
    // slot of the persistent grid) the sums of a = lrelu(y + ebias[c] + enw[c] * enoise[b, pixel]) and a^2 over the pixels of the
    // image that the slot's blocks stored, y as stored (bf16)
    const float* ebias; const float* enoise; const float* enw; double* part;   // part[((b * nslots + slot) * Cout + c) * 2 + {0, 1}]
    // C2_S, register epilogue: one SIGN bit per stored element (1 = value > 0), [pixel][Cout / 8] bytes, bit j = channel 8v + j --
    // what the LeakyReLU backward of the discriminator block needs of the pre-activation (1/16 of re-reading it)
    unsigned char* signbits;
    // EPI_BLUR: the activation mask as those sign bits (1 bit per element instead of 16) -- overrides `mask`
    const unsigned char* maskbits;
    // C2_D, register epilogue: the fade-in lerp of the discriminator's newest block (models/GAN.py:427) in the store,
    // y = fade_alpha * act(conv + bias) + fade_beta * fade_resid, on the bf16-ROUNDED activation (bit for bit what sgx_axpby computes from
    // the stored tensor); `signbits` then receives the sign bits of the activation (its LeakyReLU-backward mask: the activation itself
    // is never stored)
    const bf16_t* fade_resid;
    float fade_alpha, fade_beta;
    const float* fade_ab;                     // ... or the two coefficients in device memory ([alpha, beta]: a captured step graph must not bake them in)
    // ... or the residual COMPUTED in the store (round 5): fade_resid[pixel][c] = bf16(rb[c] + (r * (ws * W[c][0]) + g * (ws * W[c][1])) +
    // b * (ws * W[c][2])) with (r, g, b) = fade_pimg[pixel] -- from_rgb of the down-sampled image (models/GAN.py:423-427), the arithmetic of
    // sgx_rgb_in on a bf16 output, bit for bit -- instead of a [pixel][Cout] tensor written by one pass and read back here
    const float* fade_pimg; const float* fade_wr; const float* fade_rb; float fade_ws, fade_bs1, fade_bs2;
    int part_slots;                           // EPI_STATS: the slot count `part` was sized for (launch_conv2 refuses any other nslots)
    const bf16_t* wcorr;                      // conv3_kernel<UB>: the 22 border-correction tiles [22][32][32] behind the 9 composite taps of the same pack
    int nt;                                   // nontemporal hints (launch_conv2 / launch_conv3, SGX_CONV_NT, outputs of >= 192 MB): bit 0 the output stores, bit 1 the patch DMA
    int ureg;                                 // C2_U without the blur epilogue: 0 = the LDS-transposed store (default), 1 = the register epilogue (A/B: SGX_CONVU_REGSTORE=1, measured 0.5 % slower)
    int dbg;                                  // conv3_kernel, probe launches only (sgx_conv_variant + SGX_CONV3_DBG): DMA ablations, WRONG results by design
};
enum { EPI_NONE = 0, EPI_STATS = 1, EPI_BLUR = 2 };

template <int GEO> struct G2;
template <> struct G2<C2_S> { static constexpr int NPH = 1, HALO = 2, IS = 1, NTW = 9, NDX = 3, NDY = 3, NCLS = 1; };
template <> struct G2<C2_D> { static constexpr int NPH = 4, HALO = 1, IS = 2, NTW = 4, NDX = 2, NDY = 2, NCLS = 1; };
template <> struct G2<C2_U> { static constexpr int NPH = 1, HALO = 2, IS = 1, NTW = 16, NDX = 3, NDY = 3, NCLS = 4; };

template <int GEO, int NW, int MF, int KC = 32> struct C2Lds {
    static constexpr int TH = 2 * NW, PH = TH + G2<GEO>::HALO, PW = 32 + G2<GEO>::HALO, BCO = MF * 32;
    static constexpr int PROWS = PH * PW;
    static constexpr int SPR = KC / 8;                                              // 16-byte slots per row
    static constexpr int WROWS = G2<GEO>::NTW * BCO;
    static constexpr int OROW = BCO * 2 + 16;
    // epilogue scratch: per-wave pixel-major rows (the LDS-transposed stores: transposed geometry, statistics); the 128-channel block
    // (MF = 4) only has the register epilogue
    static constexpr int SCRATCH = MF > 2 ? 0 : NW * (GEO == C2_U ? 64 : 32) * OROW;
    static constexpr int P_INSTR = (PROWS * SPR + 63) / 64;
    static constexpr int P_RAW = P_INSTR * 1024;
    static constexpr int P_BYTES = ((P_RAW > SCRATCH ? P_RAW : SCRATCH) + 1023) / 1024 * 1024;
    static constexpr int W_INSTR = WROWS * SPR / 64, W_BYTES = W_INSTR * 1024;
    static_assert(WROWS * SPR % 64 == 0, "weight stage: whole DMA instructions");
    static constexpr int STAGE = P_BYTES + W_BYTES;
    static constexpr int TOTAL = 2 * STAGE;
};

// 8 channels (element offset doff = pixel * Cout + c0, c0 % 8 == 0) of the residual branch computed from the pooled image: exactly
// sgx_rgb_in's arithmetic (pointwise.hip rgb_in_kernel: weights pre-multiplied by the scale, (r w0 + g w1) + b w2, bias added last,
// no contraction) rounded to bf16 like the tensor that kernel would have stored.  fw: the block's table [BCO][4] = (ws w[c][0..2], bias)
// in LDS (fade_table_fill), c_local = the first of the 8 channels inside the block's channel range.
__device__ __forceinline__ void fade_table_fill(const Conv2Args& a, float* fw, int co0, int nch, int tid) {
    if (tid < nch) {
        const int c = co0 + tid;
        const float w0 = a.fade_ws * a.fade_wr[c * 3], w1 = a.fade_ws * a.fade_wr[c * 3 + 1], w2 = a.fade_ws * a.fade_wr[c * 3 + 2];
        const float bb = a.fade_rb ? (a.fade_rb[c] * a.fade_bs1) * a.fade_bs2 : 0.f;
        reinterpret_cast<float4*>(fw)[tid] = make_float4(w0, w1, w2, bb);
    }
}
__device__ __forceinline__ uint4 fade_resid_from_image(const float* fw, float r, float g, float b, int c_local) {
    unsigned o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t0 = reinterpret_cast<const float4*>(fw)[c_local + 2 * q], t1 = reinterpret_cast<const float4*>(fw)[c_local + 2 * q + 1];
        o[q] = pack_bf16x2(t0.w + (r * t0.x + g * t0.y + b * t0.z), t1.w + (r * t1.x + g * t1.y + b * t1.z));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// NW waves per block, each owning 2 rows x 32 pixels of the tile grid; MF 32-channel accumulator rows per wave.
// KC: input channels per K-step (32; 16 = the planar half-width stage).  CO16: 16 real output channels in the 32-channel block.
template <int GEO, int NW, int MF, int KC = 32, bool CO16 = false, int EPI = EPI_NONE>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv2_kernel(Conv2Args a) {
    using G = G2<GEO>;
    using L = C2Lds<GEO, NW, MF, KC>;
    constexpr int TH = L::TH, PW = L::PW, BCO = L::BCO, PROWS = L::PROWS, WROWS = L::WROWS;
    constexpr int P_INSTR = L::P_INSTR, P_BYTES = L::P_BYTES, W_INSTR = L::W_INSTR, STAGE = L::STAGE;
    constexpr int NPI = (P_INSTR + NW - 1) / NW, NWI = (W_INSTR + NW - 1) / NW;
    constexpr int OROW = L::OROW, VPR = CO16 ? 2 : BCO * 2 / 16;        // 16-byte vectors per stored pixel
    constexpr int NPH = G::NPH, IS = G::IS, NDX = G::NDX, NDY = G::NDY, NCLS = G::NCLS;
    constexpr int KS = KC / 16;                                          // 32x32x16 MFMAs per tap and K-step
    static_assert(KC == 32 || KC == 16, "K-step width");
    static_assert(!CO16 || MF == 1, "16 output channels: one (half used) 32-channel block");
    static_assert(GEO != C2_U || MF == 1, "four parity classes of accumulators: one 32-channel row per wave");
    static_assert(EPI != EPI_STATS || GEO == C2_S, "the statistics epilogue is built for the 3x3 geometry");
    static_assert(EPI != EPI_BLUR || GEO == C2_U, "the blur epilogue is built for the transposed geometry");
    static_assert(MF <= 2 || (EPI == EPI_NONE && GEO != C2_U), "the 128-channel block: register epilogue only");
    // EPI_BLUR: y = blur3x3(conv(x)) [* slope(mask)] -- the [1,2,1]x[1,2,1]/16 blur that follows the transposed convolution in
    // both networks (generator: conv0_up -> blur, models/CustomLayers.py:176-177; discriminator backward: the adjoint of
    // "LeakyReLU -> blur -> conv1_down", models/Blocks.py:140-145) applied to the accumulators before they are stored: the
    // vertical [1,2,1] in registers (a lane holds a 2x2 fine block of both of its wave's coarse rows; the fine rows above and
    // below come from the neighbouring waves through LDS), the horizontal [1,2,1] in the transposed store's read-back.  Tiles
    // overlap by one coarse row / column (steps TH-1, 31): the first / last fine row and column of a tile lack a neighbour and
    // are stored by the adjacent tile instead -- except at the image border, where the neighbour is the blur's zero padding.
    constexpr int TSY = EPI == EPI_BLUR ? TH - 1 : TH, TSX = EPI == EPI_BLUR ? 31 : 32;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float* const ecoef = reinterpret_cast<float*>(smem + L::TOTAL);     // EPI_STATS: [2][BCO] epilogue bias / noise weight of this channel block
                                                                        // C2_D with the residual computed in the store: [BCO][4] (fade_table_fill)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (channel block, tile slot): blocks of one XCD (id % 8) that are neighbours in id/8 share a tile slot and
    // differ in the channel block, so the patches they share are served by that XCD's L2
    const int bid = blockIdx.x, xcd = bid & 7, j8 = bid >> 3;
    const int cb = j8 % a.ncb, slot = (j8 / a.ncb) * 8 + xcd;
    const int co0 = cb * BCO;
    // bands: XCD `xcd` owns tiles [xcd*band, (xcd+1)*band) of the raster and its per = nslots/8 tile slots walk them in order,
    // so raster neighbours (which share halo rows / columns) run on the same L2 close in time
    const int band = (a.ntiles + 7) >> 3, per = a.nslots >> 3, lslot = j8 / a.ncb;
    const int band_len = a.ntiles - xcd * band < band ? a.ntiles - xcd * band : band;
    const int tile0 = a.bands ? xcd * band + lslot : slot, tstride = a.bands ? per : a.nslots;
    const int my_tiles = a.bands ? (lslot < band_len ? (band_len - lslot + per - 1) / per : 0)
                                 : (slot < a.ntiles ? (a.ntiles - slot + a.nslots - 1) / a.nslots : 0);
    constexpr int REAL = CO16 ? 16 : BCO;                 // real output channels of the block
    if constexpr (EPI == EPI_STATS) {
        // this slot's partial statistics of every image start at zero -- also for slots that get no tile (the same threads write
        // the real sums later: program order)
        if (tid < 2 * REAL) {
            const int c = tid % REAL, k = tid / REAL;
            for (int bb = 0; bb < a.B; ++bb) a.part[(((size_t)bb * a.nslots + slot) * a.Cout + co0 + c) * 2 + k] = 0.0;
        }
    }
    if (my_tiles <= 0) return;
    if constexpr (GEO == C2_D && EPI == EPI_NONE) {
        if (a.fade_pimg) fade_table_fill(a, ecoef, cb * BCO, BCO, threadIdx.x);      // (read in the epilogue, behind the main loop's barriers)
    }
    const int nchunks = a.Cin / KC;
    const int spt = nchunks * NPH;                        // K-steps per tile
    const int nsteps = my_tiles * spt;

    // ---- per-lane DMA descriptors (tile independent)
    int prel[NPI], ppos[NPI], wrel[NWI];
#pragma unroll
    for (int jj = 0; jj < NPI; ++jj) {
        const int s = (jj * NW + wave) * 64 + lane;
        // KC = 32: slot s = (row, chunk c) row-major, XOR-swizzled source; KC = 16: planar, slot = chunk * PROWS + row
        const int row = KC == 32 ? (s >> 2) : (s % PROWS), c = KC == 32 ? (s & 3) : (s / PROWS);
        const int pr = row / PW, pc = row % PW;
        prel[jj] = (IS * pr * a.W + IS * pc) * a.Cin + (KC == 32 ? ((c ^ ((pc >> 2) & 3)) << 3) : (c << 3));
        ppos[jj] = (KC == 32 ? (row < PROWS) : (c < 2)) ? ((pr << 8) | pc) : -1;
    }
#pragma unroll
    for (int jj = 0; jj < NWI; ++jj) {
        const int s = (jj * NW + wave) * 64 + lane;
        const int row = KC == 32 ? (s >> 2) : (s % WROWS), c = KC == 32 ? (s & 3) : (s / WROWS);
        const int t = row / BCO, n = row % BCO;
        const int tg0 = GEO == C2_D ? (2 * (t >> 1) + 1) * 4 + 2 * (t & 1) + 1 : t;      // phase (0,0) tap; phase shifts it
        wrel[jj] = ((tg0 * a.Cout + co0 + n) * a.Cin) + (KC == 32 ? ((c ^ ((n >> 2) & 3)) << 3) : (c << 3));
        if (CO16 && n >= 16) wrel[jj] = -1;                                            // padding rows: the page of zeros
    }
    // ---- per-lane fragment read offsets inside a stage
    int poff[NDX][KS], woff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if constexpr (KC == 32) woff[ks] = P_BYTES + l31 * 64 + (((hi + 2 * ks) ^ ((l31 >> 2) & 3)) << 4);
        else woff[ks] = P_BYTES + (hi * WROWS + l31) * 16;
#pragma unroll
        for (int dx = 0; dx < NDX; ++dx) {
            const int pc = l31 + dx;
            if constexpr (KC == 32) poff[dx][ks] = (2 * wave * PW + pc) * 64 + (((hi + 2 * ks) ^ ((pc >> 2) & 3)) << 4);
            else poff[dx][ks] = (hi * PROWS + 2 * wave * PW + pc) * 16;
        }
    }
    constexpr int PROWB = KC == 32 ? 64 : 16;             // byte distance of consecutive patch / weight rows in a fragment read

    const bf16_t* __restrict__ xg = a.x;
    const bf16_t* __restrict__ wg = a.w;
    const unsigned long long zaddr = reinterpret_cast<unsigned long long>(sgx_zero_page) + (lane & 3) * 16;

    auto tile_coords = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx_i = t % a.tiles_x; t /= a.tiles_x;
        const int ty_i = t % a.tiles_y;
        b = t / a.tiles_y; ty0 = ty_i * TSY; tx0 = tx_i * TSX;
    };
    // stage K-step `step` (tile it, channel chunk kc, phase ph) into LDS stage buffer `buf`
    auto issue = [&](int step, char* buf) {
        const int it = step / spt, q = step - it * spt;
        const int kc = q / NPH, ph = q - kc * NPH, py = ph >> 1, px = ph & 1;
        int b, ty0, tx0;
        tile_coords(tile0 + it * tstride, b, ty0, tx0);
        // input pixel of patch position (pr, pc): (iy0 + IS*pr, ix0 + IS*pc)
        const int iy0 = GEO == C2_D ? 2 * ty0 - py : ty0 - 1, ix0 = GEO == C2_D ? 2 * tx0 - px : tx0 - 1;
        const bf16_t* base = xg + (((long)b * a.H + iy0) * a.W + ix0) * a.Cin + kc * KC;
#pragma unroll
        for (int jj = 0; jj < NPI; ++jj) {
            const int ii = jj * NW + wave;
            if (ii < P_INSTR) {
                const int pp = ppos[jj];
                const int gy = iy0 + IS * (pp >> 8), gx = ix0 + IS * (pp & 255);
                // branch-free select between the patch element and the page of zeros (a ?: on pointers compiles to
                // divergent branches around every DMA instruction)
                const unsigned long long ok = ((pp >= 0) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W)) ? ~0ull : 0ull;
                const unsigned long long pa = reinterpret_cast<unsigned long long>(base + prel[jj]);
                if (a.nt & 2) glds16_nt(reinterpret_cast<const void*>(zaddr + ((pa - zaddr) & ok)), buf + ii * 1024);
                else glds16(reinterpret_cast<const void*>(zaddr + ((pa - zaddr) & ok)), buf + ii * 1024);
            }
        }
        // <= 2 K-steps per tile: step q's weights live in stage q for the whole launch (not with the blur epilogue: its row
        // exchange takes the whole stage of a tile's last K-step)
        if (EPI == EPI_BLUR || spt > 2 || step < 2) {
            const bf16_t* w0 = wg + kc * KC - (GEO == C2_D ? (long)(4 * py + px) * a.Cout * a.Cin : 0);
#pragma unroll
            for (int jj = 0; jj < NWI; ++jj) {
                const int ii = jj * NW + wave;
                if (ii < W_INSTR) {
                    if constexpr (CO16) {
                        const unsigned long long ok = wrel[jj] >= 0 ? ~0ull : 0ull;
                        const unsigned long long wa = reinterpret_cast<unsigned long long>(w0 + wrel[jj]);
                        glds16(reinterpret_cast<const void*>(zaddr + ((wa - zaddr) & ok)), buf + P_BYTES + ii * 1024);
                    } else {
                        glds16(w0 + wrel[jj], buf + P_BYTES + ii * 1024);
                    }
                }
            }
        }
    };

    f32x16 acc[NCLS * MF][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < NCLS * MF; ++m)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][f][r] = 0.f;
    };
    zero_acc();
    // EPI_STATS: this lane's channel vector (lane % VPR) -- sums over the pixels it stored for image st_b, kept across the tiles
    // of the persistent loop and written out when the image changes (and at the end): no per-tile reduction
    float st0[8], st1[8];
    int st_b = -1;
    if constexpr (EPI == EPI_STATS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { st0[j] = 0.f; st1[j] = 0.f; }
        if (tid < 2 * BCO) {
            const int c = tid % BCO;
            const float* src = tid < BCO ? a.ebias : a.enw;
            ecoef[tid] = (src && c < REAL) ? src[co0 + c] : 0.f;
        }
    }
    // lanes with the same channel vector -> lane % VPR; waves -> the block, in a fixed order; one partial per (image, slot,
    // channel): deterministic, no atomics.  Called by the whole block (two barriers); scratch = the free patch region of a stage.
    auto stats_flush = [&](char* stage) {
        if constexpr (EPI == EPI_STATS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int o = 32; o >= VPR; o >>= 1) {
                    st0[j] += __shfl_xor(st0[j], o, 64);
                    st1[j] += __shfl_xor(st1[j], o, 64);
                }
            }
            float* wtot = reinterpret_cast<float*>(stage + wave * (32 * OROW));    // [VPR][16] floats at the start of this wave's scratch
            if (lane < VPR) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { wtot[lane * 16 + j] = st0[j]; wtot[lane * 16 + 8 + j] = st1[j]; }
            }
            __syncthreads();
            if (tid < 2 * REAL) {
                const int c = tid % REAL, k = tid / REAL;
                double sum = 0.0;
#pragma unroll
                for (int w8 = 0; w8 < NW; ++w8)
                    sum += (double)reinterpret_cast<const float*>(stage + w8 * (32 * OROW))[(c >> 3) * 16 + k * 8 + (c & 7)];
                a.part[(((size_t)st_b * a.nslots + slot) * a.Cout + co0 + c) * 2 + k] = sum;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) { st0[j] = 0.f; st1[j] = 0.f; }
        }
    };

    issue(0, smem);
    for (int step = 0; step < nsteps; ++step) {
        char* cur = smem + (step & 1) * STAGE;
        __syncthreads();                         // (vmcnt(0) first) stage `step` landed; everyone is done with step-1
        if (step + 1 < nsteps) issue(step + 1, smem + ((step + 1) & 1) * STAGE);
        // EPI_STATS: the noise values of the pixels this lane stores in the tile's epilogue, requested before the MFMAs of the
        // tile's last K-step so that they have landed when the epilogue needs them
        constexpr int NSI = (GEO == C2_U ? 64 : 32) * VPR / 64;       // store iterations per row
        float nzv[2][NSI];
        if constexpr (EPI == EPI_STATS) {
            const int it_ = step / spt;
            if (step - it_ * spt == spt - 1) {
                int b_, ty_, tx_;
                tile_coords(tile0 + it_ * tstride, b_, ty_, tx_);
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int i = 0; i < NSI; ++i) {
                        const int px = (i * 64 + lane) / VPR, oy = ty_ + 2 * wave + f, ox = tx_ + px;
                        nzv[f][i] = (oy < a.OH && ox < a.OW) ? a.enoise[((size_t)b_ * a.OH + oy) * a.OW + ox] : 0.f;
                    }
            }
        }
        // residual in the store (C2_D): the pooled-image pixels of the two rows this lane stores, requested before the MFMAs of the tile's
        // last K-step (like the noise values above) -- read in the epilogue they would cost one exposed load latency per store
        float pim[2][3];
        if constexpr (GEO == C2_D && EPI == EPI_NONE) {
            const int it_ = step / spt;
            if (a.fade_pimg && step - it_ * spt == spt - 1) {
                int b_, ty_, tx_;
                tile_coords(tile0 + it_ * tstride, b_, ty_, tx_);
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int oy = ty_ + 2 * wave + f, ox = tx_ + l31;
                    const bool in = oy < a.OH && ox < a.OW;
                    const float* pp = a.fade_pimg + (in ? (((size_t)b_ * a.OH + oy) * a.OW + ox) * 3 : (size_t)0);
                    pim[f][0] = pp[0]; pim[f][1] = pp[1]; pim[f][2] = pp[2];
                }
            }
        }
        if constexpr (GEO == C2_U) {
            // position-major: each of the 9 patch offsets is read once and feeds every class that has a tap there
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf16x8 brow[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) brow[r] = *reinterpret_cast<const bf16x8*>(cur + poff[dx][ks] + r * PW * PROWB);
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
                        for (int cls = 0; cls < 4; ++cls) {
                            const int py = cls >> 1, px = cls & 1, ta = dy - py, tb = dx - px;
                            if (ta >= 0 && ta <= 1 && tb >= 0 && tb <= 1) {
                                const int tg = (3 - py - 2 * ta) * 4 + (3 - px - 2 * tb);
                                const bf16x8 af = *reinterpret_cast<const bf16x8*>(cur + woff[ks] + (tg * BCO) * PROWB);
#pragma unroll
                                for (int f = 0; f < 2; ++f)
                                    acc[cls][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, brow[f + dy], acc[cls][f], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        } else {
            // NDX*2 groups (column shift dx, k-step ks) x NDY row shifts.  Per group the NDY+1 patch rows this wave touches
            // are read once; fragments are software-pipelined one sub-step (weights) / one group (patch rows) ahead.
            constexpr int NG = NDX * KS, NSUB = NG * NDY;
            bf16x8 brow[2][NDY + 1], af[2][MF];
            auto ld_brow = [&](int g, bf16x8 (&dst)[NDY + 1]) {
                const int dx = g / KS, ks = g % KS;
#pragma unroll
                for (int r = 0; r < NDY + 1; ++r) dst[r] = *reinterpret_cast<const bf16x8*>(cur + poff[dx][ks] + r * PW * PROWB);
            };
            auto ld_af = [&](int sub, bf16x8 (&dst)[MF]) {
                const int g = sub / NDY, dy = sub % NDY, dx = g / KS, ks = g % KS;
#pragma unroll
                for (int m = 0; m < MF; ++m)
                    dst[m] = *reinterpret_cast<const bf16x8*>(cur + woff[ks] + ((dy * NDX + dx) * BCO + m * 32) * PROWB);
            };
            ld_brow(0, brow[0]);
            ld_af(0, af[0]);
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const int g = sub / NDY, dy = sub % NDY;
                if (sub + 1 < NSUB) ld_af(sub + 1, af[(sub + 1) & 1]);
                if (dy == 0 && g + 1 < NG) ld_brow(g + 1, brow[(g + 1) & 1]);
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sub & 1][m], brow[g & 1][f + dy], acc[m][f], 0, 0, 0);
            }
        }
        const int it = step / spt, q = step - it * spt;
        if (q == spt - 1) {
            // ---- epilogue: bias, activation, bf16; transposed through a wave-private LDS scratch (the patch region of
            // the stage just consumed) so that every lane stores 16 bytes and neighbouring lanes cover whole channel rows
            int b, ty0, tx0;
            tile_coords(tile0 + it * tstride, b, ty0, tx0);
            if constexpr (GEO == C2_U && EPI == EPI_BLUR) {
                __syncthreads();                 // every wave is done reading this stage (patch AND weights)
                const int r0 = ty0 + 2 * wave;   // this wave's coarse rows r0 (f = 0), r0 + 1 (f = 1)
                // the blur pads with zeros OUTSIDE the image: fine pixels of coarse positions beyond it are not conv outputs
                // (only tiles that reach over the right / bottom border have any)
                if (tx0 + 32 > a.W || ty0 + TH > a.H) {
                    const bool colok = tx0 + l31 < a.W;
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const float keep = (colok && r0 + f < a.H) ? 1.f : 0.f;
#pragma unroll
                        for (int cls = 0; cls < 4; ++cls)
#pragma unroll
                            for (int j = 0; j < 16; ++j) acc[cls][f][j] *= keep;
                    }
                }
                // publish this wave's first (f 0, py 0) and last (f 1, py 1) fine row, both px, 16 channels per lane as bf16
                auto ex_at = [&](int w8, int side, int px) { return cur + ((((w8 * 2 + side) * 2 + px) * 64) + lane) * 32; };
#pragma unroll
                for (int px = 0; px < 2; ++px) {
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        const f32x16& v = side ? acc[2 + px][1] : acc[px][0];
                        uint4 lo, hi4;
                        lo.x = pack_bf16x2(v[0], v[1]); lo.y = pack_bf16x2(v[2], v[3]); lo.z = pack_bf16x2(v[4], v[5]); lo.w = pack_bf16x2(v[6], v[7]);
                        hi4.x = pack_bf16x2(v[8], v[9]); hi4.y = pack_bf16x2(v[10], v[11]); hi4.z = pack_bf16x2(v[12], v[13]); hi4.w = pack_bf16x2(v[14], v[15]);
                        char* e = ex_at(wave, side, px);
                        *reinterpret_cast<uint4*>(e) = lo;
                        *reinterpret_cast<uint4*>(e + 16) = hi4;
                    }
                }
                __syncthreads();
                // vertical [1,2,1] over the wave's four fine rows R0..R3 = (f0,py0) (f0,py1) (f1,py0) (f1,py1); the neighbour waves'
                // rows 8 channels at a time (register budget: 128 accumulators live).  The 1/16 of the filter is in the weights.
#pragma unroll
                for (int px = 0; px < 2; ++px) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        uint4 uq = make_uint4(0u, 0u, 0u, 0u), dq = make_uint4(0u, 0u, 0u, 0u);
                        if (wave > 0) uq = *reinterpret_cast<const uint4*>(ex_at(wave - 1, 1, px) + hf * 16);   // (wave-uniform branches)
                        if (wave < NW - 1) dq = *reinterpret_cast<const uint4*>(ex_at(wave + 1, 0, px) + hf * 16);
                        const unsigned uw[4] = {uq.x, uq.y, uq.z, uq.w}, dw[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int j = hf * 8 + q;
                            const float up = (q & 1) ? __uint_as_float(uw[q >> 1] & 0xffff0000u) : __uint_as_float(uw[q >> 1] << 16);
                            const float dn = (q & 1) ? __uint_as_float(dw[q >> 1] & 0xffff0000u) : __uint_as_float(dw[q >> 1] << 16);
                            const float t0 = acc[px][0][j], t1 = acc[2 + px][0][j], t2 = acc[px][1][j], t3 = acc[2 + px][1][j];
                            acc[px][0][j] = up + 2.f * t0 + t1;
                            acc[2 + px][0][j] = t0 + 2.f * t1 + t2;
                            acc[px][1][j] = t1 + 2.f * t2 + t3;
                            acc[2 + px][1][j] = t2 + 2.f * t3 + dn;
                        }
                        asm volatile("" ::: "memory");       // keep the next group's LDS loads from being hoisted over this one
                    }
                }
                __syncthreads();                 // the exchange area is read: the wave-private store scratch may overwrite it
                char* scr = cur + wave * (64 * OROW);
                // the horizontal [1,2,1] in the read-back of the transposed store (three 16-byte reads of the wave's row instead of
                // one; measured faster than DPP wave shifts on the accumulators: 817 vs 962 us on the 1024^2 layer at batch 32).
                // The mask of row r + 1 is requested while row r is processed (its latency is not hidden by anything else here).
                constexpr int NSU = 64 * VPR / 64;
                auto row_of = [&](int r, int& oy, bool& ok) {        // r = 2 * f + py
                    const int f = r >> 1, py = r & 1;
                    oy = 2 * (r0 + f) + py;
                    // a tile's first / last fine row has no neighbour row in this block unless that neighbour is outside the image
                    ok = oy < a.OH && !(wave == 0 && r == 0 && ty0 > 0) && !(wave == NW - 1 && r == 3 && ty0 + TH < a.H);
                };
                auto col_of = [&](int i, int& fpx, int& v, int& ox, bool& ok) {
                    const int idx = i * 64 + lane;
                    fpx = idx / VPR; v = idx % VPR; ox = 2 * tx0 + fpx;
                    ok = ox < a.OW && (fpx >= 1 || tx0 == 0) && (fpx <= 62 || tx0 + 32 >= a.W);
                };
                const uint4 ones4 = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);     // bf16 1.0 pairs: slope 1
                uint4 mk[2][NSU];
                auto load_mask = [&](int r, uint4 (&m)[NSU]) {
                    int oy; bool rok;
                    row_of(r, oy, rok);
#pragma unroll
                    for (int i = 0; i < NSU; ++i) {
                        int fpx, v, ox; bool cok;
                        col_of(i, fpx, v, ox, cok);
                        if (a.maskbits) {
                            // one byte = the signs of this lane's 8 channels (bit j = channel 8v + j): expanded to +-1.0 bf16 pairs so that
                            // the multiply below is the same code as for a mask tensor
                            const unsigned bb = (rok && cok) ? a.maskbits[((((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0) / 8 + v] : 0xffu;
                            unsigned w4[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                w4[q] = (((bb >> (2 * q)) & 1u) ? 0x3f80u : 0xbf80u) | ((((bb >> (2 * q + 1)) & 1u) ? 0x3f80u : 0xbf80u) << 16);
                            m[i] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                        } else {
                            m[i] = (a.mask && rok && cok) ? *reinterpret_cast<const uint4*>(a.mask + (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + v * 8) : ones4;
                        }
                    }
                };
                load_mask(0, mk[0]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = r >> 1, py = r & 1;
                    if (r + 1 < 4) load_mask(r + 1, mk[(r + 1) & 1]);
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x16& v = acc[py * 2 + px][f];
                            uint2 o;
                            o.x = pack_bf16x2(v[4 * g], v[4 * g + 1]);
                            o.y = pack_bf16x2(v[4 * g + 2], v[4 * g + 3]);
                            *reinterpret_cast<uint2*>(scr + (2 * l31 + px) * OROW + (8 * g + 4 * hi) * 2) = o;
                        }
                    }
                    int oy; bool rowok;
                    row_of(r, oy, rowok);
#pragma unroll
                    for (int i = 0; i < NSU; ++i) {
                        int fpx, v, ox; bool colv;
                        col_of(i, fpx, v, ox, colv);
                        const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
                        const uint4 cc = *reinterpret_cast<const uint4*>(scr + fpx * OROW + v * 16);
                        const uint4 ll = fpx > 0 ? *reinterpret_cast<const uint4*>(scr + (fpx - 1) * OROW + v * 16) : zero4;
                        const uint4 rr = fpx < 63 ? *reinterpret_cast<const uint4*>(scr + (fpx + 1) * OROW + v * 16) : zero4;
                        if (rowok && colv) {
                            const size_t doff = (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + v * 8;
                            const unsigned cw[4] = {cc.x, cc.y, cc.z, cc.w}, lw[4] = {ll.x, ll.y, ll.z, ll.w}, rw[4] = {rr.x, rr.y, rr.z, rr.w};
                            const uint4 mm = mk[r & 1][i];
                            const unsigned mw[4] = {mm.x, mm.y, mm.z, mm.w};
                            unsigned ow[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float h0 = __uint_as_float(lw[q] << 16) + 2.f * __uint_as_float(cw[q] << 16) + __uint_as_float(rw[q] << 16);
                                const float h1 = __uint_as_float(lw[q] & 0xffff0000u) + 2.f * __uint_as_float(cw[q] & 0xffff0000u) +
                                                 __uint_as_float(rw[q] & 0xffff0000u);
                                ow[q] = pack_bf16x2(h0 * lrelu_slope(__uint_as_float(mw[q] << 16)), h1 * lrelu_slope(__uint_as_float(mw[q] & 0xffff0000u)));
                            }
                            st16(a.y + doff, make_uint4(ow[0], ow[1], ow[2], ow[3]), a.nt & 1);
                        }
                    }
                }
            } else if constexpr (GEO == C2_U) {
              if (a.ureg) {
                // ---- register epilogue (round 6), the one of the S / D geometries: one v_permlane32_swap per dword and pair of channel groups
                // hands every lane 8 CONSECUTIVE channels of its fine pixel (2 l31 + px); the two half-waves fill the pixel's 32-byte sectors
                // together, the four stores (px, k) of a lane pair complete the 128-byte line of two neighbouring fine pixels.  No LDS
                // transposition (whose 8-byte writes at a row stride of 160 bytes were a 4-way bank conflict: 31 % of this kernel's LDS
                // cycles, profiles/r05_pmc_mfma_bf16_b32.json), no barrier before the store.
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        const int oy = 2 * (ty0 + 2 * wave + f) + py;
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            const int ox = 2 * (tx0 + l31) + px;
                            const bool inimg = oy < a.OH && ox < a.OW;
                            const size_t pix = (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + 8 * hi;
                            const f32x16& v = acc[py * 2 + px][f];
#pragma unroll
                            for (int k = 0; k < (CO16 ? 1 : 2); ++k) {
                                const unsigned lx = pack_bf16x2(v[8 * k], v[8 * k + 1]), ly = pack_bf16x2(v[8 * k + 2], v[8 * k + 3]);
                                const unsigned ux = pack_bf16x2(v[8 * k + 4], v[8 * k + 5]), uy = pack_bf16x2(v[8 * k + 6], v[8 * k + 7]);
                                auto rx = __builtin_amdgcn_permlane32_swap(lx, ux, false, false);
                                auto ry = __builtin_amdgcn_permlane32_swap(ly, uy, false, false);
                                if (inimg) st16(a.y + pix + 16 * k, make_uint4(rx[0], ry[0], rx[1], ry[1]), a.nt & 1);
                            }
                        }
                    }
                }
              } else {
                __syncthreads();                 // every wave is done reading this stage's patch
                char* scr = cur + wave * (64 * OROW);
                // LDS-transposed store.  Round 6: a lane's four 8-byte pieces of scratch row 2 l31 + px (80 bytes: the row pitch is 40 dwords
                // per l31, i.e. 8 (l31 & 3) mod 32 banks -- the 16 lanes of a ds_write_b64 group landed on FOUR bank pairs, a 4-way conflict and
                // 31 % of this kernel's LDS cycles, profiles/r05_pmc_mfma_bf16_b32.json) are XOR-swizzled inside the row by the two lane
                // bits the pitch loses: 16-byte chunk g ^ bit 2 of l31, 8-byte half hi ^ bit 3 of l31 -- 16 lanes, 16 distinct bank pairs; the
                // read-back undoes it (chunk v ^ bit, halves swapped by a select)
                const int wsw = ((((l31 >> 2) & 1) << 4) | (((l31 >> 3) & 1) << 3));           // byte XOR inside the row
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x16& v = acc[py * 2 + px][f];
                                uint2 o;
                                o.x = pack_bf16x2(v[4 * g], v[4 * g + 1]);
                                o.y = pack_bf16x2(v[4 * g + 2], v[4 * g + 3]);
                                *reinterpret_cast<uint2*>(scr + (2 * l31 + px) * OROW + (((8 * g + 4 * hi) * 2) ^ wsw)) = o;
                            }
                        }
                        const int oy = 2 * (ty0 + 2 * wave + f) + py;
#pragma unroll
                        for (int i = 0; i < 64 * VPR / 64; ++i) {
                            const int idx = i * 64 + lane, fpx = idx / VPR, v = idx % VPR;
                            const int sl = fpx >> 1;                                           // the l31 that wrote this row
                            const uint4 raw = *reinterpret_cast<const uint4*>(scr + fpx * OROW + ((v ^ ((sl >> 2) & 1)) * 16));
                            const bool swp = (sl >> 3) & 1;
                            const uint4 val = make_uint4(swp ? raw.z : raw.x, swp ? raw.w : raw.y, swp ? raw.x : raw.z, swp ? raw.y : raw.w);
                            const int ox = 2 * tx0 + fpx;
                            if (oy < a.OH && ox < a.OW)
                                st16(a.y + (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + v * 8, val, a.nt & 1);
                        }
                    }
                }
              }
            } else if constexpr (EPI == EPI_NONE) {
                // ---- register epilogue (round 3): a lane holds its pixel's channels 8g+4hi .. +3 (g = 0..3 per 32-channel row), the
                // other half-wave the 4 channels in between.  One v_permlane32_swap per dword and pair of groups hands every lane 8
                // CONSECUTIVE channels (lanes 0-31: 16k..16k+7, lanes 32-63: 16k+8..16k+15), i.e. one 16-byte store per lane and
                // pair, the two half-waves filling each pixel's 32-byte sectors together -- no LDS transposition, no barrier
                float4 bv[MF][4];
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        bv[m][g] = (a.bias && !(CO16 && g >= 2)) ? *reinterpret_cast<const float4*>(a.bias + co0 + m * 32 + 8 * g + 4 * hi)
                                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int oy = ty0 + 2 * wave + f, ox = tx0 + l31;
                    const bool inimg = oy < a.OH && ox < a.OW;
                    const size_t pix = (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + 8 * hi;
#pragma unroll
                    for (int m = 0; m < MF; ++m) {
                        uint2 o2[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float v[4] = {acc[m][f][4 * g], acc[m][f][4 * g + 1], acc[m][f][4 * g + 2], acc[m][f][4 * g + 3]};
                            v[0] += bv[m][g].x; v[1] += bv[m][g].y; v[2] += bv[m][g].z; v[3] += bv[m][g].w;
                            if (a.act == SGX_ACT_LRELU) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = lrelu(v[i]);
                            }
                            o2[g].x = pack_bf16x2(v[0], v[1]);
                            o2[g].y = pack_bf16x2(v[2], v[3]);
                        }
                        unsigned sbw = 0;                // this lane's sign bytes of the 32-channel row: byte (k) at bit 16 k
#pragma unroll
                        for (int k = 0; k < (CO16 ? 1 : 2); ++k) {
                            uint2 lo = o2[2 * k], up = o2[2 * k + 1];
                            auto rx = __builtin_amdgcn_permlane32_swap(lo.x, up.x, false, false);
                            auto ry = __builtin_amdgcn_permlane32_swap(lo.y, up.y, false, false);
                            uint4 val = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                            if (inimg) {
                                const size_t doff = pix + m * 32 + 16 * k;
                                if (GEO == C2_S && a.mask) val = lrelu_mask_bf16x8(val, *reinterpret_cast<const uint4*>(a.mask + doff));
                                if ((GEO == C2_S || GEO == C2_D) && a.signbits) {
                                    const unsigned wv[4] = {val.x, val.y, val.z, val.w};
                                    unsigned bits = 0;
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        bits |= ((short)(wv[q] & 0xffffu) > 0 ? 1u : 0u) << (2 * q);
                                        bits |= ((short)(wv[q] >> 16) > 0 ? 1u : 0u) << (2 * q + 1);
                                    }
                                    sbw |= bits << (16 * k);
                                }
                                if (GEO == C2_D && (a.fade_resid || a.fade_pimg)) {
                                    const float fade_a = a.fade_ab ? a.fade_ab[0] : a.fade_alpha, fade_b = a.fade_ab ? a.fade_ab[1] : a.fade_beta;
                                    const uint4 rq = a.fade_pimg ? fade_resid_from_image(ecoef, pim[f][0], pim[f][1], pim[f][2], m * 32 + 16 * k + 8 * hi)
                                                                 : *reinterpret_cast<const uint4*>(a.fade_resid + doff);
                                    const unsigned yv[4] = {val.x, val.y, val.z, val.w}, rv[4] = {rq.x, rq.y, rq.z, rq.w};
                                    unsigned ov[4];
#pragma unroll
                                    for (int q = 0; q < 4; ++q)
                                        ov[q] = pack_bf16x2(fade_a * __uint_as_float(yv[q] << 16) + fade_b * __uint_as_float(rv[q] << 16),
                                                            fade_a * __uint_as_float(yv[q] & 0xffff0000u) + fade_b * __uint_as_float(rv[q] & 0xffff0000u));
                                    val = make_uint4(ov[0], ov[1], ov[2], ov[3]);
                                }
                                st16(a.y + doff, val, a.nt & 1);
                            }
                        }
                        if ((GEO == C2_S || GEO == C2_D) && a.signbits) {
                            // the four sign bytes of a pixel's 32-channel row (bytes 2k + hi) leave as ONE aligned word from the lower half-wave
                            // (128 contiguous bytes per 32 pixels) instead of four scattered byte stores: the upper half-wave's two bytes come over
                            // with one v_permlane32_swap (16 channels: two bytes, one 16-bit store)
                            auto sw = __builtin_amdgcn_permlane32_swap(sbw, sbw, false, false);
                            if (inimg && hi == 0) {
                                const unsigned word = sw[0] | (sw[1] << 8);
                                const size_t bidx = (pix + m * 32) >> 3;
                                if constexpr (CO16) *reinterpret_cast<unsigned short*>(a.signbits + bidx) = (unsigned short)word;
                                else *reinterpret_cast<unsigned*>(a.signbits + bidx) = word;
                            }
                        }
                    }
                }
            } else {
                float4 bv[MF][4];
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        bv[m][g] = (a.bias && !(CO16 && g >= 2)) ? *reinterpret_cast<const float4*>(a.bias + co0 + m * 32 + 8 * g + 4 * hi)
                                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                __syncthreads();                 // every wave is done reading this stage's patch
                if constexpr (EPI == EPI_STATS) {
                    if (b != st_b) {             // the loop moved on to another image (block-uniform)
                        if (st_b >= 0) stats_flush(cur);
                        st_b = b;
                    }
                }
                char* scr = cur + wave * (32 * OROW);
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int m = 0; m < MF; ++m) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int ch = m * 32 + 8 * g + 4 * hi;
                            float v[4] = {acc[m][f][4 * g], acc[m][f][4 * g + 1], acc[m][f][4 * g + 2], acc[m][f][4 * g + 3]};
                            v[0] += bv[m][g].x; v[1] += bv[m][g].y; v[2] += bv[m][g].z; v[3] += bv[m][g].w;
                            if (a.act == SGX_ACT_LRELU) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = lrelu(v[i]);
                            }
                            uint2 o;
                            o.x = pack_bf16x2(v[0], v[1]);
                            o.y = pack_bf16x2(v[2], v[3]);
                            *reinterpret_cast<uint2*>(scr + l31 * OROW + ch * 2) = o;
                        }
                    }
                    const int oy = ty0 + 2 * wave + f;
#pragma unroll
                    for (int i = 0; i < 32 * VPR / 64; ++i) {
                        const int idx = i * 64 + lane, px = idx / VPR, v = idx % VPR;
                        uint4 val = *reinterpret_cast<const uint4*>(scr + px * OROW + v * 16);
                        const int ox = tx0 + px;
                        if (oy < a.OH && ox < a.OW) {
                            const size_t doff = (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + v * 8;
                            if (GEO == C2_S && a.mask) val = lrelu_mask_bf16x8(val, *reinterpret_cast<const uint4*>(a.mask + doff));
                            st16(a.y + doff, val, a.nt & 1);
                            if constexpr (EPI == EPI_STATS) {
                                // (64 % VPR == 0: v = lane % VPR is the same channel vector in every iteration)
                                const float nz = nzv[f][i];
                                const unsigned wv[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float x0 = __uint_as_float(wv[q] << 16), x1 = __uint_as_float(wv[q] & 0xffff0000u);
                                    const float a0 = lrelu(x0 + ecoef[v * 8 + 2 * q] + ecoef[BCO + v * 8 + 2 * q] * nz);
                                    const float a1 = lrelu(x1 + ecoef[v * 8 + 2 * q + 1] + ecoef[BCO + v * 8 + 2 * q + 1] * nz);
                                    st0[2 * q] += a0; st1[2 * q] += a0 * a0;
                                    st0[2 * q + 1] += a1; st1[2 * q + 1] += a1 * a1;
                                }
                            }
                        }
                    }
                }
            }
            zero_acc();
        }
    }
    if constexpr (EPI == EPI_STATS) {
        __syncthreads();                         // (every wave is past its last scratch read)
        if (st_b >= 0) stats_flush(smem);
    }
}

static int conv2_ncu() { return sgx_ncu(); }

// =====================================================================================================================================
// conv3_kernel (round 5): the same convolution, the same LDS image, the same accumulation order (bit-identical results) -- with the
// LDS -> MFMA pipeline written by hand.  What the round-4 counters and the disassembly of conv2_kernel<S,8,2> say about its K-step:
//   * hipcc re-serialises the software-pipelined fragment reads of the source: every batch of ds_read_b128 is followed by
//     s_waitcnt lgkmcnt(0) and then by the 2-4 MFMAs that need it (it reuses the fragment registers at once), so a wave alternates
//     "issue 4 MFMAs (128 cycles of pipe time) -> issue reads -> wait ~100-200 cycles of LDS latency" and leaves the matrix pipe idle
//     a third of the time; two waves per SIMD in lockstep (one barrier per K-step) only partly fill each other's gaps: MFMA busy 0.38
//     of the 2.4 GHz peak (0.52 of the clock the chip really runs this kernel at);
//   * ~90 scalar instructions of integer division per K-step (tile index -> image, row, column, twice) sit between the barrier
//     and the first fragment read of BOTH waves of a SIMD.
// Here: fragment reads are inline-asm ds_read_b128 into explicitly double / triple buffered registers, issued PD sub-steps (one
// sub-step = the MF x RPW MFMAs of one tap and 16 channels) ahead of their use, with COUNTED s_waitcnt lgkmcnt(N) (never 0 inside
// a K-step) tied to the fragment registers by "+v" operands, so that neither the compiler's scheduler nor its waitcnt insertion can
// undo the pipeline; tile coordinates are decoded once per tile, not per K-step; RPW = 4 pixel rows per wave (128 accumulators, one
// wave per SIMD in the 4-wave block: the same 16 x 32 pixel tile, 40 % fewer fragment reads per MFMA) is a template parameter next to
// the 2-row / 8-wave shape; DIL interleaves the next stage's LDS-DMA instructions with the sub-steps instead of issuing them all
// behind the barrier.  Geometries S and D, 32-channel K-steps, the register epilogue with all its options (mask, sign bits, fade).
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int OFF> __device__ __forceinline__ void dsr128(i32x4& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read immediate offset");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// s_waitcnt lgkmcnt(N) that the registers it makes valid pass THROUGH: an MFMA that reads them cannot be scheduled above it
template <int N> __device__ __forceinline__ void lgkm_wait(i32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void lgkm_wait(i32x4& a, i32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N> __device__ __forceinline__ void lgkm_wait(i32x4& a, i32x4& b, i32x4& c, i32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int GEO, int NW, int MF, int RPW> struct C3Lds {
    static constexpr int TH = RPW * NW, PH = TH + G2<GEO>::HALO, PW = 32 + G2<GEO>::HALO, BCO = MF * 32;
    static constexpr int PROWS = PH * PW, WROWS = G2<GEO>::NTW * BCO;
    static constexpr int P_INSTR = (PROWS * 4 + 63) / 64, P_BYTES = P_INSTR * 1024;
    static constexpr int W_INSTR = WROWS * 4 / 64, W_BYTES = W_INSTR * 1024;
    static_assert(WROWS * 4 % 64 == 0, "weight stage: whole DMA instructions");
    static constexpr int STAGE = P_BYTES + W_BYTES, TOTAL = 2 * STAGE;
};

// Issue-order bookkeeping of one K-step's fragment reads (all compile time).  Prologue: brow(group 0), af(0) .. af(PD-1).  Sub-step
// s then issues af(s + PD) and, on the first sub-step of a group g, brow(g + 1), BEFORE it waits for its own operands.
template <int NG, int NDY, int MF, int R, int PD> struct C3Seq {
    static constexpr int NSUB = NG * NDY;
    static constexpr int issued_through(int s) {            // reads issued up to and including sub-step s's issue phase (s = -1: the prologue)
        int n = R + MF * (PD < NSUB ? PD : NSUB);
        for (int t = 0; t <= s; ++t) {
            if (t + PD < NSUB) n += MF;
            if (t % NDY == 0 && t / NDY + 1 < NG) n += R;
        }
        return n;
    }
    static constexpr int last_af(int s) {                   // 1-based issue index of the last read of af(s)
        if (s < PD) return R + MF * (s + 1);
        const int t = s - PD;                               // issued in sub-step t, first thing
        return issued_through(t - 1) + MF;
    }
    static constexpr int last_brow(int g) {
        if (g == 0) return R;
        const int t = (g - 1) * NDY;                        // issued in sub-step t, after its af
        return issued_through(t - 1) + (t + PD < NSUB ? MF : 0) + R;
    }
    static constexpr int wait_count(int s) {                // lgkmcnt to wait for before sub-step s's MFMAs
        const int need = (s % NDY == 0 && last_brow(s / NDY) > last_af(s)) ? last_brow(s / NDY) : last_af(s);
        const int k = issued_through(s) - need;
        return k > 15 ? 15 : k;
    }
};

template <int GEO, int NW, int MF, int RPW, int PD, int DIL, int UB = 0>
__global__ __launch_bounds__(NW * 64, UB == 2 ? 2 * NW / 4 : NW / 4) void conv3_kernel(Conv2Args a) {      // (UB == 2: two blocks per CU)
    using G = G2<GEO>;
    using L = C3Lds<GEO, NW, MF, RPW>;
    static_assert(GEO == C2_S || GEO == C2_D, "3x3 and stride-2 geometries");
    constexpr int TH = L::TH, PW = L::PW, BCO = L::BCO, PROWS = L::PROWS;
    constexpr int P_INSTR = L::P_INSTR, P_BYTES = L::P_BYTES, W_INSTR = L::W_INSTR, STAGE = L::STAGE;
    constexpr int NPI = (P_INSTR + NW - 1) / NW, NWI = (W_INSTR + NW - 1) / NW;
    constexpr int NPH = G::NPH, IS = G::IS, NDX = G::NDX, NDY = G::NDY;
    constexpr int KS = 2, NG = NDX * KS, NSUB = NG * NDY, R = RPW + NDY - 1, NAB = PD + 1;
    using SQ = C3Seq<NG, NDY, MF, R, PD>;
    static_assert(MF == 1 || MF == 2, "one or two 32-channel accumulator rows");
    static_assert(R == 3 || R == 4 || R == 5 || R == 6, "patch rows per group");
    // UB: the transposed 4x4 stride-2 convolution AND the [1,2,1]x[1,2,1] blur behind it as ONE 3x3 stride-1 convolution over the
    // coarse grid to 4 x 16 "virtual" channels n = (px * 2 + py) * 16 + co (sgx_pack_upblur composes the weights: blur o convT is a
    // 6x6 stride-2 transposed kernel = 3x3 taps per output parity class), stored depth-to-space: virtual channel (py, px, co) of
    // coarse pixel (i, j) is channel co of fine pixel (2i + py, 2j + px).  The blur zero-pads the FINE grid, so on the image's first /
    // last fine row and column the composite over-counts the transposed convolution's virtual outputs just outside the image; those
    // terms are linear in one input row / column and are taken out again by a few extra MFMAs with correction taps (a.wcorr) in the
    // tiles that touch the border (see the border block below).  Cin = 32 (one K-step per tile), Cout = 16.
    static_assert(!UB || (GEO == C2_S && MF == 2), "UB: the 3x3 geometry to 64 virtual channels");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, xcd = bid & 7, j8 = bid >> 3;
    const int cb = j8 % a.ncb, slot = (j8 / a.ncb) * 8 + xcd;
    const int co0 = cb * BCO;
    const int band = (a.ntiles + 7) >> 3, per = a.nslots >> 3, lslot = j8 / a.ncb;
    const int band_len = a.ntiles - xcd * band < band ? a.ntiles - xcd * band : band;
    const int tile0 = a.bands ? xcd * band + lslot : slot, tstride = a.bands ? per : a.nslots;
    const int my_tiles = a.bands ? (lslot < band_len ? (band_len - lslot + per - 1) / per : 0)
                                 : (slot < a.ntiles ? (a.ntiles - slot + a.nslots - 1) / a.nslots : 0);
    if (my_tiles <= 0) return;
    const int nchunks = a.Cin / 32;
    const int spt = nchunks * NPH;
    const int nsteps = my_tiles * spt;

    // ---- per-lane DMA descriptors (tile independent), as conv2_kernel
    int prel[NPI], ppos[NPI], wrel[NWI];
#pragma unroll
    for (int jj = 0; jj < NPI; ++jj) {
        const int s = (jj * NW + wave) * 64 + lane;
        const int row = s >> 2, c = s & 3;
        const int pr = row / PW, pc = row % PW;
        prel[jj] = (IS * pr * a.W + IS * pc) * a.Cin + ((c ^ ((pc >> 2) & 3)) << 3);
        ppos[jj] = row < PROWS ? ((pr << 8) | pc) : -1;
    }
#pragma unroll
    for (int jj = 0; jj < NWI; ++jj) {
        const int s = (jj * NW + wave) * 64 + lane;
        const int row = s >> 2, c = s & 3;
        const int t = row / BCO, n = row % BCO;
        const int tg0 = GEO == C2_D ? (2 * (t >> 1) + 1) * 4 + 2 * (t & 1) + 1 : t;
        wrel[jj] = ((tg0 * a.Cout + co0 + n) * a.Cin) + ((c ^ ((n >> 2) & 3)) << 3);
    }
    // ---- per-lane fragment read addresses inside a stage: one base per (column shift, k half), rows and taps by immediate offset
    unsigned pbase[NG], wbase[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        wbase[ks] = P_BYTES + l31 * 64 + (((hi + 2 * ks) ^ ((l31 >> 2) & 3)) << 4);
#pragma unroll
        for (int dx = 0; dx < NDX; ++dx) {
            const int pc = l31 + dx;
            pbase[dx * KS + ks] = (RPW * wave * PW + pc) * 64 + (((hi + 2 * ks) ^ ((pc >> 2) & 3)) << 4);
        }
    }
    const bf16_t* __restrict__ xg = a.x;
    const bf16_t* __restrict__ wg = a.w;
    const unsigned long long zaddr = reinterpret_cast<unsigned long long>(sgx_zero_page) + (lane & 3) * 16;

    auto tile_coords = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx_i = t % a.tiles_x; t /= a.tiles_x;
        const int ty_i = t % a.tiles_y;
        b = t / a.tiles_y; ty0 = ty_i * TH; tx0 = tx_i * 32;
    };
    // ---- the ISSUE cursor: the K-step whose operands are staged next -- its tile decoded once per tile
    int i_t = tile0, i_q = 0, i_n = 0, ib, ity0, itx0;
    tile_coords(i_t, ib, ity0, itx0);
    // one DMA instruction of the issue cursor's K-step: piece j < NPI = patch, else weights
    auto dma_piece = [&](auto J, char* buf) {
        constexpr int j = decltype(J)::value;
        const int kc = i_q / NPH, ph = i_q - kc * NPH, py = ph >> 1, px = ph & 1;
        if constexpr (j < NPI) {
            const int ii = j * NW + wave;
            // ablations (a.dbg, probe only): 4 = no DMA at all after the first two stages, 8 = no patch DMA on odd K-steps (what staging a
            // patch once for two channel blocks would issue), 1 = patch pieces read 1 KB of CONTIGUOUS memory (8 full lines per instruction)
            const bool skip = a.dbg && i_n >= 2 && ((a.dbg & 4) || ((a.dbg & 8) && (i_n & 1)));
            if (ii < P_INSTR && !skip) {
                const int iy0 = GEO == C2_D ? 2 * ity0 - py : ity0 - 1, ix0 = GEO == C2_D ? 2 * itx0 - px : itx0 - 1;
                const bf16_t* base = xg + (((long)ib * a.H + iy0) * a.W + ix0) * a.Cin + kc * 32;
                const int pp = ppos[j];
                const int gy = iy0 + IS * (pp >> 8), gx = ix0 + IS * (pp & 255);
                const unsigned long long ok = ((pp >= 0) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W)) ? ~0ull : 0ull;
                unsigned long long pa = reinterpret_cast<unsigned long long>(base + prel[j]);
                if (a.dbg & 1) pa = reinterpret_cast<unsigned long long>(xg + (((long)ib * a.H + (ity0 > 0 ? ity0 : 0)) * a.W) * a.Cin + (ii * 64 + lane) * 8);
                if (a.nt & 2) glds16_nt(reinterpret_cast<const void*>(zaddr + ((pa - zaddr) & ok)), buf + ii * 1024);
                else glds16(reinterpret_cast<const void*>(zaddr + ((pa - zaddr) & ok)), buf + ii * 1024);
            }
        } else {
            constexpr int jw = j - NPI;
            const int ii = jw * NW + wave;
            // <= 2 K-steps per tile: step q's weights live in stage q for the whole launch
            // (ablations: 4 / 16 = no weight DMA after the first two stages, 2 = weight pieces read 1 KB of contiguous memory -- what a
            // (K-chunk, tap)-contiguous weight pack would make of them)
            if (ii < W_INSTR && (UB ? i_n == 0 : (spt > 2 || i_n < 2)) && !((a.dbg & 20) && i_n >= 2)) {
                const bf16_t* w0 = wg + kc * 32 - (GEO == C2_D ? (long)(4 * py + px) * a.Cout * a.Cin : 0);
                glds16((a.dbg & 2) ? wg + ((size_t)(cb * nchunks + kc) * NPH + ph) * (W_INSTR * 512) + (ii * 64 + lane) * 8 : w0 + wrel[jw], buf + P_BYTES + ii * 1024);
            }
        }
    };
    auto advance_issue = [&]() {
        ++i_n;
        if (++i_q == spt) {
            i_q = 0; i_t += tstride;
            if (i_n < nsteps) tile_coords(i_t, ib, ity0, itx0);
        }
    };

    f32x16 acc[MF][RPW];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int f = 0; f < RPW; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][f][r] = 0.f;
    };
    zero_acc();

    static_for<0, NPI + NWI>([&](auto J) { dma_piece(J, smem); });
    // UB == 2 (round 6): the COMPACT image -- [patch 0][composite taps][patch 1], 80 KB with the 4-wave block's 10 x 34 patch -- for TWO blocks
    // per CU: they share nothing and drift apart, so one block's depth-to-space store runs under the other's MFMAs (the 8-wave block's
    // waves all store at once behind one barrier: 3.4 TB/s, 0.27 of the MFMA pipe).  The ten LDS-resident correction tiles are read from
    // global memory instead (L2 hits; one tile column in sixteen needs them at 512^2).
    if constexpr (UB == 1) {
        // the LDS-resident correction tiles (first / last fine column, corners: 10 tiles of 32 rows x 64 bytes) -> the weight region of
        // stage 1 (its own copy of the composite taps is not needed: one K-step per tile, the taps are read from stage 0), once
#pragma unroll
        for (int jj = 0; jj < (20 + NW - 1) / NW; ++jj) {
            const int ii = jj * NW + wave;
            if (ii < 20) {
                const int s = ii * 64 + lane, row = s >> 2, c = s & 3, n = row & 31;
                glds16(a.wcorr + (size_t)row * 32 + ((c ^ ((n >> 2) & 3)) << 3), smem + STAGE + P_BYTES + ii * 1024);
            }
        }
    }
    advance_issue();
    int c_t = tile0, c_q = 0;
    unsigned mwp[RPW][2];                            // UB: sign-bit words of the tile being computed (c_t), see mwn in the loop
    if constexpr (UB) {
        int b_, ty_, tx_;
        tile_coords(tile0, b_, ty_, tx_);
#pragma unroll
        for (int f = 0; f < RPW; ++f)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int i = ty_ + RPW * wave + f, j = tx_ + l31;
                mwp[f][py] = (a.maskbits && i < a.H && j < a.W)
                                 ? *reinterpret_cast<const unsigned*>(a.maskbits + ((((size_t)b_ * a.OH + 2 * i + py) * a.OW) + 2 * j) * 2) : 0xffffffffu;
            }
    }
    for (int step = 0; step < nsteps; ++step) {
        const unsigned so = (step & 1) * STAGE;
        char* const nxt = smem + (STAGE - so);
        __syncthreads();                         // (vmcnt(0) first) stage `step` landed; everyone is done with step-1
        const bool more = step + 1 < nsteps;
        if constexpr (!DIL) {
            if (more) static_for<0, NPI + NWI>([&](auto J) { dma_piece(J, nxt); });
        }
        unsigned pb[NG], wb[KS];
#pragma unroll
        for (int g = 0; g < NG; ++g) pb[g] = pbase[g] + so;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wb[ks] = wbase[ks] + (UB ? 0u : so);   // (UB: the composite taps live in stage 0's weight region)
        // UB: the sign-bit words of the NEXT tile (the one whose patch is being staged: the issue cursor's) are requested now; they are
        // consumed one K-step later, behind the barrier that also waits for that stage -- never a wait of their own in the epilogue.
        // (Measured and dropped: requesting the current tile's words here and waiting for them in the epilogue with a counted vmcnt --
        // LDS-DMA loads and ordinary loads do not retire in one order: the words were stale in every run, tools/repro_check.py.)
        unsigned mwn[RPW][2];
        if constexpr (UB) {
#pragma unroll
            for (int f = 0; f < RPW; ++f)
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int i = ity0 + RPW * wave + f, j = itx0 + l31;
                    mwn[f][py] = (a.maskbits && more && i < a.H && j < a.W)
                                     ? *reinterpret_cast<const unsigned*>(a.maskbits + ((((size_t)ib * a.OH + 2 * i + py) * a.OW) + 2 * j) * 2) : 0xffffffffu;
                }
        }
        i32x4 brow[2][R], af[NAB][MF];
        auto ld_brow = [&](auto Gi) {
            constexpr int g = decltype(Gi)::value;
            static_for<0, R>([&](auto Ri) { constexpr int r = decltype(Ri)::value; dsr128<r * PW * 64>(brow[g & 1][r], pb[g]); });
        };
        auto ld_af = [&](auto Si) {
            constexpr int sub = decltype(Si)::value, g = sub / NDY, dy = sub % NDY, dx = g / KS, ks = g % KS;
            static_for<0, MF>([&](auto Mi) { constexpr int m = decltype(Mi)::value; dsr128<((dy * NDX + dx) * BCO + m * 32) * 64>(af[sub % NAB][m], wb[ks]); });
        };
        ld_brow(std::integral_constant<int, 0>{});
        static_for<0, (PD < NSUB ? PD : NSUB)>([&](auto Si) { ld_af(Si); });
        static_for<0, NSUB>([&](auto Si) {
            constexpr int sub = decltype(Si)::value, g = sub / NDY, dy = sub % NDY;
            if constexpr (sub + PD < NSUB) ld_af(std::integral_constant<int, sub + PD>{});
            if constexpr (dy == 0 && g + 1 < NG) ld_brow(std::integral_constant<int, g + 1>{});
            if constexpr (DIL) {
                // the next stage's DMA instructions, spread over the first sub-steps (they must have landed by the next barrier)
                constexpr int per_sub = (NPI + NWI + NSUB - 3) / (NSUB - 2), j0 = sub * per_sub;
                if (more) static_for<j0, (j0 + per_sub < NPI + NWI ? j0 + per_sub : NPI + NWI)>([&](auto J) { dma_piece(J, nxt); });
            }
            constexpr int K = SQ::wait_count(sub);
            i32x4(&A)[MF] = af[sub % NAB];
            i32x4(&Bv)[R] = brow[g & 1];
            if constexpr (dy == 0) {
                // first use of this group's patch rows: everything this sub-step reads passes through the wait
                if constexpr (MF == 2) lgkm_wait<K>(A[0], A[1]); else lgkm_wait<K>(A[0]);
                if constexpr (R == 3) { lgkm_wait<K>(Bv[0], Bv[1]); lgkm_wait<K>(Bv[2]); }
                else if constexpr (R == 4) lgkm_wait<K>(Bv[0], Bv[1], Bv[2], Bv[3]);
                else if constexpr (R == 5) { lgkm_wait<K>(Bv[0], Bv[1], Bv[2], Bv[3]); lgkm_wait<K>(Bv[4]); }
                else { lgkm_wait<K>(Bv[0], Bv[1], Bv[2], Bv[3]); lgkm_wait<K>(Bv[4], Bv[5]); }
            } else {
                if constexpr (MF == 2) lgkm_wait<K>(A[0], A[1]); else lgkm_wait<K>(A[0]);
            }
            static_for<0, MF>([&](auto Mi) {
                constexpr int m = decltype(Mi)::value;
                static_for<0, RPW>([&](auto Fi) {
                    constexpr int f = decltype(Fi)::value;
                    acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[m]), __builtin_bit_cast(bf16x8, Bv[f + dy]), acc[m][f], 0, 0, 0);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if (more) advance_issue();
        if constexpr (UB) {
            // (one K-step per tile)  ---- border corrections, then the depth-to-space store
            int b, ty0, tx0;
            tile_coords(c_t, b, ty0, tx0);
            c_t += tstride;
            const int row0 = ty0 + RPW * wave;                          // this wave's coarse rows row0 .. row0 + RPW - 1
            const bool left = tx0 == 0, right = tx0 + 32 >= a.W;        // (a.W % 32 == 0: the last column is lane 31 of the last tile column)
            const bool rowb = row0 == 0 || row0 + RPW >= a.H;
            if (left || right || rowb) {
                // B fragment of patch row pr, column shift dx, k half ks of the stage just consumed (valid until the next barrier)
                auto patch_b = [&](int pr, int dx, int ks) {
                    const int pc = l31 + dx;
                    return *reinterpret_cast<const i32x4*>(smem + so + ((RPW * wave + pr) * PW + pc) * 64 + (((hi + 2 * ks) ^ ((pc >> 2) & 3)) << 4));
                };
                // A fragment of LDS-resident correction tile t (32 rows (py, co) x 32 channels; the weight region of stage 1)
                auto corr_lds = [&](int t, int ks) {
                    if constexpr (UB == 2) return *reinterpret_cast<const i32x4*>(a.wcorr + ((size_t)t * 32 + l31) * 32 + ks * 16 + hi * 8);
                    else return *reinterpret_cast<const i32x4*>(smem + STAGE + P_BYTES + (t * 32 + l31) * 64 + (((hi + 2 * ks) ^ ((l31 >> 2) & 3)) << 4));
                };
                // ... and of the first / last fine ROW's correction tiles, read from global memory (one wave in the first / last tile row only)
                auto corr_glb = [&](int t, int ks) {
                    return *reinterpret_cast<const i32x4*>(a.wcorr + ((size_t)(10 + t) * 32 + l31) * 32 + ks * 16 + hi * 8);
                };
                const i32x4 zero4 = {0, 0, 0, 0};
#pragma unroll
                for (int f = 0; f < RPW; ++f) {
                    const int r = row0 + f;
                    if (r >= a.H) continue;
                    const bool top = r == 0, bot = r == a.H - 1;
                    if (left || right) {
                        // first / last fine column: the column just outside the image contributed k0 (k2) * convT(column -1 (2W)), which only
                        // input column 0 (W - 1) reaches, through kernel column 0 (3).  Only the border pixel's lane contributes (the others see
                        // a zero B operand); tiles Cl = 0..2, Cr = 3..5 (negated in the pack) go to the px = 0 / px = 1 accumulators
#pragma unroll 1
                        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll 1
                            for (int dy = 0; dy < 3; ++dy) {
                                const i32x4 bfull = patch_b(f + dy, 1, ks);
                                if (left) acc[0][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, corr_lds(dy, ks)), __builtin_bit_cast(bf16x8, l31 == 0 ? bfull : zero4), acc[0][f], 0, 0, 0);
                                if (right) acc[1][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, corr_lds(3 + dy, ks)), __builtin_bit_cast(bf16x8, l31 == 31 ? bfull : zero4), acc[1][f], 0, 0, 0);
                            }
                            // corners: the row AND the column term both took the (row, column) cross term out: one goes back in (tiles 6..9 = TL TR BL BR)
                            if (top || bot) {
                                const i32x4 bc = patch_b(f + 1, 1, ks);
                                if (top && left) acc[0][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, corr_lds(6, ks)), __builtin_bit_cast(bf16x8, l31 == 0 ? bc : zero4), acc[0][f], 0, 0, 0);
                                if (top && right) acc[1][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, corr_lds(7, ks)), __builtin_bit_cast(bf16x8, l31 == 31 ? bc : zero4), acc[1][f], 0, 0, 0);
                                if (bot && left) acc[0][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, corr_lds(8, ks)), __builtin_bit_cast(bf16x8, l31 == 0 ? bc : zero4), acc[0][f], 0, 0, 0);
                                if (bot && right) acc[1][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, corr_lds(9, ks)), __builtin_bit_cast(bf16x8, l31 == 31 ? bc : zero4), acc[1][f], 0, 0, 0);
                            }
                        }
                    }
                    if (top || bot) {
                        // first / last fine row: the same for input row 0 (H - 1) and kernel row 0 (3), on the wave's own row, all lanes; global
                        // tiles (dx * 2 + px) for the first row, 6 + (dx * 2 + px) for the last.  The six fragments of a k half are requested together.
#pragma unroll 1
                        for (int ks = 0; ks < KS; ++ks) {
                            if (top) {
                                i32x4 at[3][2];
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                                    for (int m = 0; m < 2; ++m) at[dx][m] = corr_glb(dx * 2 + m, ks);
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx) {
                                    const i32x4 bf = patch_b(f + 1, dx, ks);
#pragma unroll
                                    for (int m = 0; m < 2; ++m)
                                        acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, at[dx][m]), __builtin_bit_cast(bf16x8, bf), acc[m][f], 0, 0, 0);
                                }
                            }
                            if (bot) {
                                i32x4 ab[3][2];
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                                    for (int m = 0; m < 2; ++m) ab[dx][m] = corr_glb(6 + dx * 2 + m, ks);
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx) {
                                    const i32x4 bf = patch_b(f + 1, dx, ks);
#pragma unroll
                                    for (int m = 0; m < 2; ++m)
                                        acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ab[dx][m]), __builtin_bit_cast(bf16x8, bf), acc[m][f], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
            // ---- depth-to-space store.  acc[m = px][f]: lane column l31 = coarse column j, registers 4g .. 4g + 3 = virtual channels
            // 8g + 4hi .. + 3 of the M-tile, i.e. py = g >> 1, co = 8 (g & 1) + 4 hi + r.  One permlane32_swap pair per py hands a lane
            // 8 consecutive co of fine pixel (2i + py, 2j + px): lanes < 32 co 0-7, lanes >= 32 co 8-15 -- one 16-byte store.
            // Mask (the LeakyReLU backward of the discriminator's chain): sign bits [fine pixel][2 bytes]; the 4 bytes of a lane's two fine
            // pixels (px = 0, 1) of each fine row were requested before the K-step's MFMAs (mwp); the slope multiplies in fp32 BEFORE
            // the one rounding to bf16.
#pragma unroll
            for (int f = 0; f < RPW; ++f) {
                const int i = row0 + f, j = tx0 + l31;
                const bool inimg = i < a.H && j < a.W;
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    uint2 o2[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4] = {acc[px][f][4 * g], acc[px][f][4 * g + 1], acc[px][f][4 * g + 2], acc[px][f][4 * g + 3]};
                        if (a.maskbits) {
                            const unsigned nib = (mwp[f][g >> 1] >> (16 * px + 8 * (g & 1) + 4 * hi)) & 15u;     // byte (px, co >> 3), bits 4hi .. 4hi + 3
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] *= ((nib >> r) & 1u) ? 1.f : SGX_LRELU;
                        }
                        o2[g].x = pack_bf16x2(v[0], v[1]);
                        o2[g].y = pack_bf16x2(v[2], v[3]);
                    }
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        uint2 lo = o2[2 * py], up = o2[2 * py + 1];
                        auto rx = __builtin_amdgcn_permlane32_swap(lo.x, up.x, false, false);
                        auto ry = __builtin_amdgcn_permlane32_swap(lo.y, up.y, false, false);
                        if (inimg) st16(a.y + ((((size_t)b * a.OH + 2 * i + py) * a.OW) + 2 * j + px) * 16 + 8 * hi, make_uint4(rx[0], ry[0], rx[1], ry[1]), a.nt & 1);
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < RPW; ++f)
#pragma unroll
                for (int py = 0; py < 2; ++py) mwp[f][py] = mwn[f][py];
            zero_acc();
        } else
        if (++c_q == spt) {
            c_q = 0;
            // ---- register epilogue (conv2_kernel's, for RPW rows): bias, activation, bf16, v_permlane32_swap pairing -> 16-byte stores
            int b, ty0, tx0;
            tile_coords(c_t, b, ty0, tx0);
            c_t += tstride;
            float4 bv[MF][4];
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    bv[m][g] = a.bias ? *reinterpret_cast<const float4*>(a.bias + co0 + m * 32 + 8 * g + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int f = 0; f < RPW; ++f) {
                const int oy = ty0 + RPW * wave + f, ox = tx0 + l31;
                const bool inimg = oy < a.OH && ox < a.OW;
                const size_t pix = (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + 8 * hi;
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    uint2 o2[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4] = {acc[m][f][4 * g], acc[m][f][4 * g + 1], acc[m][f][4 * g + 2], acc[m][f][4 * g + 3]};
                        v[0] += bv[m][g].x; v[1] += bv[m][g].y; v[2] += bv[m][g].z; v[3] += bv[m][g].w;
                        if (a.act == SGX_ACT_LRELU) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = lrelu(v[i]);
                        }
                        o2[g].x = pack_bf16x2(v[0], v[1]);
                        o2[g].y = pack_bf16x2(v[2], v[3]);
                    }
                    unsigned sbw = 0;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        uint2 lo = o2[2 * k], up = o2[2 * k + 1];
                        auto rx = __builtin_amdgcn_permlane32_swap(lo.x, up.x, false, false);
                        auto ry = __builtin_amdgcn_permlane32_swap(lo.y, up.y, false, false);
                        uint4 val = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                        if (inimg) {
                            const size_t doff = pix + m * 32 + 16 * k;
                            if (GEO == C2_S && a.mask) val = lrelu_mask_bf16x8(val, *reinterpret_cast<const uint4*>(a.mask + doff));
                            if (a.signbits) {
                                const unsigned wv[4] = {val.x, val.y, val.z, val.w};
                                unsigned bits = 0;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    bits |= ((short)(wv[q] & 0xffffu) > 0 ? 1u : 0u) << (2 * q);
                                    bits |= ((short)(wv[q] >> 16) > 0 ? 1u : 0u) << (2 * q + 1);
                                }
                                sbw |= bits << (16 * k);
                            }
                            if (GEO == C2_D && a.fade_resid) {
                                const float fade_a = a.fade_ab ? a.fade_ab[0] : a.fade_alpha, fade_b = a.fade_ab ? a.fade_ab[1] : a.fade_beta;
                                const uint4 rq = *reinterpret_cast<const uint4*>(a.fade_resid + doff);
                                const unsigned yv[4] = {val.x, val.y, val.z, val.w}, rv[4] = {rq.x, rq.y, rq.z, rq.w};
                                unsigned ov[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    ov[q] = pack_bf16x2(fade_a * __uint_as_float(yv[q] << 16) + fade_b * __uint_as_float(rv[q] << 16),
                                                        fade_a * __uint_as_float(yv[q] & 0xffff0000u) + fade_b * __uint_as_float(rv[q] & 0xffff0000u));
                                val = make_uint4(ov[0], ov[1], ov[2], ov[3]);
                            }
                            st16(a.y + doff, val, a.nt & 1);
                        }
                    }
                    if (a.signbits) {                 // (one aligned word per pixel and 32 channels: conv2_kernel's epilogue)
                        auto sw = __builtin_amdgcn_permlane32_swap(sbw, sbw, false, false);
                        if (inimg && hi == 0) *reinterpret_cast<unsigned*>(a.signbits + ((pix + m * 32) >> 3)) = sw[0] | (sw[1] << 8);
                    }
                }
            }
            zero_acc();
        }
    }
}

template <int GEO, int NW, int MF, int RPW, int PD, int DIL, int UB = 0>
static int launch_conv3(Conv2Args& a, hipStream_t st) {
    using L = C3Lds<GEO, NW, MF, RPW>;
    constexpr int LDS = UB == 2 ? L::TOTAL - L::W_BYTES : L::TOTAL;      // (UB == 2: no second weight region)
    constexpr int BPC = UB == 2 ? (160 * 1024) / LDS : 1;                // resident blocks per CU the launch is sized for
    static_assert(LDS <= 160 * 1024 && BPC >= 1 && (UB != 2 || BPC == 2), "LDS budget");
    auto kern = conv3_kernel<GEO, NW, MF, RPW, PD, DIL, UB>;
    sgx_lds_opt_in<conv3_kernel<GEO, NW, MF, RPW, PD, DIL, UB>>(LDS);
    const int gh = GEO == C2_D ? a.OH : a.H, gw = GEO == C2_D ? a.OW : a.W;
    a.tiles_x = (gw + 31) / 32; a.tiles_y = (gh + L::TH - 1) / L::TH;
    a.ntiles = a.B * a.tiles_y * a.tiles_x;
    a.ncb = a.Cout / L::BCO;
    {
        static const int cnt = [] { const char* e = getenv("SGX_CONV_NT"); return e ? atoi(e) : 0; }();      // probe: bit 0 stores, bit 1 patch loads
        a.nt = sgx_nt_for(2.0 * a.B * a.OH * a.OW * (UB ? 16 : a.Cout)) ? cnt : 0;
    }
    int per = sgx_ncu() * BPC / (8 * a.ncb);
    const int need = (a.ntiles + 7) / 8;
    if (per > need) per = need;
    if (per < 1) per = 1;
    a.nslots = per * 8;
    static const int bands_on = [] { const char* e = getenv("SGX_TILE_BANDS"); return e ? atoi(e) : 1; }();
    a.bands = (bands_on && a.ntiles >= 8 * a.nslots) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * a.ncb * per)), dim3(NW * 64), LDS, st, a);
    SGX_LAUNCH_CHECK("conv3_kernel");
    return 0;
}

template <int GEO, int NW, int MF, int KC = 32, bool CO16 = false, int EPI = EPI_NONE>
static int launch_conv2(Conv2Args& a, hipStream_t st) {
    using L = C2Lds<GEO, NW, MF, KC>;
    constexpr int LDS = L::TOTAL + ((EPI != EPI_NONE || GEO == C2_D) ? 1024 : 0);      // + the epilogue's coefficient table (statistics; residual in the store)
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv2_kernel<GEO, NW, MF, KC, CO16, EPI>;
    sgx_lds_opt_in<conv2_kernel<GEO, NW, MF, KC, CO16, EPI>>(LDS);
    const int gh = GEO == C2_D ? a.OH : a.H, gw = GEO == C2_D ? a.OW : a.W;       // the tile grid
    a.tiles_x = (gw + 31) / 32; a.tiles_y = (gh + L::TH - 1) / L::TH;
    if constexpr (EPI == EPI_BLUR) {              // tiles overlap by one coarse row / column: steps TH - 1 and 31
        a.tiles_y = gh <= L::TH ? 1 : (gh - L::TH + L::TH - 2) / (L::TH - 1) + 1;
        a.tiles_x = gw <= 32 ? 1 : (gw - 32 + 30) / 31 + 1;
    }
    a.ntiles = a.B * a.tiles_y * a.tiles_x;
    a.ncb = CO16 ? 1 : a.Cout / L::BCO;
    {
        static const int cnt = [] { const char* e = getenv("SGX_CONV_NT"); return e ? atoi(e) : 0; }();      // probe: bit 0 stores, bit 1 patch loads
        a.nt = sgx_nt_for(2.0 * a.B * a.OH * a.OW * (CO16 ? 16 : a.Cout)) ? cnt : 0;
    }
    if constexpr (GEO == C2_U && EPI == EPI_NONE) {
        // A/B switch; measured on one box (round 6, default bench line, two interleaved rounds): register store 291.2 / 291.4 img/s at batch 4 and
        // 523.0 / 521.4 at batch 32, LDS-transposed store 292.8 / 293.9 and 525.0 / 525.4 -- whole 64-byte pixels per store instruction win
        static const int ureg = [] { const char* e = getenv("SGX_CONVU_REGSTORE"); return e ? atoi(e) : 0; }();
        a.ureg = ureg;
    }
    // resident blocks per CU: one for the round-2 instantiations (their LDS stages fill the CU); the half-width stages of the
    // 16-channel layers leave room for two (LDS <= 80 KB per block, <= 128 registers), so that one block's store epilogue
    // overlaps the other's loads
    int bpc = 1;
    if constexpr (KC == 16) {
        static const int bpc_max = [] { const char* e = getenv("SGX_CONV2_BPC"); return e ? atoi(e) : 2; }();   // A/B
        bpc = (160 * 1024) / LDS;
        if (bpc > 32 / NW) bpc = 32 / NW;
        if (bpc > bpc_max) bpc = bpc_max;
        if (bpc < 1) bpc = 1;
    }
    int per = conv2_ncu() * bpc / (8 * a.ncb);           // tile slots per XCD
    const int need = (a.ntiles + 7) / 8;
    if (per > need) per = need;
    if (per < 1) per = 1;
    a.nslots = per * 8;
    static const int bands_on = [] { const char* e = getenv("SGX_TILE_BANDS"); return e ? atoi(e) : 1; }();   // measured (round 2): halo over-fetch gone (PMC), 80.8 vs 81.2 ms at batch 32, nothing at batch 4
    a.bands = (bands_on && a.ntiles >= 8 * a.nslots) ? 1 : 0;      // enough tiles per slot for the order to matter
    if constexpr (EPI == EPI_STATS)
        SGX_REQUIRE(a.part_slots == 0 || a.part_slots == a.nslots, SGX_EINVAL, "conv2 statistics: partials sized for %d tile slots, the launch uses %d", a.part_slots, a.nslots);
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * a.ncb * per)), dim3(NW * 64), LDS, st, a);
    SGX_LAUNCH_CHECK("conv2_kernel");
    return 0;
}

// Which layers take this kernel, and with which block shape (SGX_CONV2=0 switches it off: A/B against conv.hip).
// geo: 0 = 3x3, 1 = 4x4 stride-2 down, 2 = 4x4 stride-2 up (H, W = input size).  ``variant``: -1 = choose (environment switch +
// heuristics), 4 / 8 = force the 4- / 8-wave block (A/B probes, tests).  nw == 0: the shape stays with the first generation.
struct Conv2Pick { int nw; bool mf2, k16; };
static Conv2Pick conv2_pick(int geo, int B, int H, int W, int Cin, int Cout, int variant, bool allow_u16 = false) {
    // bit 0: S, 1: D, 2: U.  All three on: profiles/r02_conv2_probe.txt (S) and r02_conv2_probe_DU.txt (D, U) -- the shape
    // heuristics below reproduce the per-shape winner of those tables.  bit 3: the 16-channel variants (round 3); bit 4: also the
    // transposed 32->16 one.
    static const int on = [] { const char* e = getenv("SGX_CONV2"); return e ? atoi(e) : 15; }();
    const Conv2Pick none{0, false, false};
    if (variant < 0 && !((on >> geo) & 1)) return none;
    const int gw = geo == C2_D ? W / 2 : W, gh = geo == C2_D ? H / 2 : H;
    const bool kc16 = Cin == 16, co16 = Cout == 16;
    if (kc16 || co16) {
        // the 16-channel layers: 3x3 16->16, stride-2 16->32, transposed 32->16 (and nothing else: the networks have no others)
        const bool shape_ok = (geo == C2_S && kc16 && co16) || (geo == C2_D && kc16 && Cout == 32) || (geo == C2_U && Cin == 32 && co16);
        if (!shape_ok || gw % 32 != 0 || gh < 1 || (geo == C2_D && ((H | W) & 1))) return none;
        if (variant < 0 && !(on & 8)) return none;
        // measured alone at batch 32 / 4 (tools/conv16_probe.py, profiles/r03_conv16_probe.txt): 3x3 16->16 @1024^2 first
        // generation 619 / 78 us, this kernel 498 / 59 us (4.3-4.5 TB/s); stride-2 16->32 478 / 65 -> 369 / 43 us; the
        // transposed 32->16 @512^2 stays with the first generation (338 vs 384 us: half of every MFMA is padding AND the
        // full-width stage leaves one block per CU) unless bit 4 of SGX_CONV2 asks for it
        if (variant < 0 && geo == C2_U && !(on & 16) && !allow_u16) return none;
        const long blocks4 = (long)B * ((gh + 7) / 8) * (gw / 32), blocks8 = (long)B * ((gh + 15) / 16) * (gw / 32);
        if (variant < 0 && blocks4 < conv2_ncu()) return none;
        static const int force16 = [] { const char* e = getenv("SGX_CONV2_NW16"); return e ? atoi(e) : 0; }();
        return Conv2Pick{variant > 0 ? variant : (force16 ? force16 : (blocks8 >= 2 * conv2_ncu() ? 8 : 4)), false, true};
    }
    const int bco = (geo == C2_S && Cout % 64 == 0) ? 64 : 32;       // (3x3 with 32 output channels: the MF = 1 block)
    if (Cin % 32 != 0 || Cout % bco != 0 || gw % 32 != 0 || Cout / bco > 32 || gh < 1 || (geo == C2_D && ((H | W) & 1))) return none;
    bool mf2 = geo != C2_U && Cout % 64 == 0;
    if (variant < 0 && geo == C2_D && !mf2) return none;   // 32-channel stride-2 blocks: measured no better than the first generation
    static const int force_nw = [] { const char* e = getenv("SGX_CONV2_NW"); return e ? atoi(e) : 0; }();
    if (geo == C2_S && mf2 && variant < 0 && !force_nw) {
        // small grids (batch 4 at 32^2..64^2): the 32-channel block doubles the block count.  Measured alone (round 6, tools/conv2_probe.py
        // variants 22 / 23, bit-identical outputs): batch 4, 64^2 256->256: (8 waves, 64 ch) 128 blocks 33.0 us, (4, 64) 256 blocks 29.3,
        // (8, 32) 256 blocks 24.0, (4, 32) 512 blocks 31.9; 32^2 512->512: first generation 36.4, (4, 64) 48.0, (8, 32) 37.3, (4, 32) 256
        // blocks 31.1; batch 8, 32^2: (4, 64) 52.4, (8, 32) 44.5, (4, 32) 58.2 -- the first shape in this order that fills the chip
        static const int small = [] { const char* e = getenv("SGX_CONV2_SMALL"); return e ? atoi(e) : 1; }();     // A/B: 0 = the round-2 rule
        const long t8 = (long)B * ((gh + 15) / 16) * (gw / 32), t4 = (long)B * ((gh + 7) / 8) * (gw / 32);
        const int c64 = Cout / 64, c32 = Cout / 32, ncu = conv2_ncu();
        if (small && t8 * c64 < ncu) {
            if (t8 * c32 >= ncu) return Conv2Pick{8, false, false};
            if (t4 * c64 >= ncu) return Conv2Pick{4, true, false};
            if (t4 * c32 >= ncu) return Conv2Pick{4, false, false};
            return none;
        }
    }
    if (geo == C2_D && mf2 && variant < 0 && !force_nw) {
        // the same for the stride-2 geometry, where only (8 waves, 32 channels) ever beat the round-2 choice: batch 4, 128^2 128->256:
        // (4, 64) 33.6 us, first generation 33.4, (8, 32) 28.5 (the other small-grid shapes: (8, 64) or the first generation stay best)
        static const int small = [] { const char* e = getenv("SGX_CONV2_SMALL"); return e ? atoi(e) : 1; }();
        const long t8 = (long)B * ((gh + 15) / 16) * (gw / 32);
        if (small && t8 * (Cout / 64) < conv2_ncu() && t8 * (Cout / 32) >= conv2_ncu()) return Conv2Pick{8, false, false};
    }
    const int cbs = Cout / (mf2 ? 64 : 32);
    const long blocks8 = (long)B * ((gh + 15) / 16) * (gw / 32) * cbs, blocks4 = (long)B * ((gh + 7) / 8) * (gw / 32) * cbs;
    // measured (profiles/r02_conv2_probe.txt): the 8-wave block wins once its 512-pixel tiles fill the chip, the 4-wave
    // block (256-pixel tiles) down to one block per CU, below that the first-generation kernel's 64-pixel tiles do
    if (variant < 0 && !force_nw && blocks4 < conv2_ncu()) return none;
    return Conv2Pick{variant > 0 ? variant : (force_nw ? force_nw : (blocks8 >= conv2_ncu() ? 8 : 4)), mf2, false};
}

// conv3_kernel by configuration id (sgx_conv_variant 30 + id): 0: 8 waves x 2 rows, fragments one sub-step ahead; 1: two ahead;
// 2: two ahead + DMA interleaved; 3: 4 waves x 4 rows, one ahead; 4: two ahead; 5: one ahead + DMA interleaved; 6: 4 waves x 2 rows
// (the small-grid block), two ahead.  Shapes: 3x3 / stride-2, Cin % 32 = 0, Cout % 64 = 0, tile grid width % 32 = 0.
static int conv3_variant(int geo, Conv2Args& a, int id, hipStream_t st, int* launched) {
    const int gw = geo == C2_D ? a.W / 2 : a.W;
    if ((geo != C2_S && geo != C2_D) || a.Cin % 32 || a.Cout % 64 || gw % 32 || a.Cout / 64 > 32 || (geo == C2_D && ((a.H | a.W) & 1)) || id < 0 || id > 6) return 0;
    *launched = 1;
#define C3_CASE(ID, NW, RPW, PD, DIL)                                                                   \
    case ID: return geo == C2_S ? launch_conv3<C2_S, NW, 2, RPW, PD, DIL>(a, st) : launch_conv3<C2_D, NW, 2, RPW, PD, DIL>(a, st);
    switch (id) {
        C3_CASE(0, 8, 2, 1, 0)
        C3_CASE(1, 8, 2, 2, 0)
        C3_CASE(2, 8, 2, 2, 1)
        C3_CASE(3, 4, 4, 1, 0)
        C3_CASE(4, 4, 4, 2, 0)
        C3_CASE(5, 4, 4, 1, 1)
        C3_CASE(6, 4, 2, 2, 0)
    }
#undef C3_CASE
    return 0;
}

// ---- transposed 4x4 stride-2 convolution + [1,2,1]x[1,2,1] blur (+ LeakyReLU-backward mask from sign bits) as ONE 3x3 convolution
// to the four output parity classes (conv3_kernel<UB>): generator conv0_up -> blur (models/CustomLayers.py:143-152,175-177) and the
// discriminator's backward of "LeakyReLU -> blur -> conv1_down" (models/Blocks.py:140-146).
// sgx_pack_upblur: the layer's parameter w [O][I][3][3] (fp32) -> wc, bf16 [9][4N][K] + [22][2N][K] with (N, K) = (O, I) for the forward of
// an up layer (modes U / UF) and (I, O) for the data gradient of a down layer (mode D, adjoint); the 16 transposed-convolution taps are
// synthesised in fp32 exactly as sgx_pack_weight does (scale = w_mul / 16: the blur's normalisation), composed in fp32, rounded once:
// 9 composite taps (dy * 3 + dx; row n = (px * 2 + py) * N + co), then 22 correction tiles of 2N rows (py, co) for ONE px class each:
// 0..2 first fine column (dy; px = 0), 3..5 last fine column (px = 1), 6..9 corners TL TR BL BR (kept in LDS by the kernel), 10 + dx * 2 + px
// first fine row, 16 + dx * 2 + px last fine row (read from global memory by the one wave that owns that row).
// 1-D composition (k = [1,2,1] blur taps, w = kernel taps, output Y = 2i + p, input rows i - 1, i, i + 1 <-> d = 0, 1, 2):
//   p = 0: d0: k0 w2 + k1 w3;  d1: k0 w0 + k1 w1 + k2 w2;  d2: k2 w0        p = 1: d0: k0 w3;  d1: k0 w1 + k1 w2 + k2 w3;  d2: k1 w0 + k2 w1
__device__ __forceinline__ int upblur_terms(int p, int d, int (&ka)[3], int (&kw)[3]) {
    if (p == 0) {
        if (d == 0) { ka[0] = 0; kw[0] = 2; ka[1] = 1; kw[1] = 3; return 2; }
        if (d == 1) { ka[0] = 0; kw[0] = 0; ka[1] = 1; kw[1] = 1; ka[2] = 2; kw[2] = 2; return 3; }
        ka[0] = 2; kw[0] = 0; return 1;
    }
    if (d == 0) { ka[0] = 0; kw[0] = 3; return 1; }
    if (d == 1) { ka[0] = 0; kw[0] = 1; ka[1] = 1; kw[1] = 2; ka[2] = 2; kw[2] = 3; return 3; }
    ka[0] = 1; kw[0] = 0; ka[1] = 2; kw[1] = 1; return 2;
}
__global__ __launch_bounds__(256) void pack_upblur_kernel(const float* __restrict__ w, bf16_t* __restrict__ wc, int O, int I, int mode, int adjoint, float scale) {
    const int N = adjoint ? I : O, K = adjoint ? O : I;
    const int nmain = 9 * 4 * N * K, total = nmain + 22 * 2 * N * K;
    const float kb[3] = {1.f, 2.f, 1.f};
    const float sc = scale * (mode == SGX_PACK_D ? 0.25f : 1.f);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        int k, co, px, py, slot;                      // slot 0..8: composite tap dy * 3 + dx; 9 + t: correction tile t (rows (py, co) of ONE px class)
        if (e < nmain) {
            k = e % K;
            const int n4 = (e / K) % (4 * N);
            slot = e / (K * 4 * N); co = n4 % N; py = (n4 / N) & 1; px = n4 / (2 * N);
        } else {
            const int r = e - nmain;
            k = r % K;
            const int n2 = (r / K) % (2 * N), t = r / (K * 2 * N);
            slot = 9 + t; co = n2 % N; py = n2 / N;
            px = t < 3 ? 0 : (t < 6 ? 1 : (t < 10 ? (t - 6) & 1 : (t - 10) & 1));
        }
        const float* wp = w + ((size_t)(adjoint ? k : co) * I + (adjoint ? co : k)) * 9;       // parameter [o][i][3][3]
        // the transposed convolution's tap (ky, kx) as sgx_pack_weight synthesises it from the 3x3 parameter (modes D / U / UF)
        auto T = [&](int ky, int kx) {
            float v = 0.f;
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    const int y = ky - a2, x = kx - b2;
                    if (y >= 0 && y < 3 && x >= 0 && x < 3) v += wp[mode == SGX_PACK_UF ? (2 - y) * 3 + (2 - x) : y * 3 + x];
                }
            return v * sc;
        };
        int ay[3], wy[3], ax[3], wx[3];
        float v = 0.f;
        if (slot < 9) {
            const int dy = slot / 3, dx = slot % 3;
            const int ny = upblur_terms(py, dy, ay, wy), nx = upblur_terms(px, dx, ax, wx);
            for (int i = 0; i < ny; ++i)
                for (int j = 0; j < nx; ++j) v += kb[ay[i]] * kb[ax[j]] * T(wy[i], wx[j]);
        } else {
            const int t = slot - 9;
            if (t < 6) {                                  // first (0..2, px = 0) / last (3..5, px = 1) fine column: kernel column 0 / 3
                const bool last = t >= 3;
                const int ny = upblur_terms(py, t % 3, ay, wy);
                for (int i = 0; i < ny; ++i) v -= kb[ay[i]] * kb[last ? 2 : 0] * T(wy[i], last ? 3 : 0);
            } else if (t < 10) {                          // corners TL TR BL BR: the cross term that both the row and the column correction removed
                const int cy = (t - 6) >> 1, cx = (t - 6) & 1;
                if (py == cy) v = kb[cy ? 2 : 0] * kb[cx ? 2 : 0] * T(cy ? 3 : 0, cx ? 3 : 0);
            } else {                                      // first (10..15) / last (16..21) fine row, tile (dx * 2 + px): kernel row 0 / 3, class py = 0 / 1
                const bool last = t >= 16;
                const int dx = ((t - 10) % 6) >> 1;
                if (py == (last ? 1 : 0)) {
                    const int nx = upblur_terms(px, dx, ax, wx);
                    for (int j = 0; j < nx; ++j) v -= kb[last ? 2 : 0] * kb[ax[j]] * T(last ? 3 : 0, wx[j]);
                }
            }
        }
        wc[e] = f2bf(v);
    }
}
extern "C" int sgx_pack_upblur(const float* w, void* wc, int O, int I, int mode, int adjoint, float scale, void* stream) {
    SGX_REQUIRE(w && wc && O > 0 && I > 0 && (mode == SGX_PACK_D || mode == SGX_PACK_U || mode == SGX_PACK_UF), SGX_EINVAL, "pack_upblur: bad arguments");
    SGX_REQUIRE((mode == SGX_PACK_D) == (adjoint != 0), SGX_EINVAL, "pack_upblur: the transposed convolution is the forward of an up layer or the adjoint of a down layer");
    const int total = (9 * 4 + 22 * 2) * O * I;
    hipLaunchKernelGGL(pack_upblur_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, (hipStream_t)stream, w,
                       static_cast<bf16_t*>(wc), O, I, mode, adjoint, scale);
    SGX_LAUNCH_CHECK("pack_upblur_kernel");
    return 0;
}
extern "C" int sgx_conv_upblur_ok(int B, int H, int W, int Cin, int Cout, int dtype) {
    static const int on = [] { const char* e = getenv("SGX_CONV_UPBLUR"); return e ? atoi(e) : 1; }();   // A/B switch
    return on && dtype == SGX_BF16 && Cin == 32 && Cout == 16 && W % 32 == 0 && B > 0 && H > 0 ? 1 : 0;
}
extern "C" int sgx_conv_upblur(const void* x, const void* wc, void* y, const void* maskbits, int B, int H, int W, int Cin, int Cout, int dtype,
                               void* stream) {
    SGX_REQUIRE(x && wc && y, SGX_EINVAL, "conv_upblur: null argument");
    SGX_REQUIRE(dtype == SGX_BF16 && Cin == 32 && Cout == 16 && W % 32 == 0 && B > 0 && H > 0, SGX_EUNSUPPORTED,
                "conv_upblur: shape B%d %dx%d %d->%d dtype %d has no kernel (sgx_conv_upblur_ok == 0)", B, H, W, Cin, Cout, dtype);
    SGX_NOTE(2.0 * 36 * Cin * Cout * B * H * W, 2.0 * ((double)B * H * W * (Cin + 4.0 * Cout) + 36.0 * Cin * Cout) + (maskbits ? 0.5 * B * H * W * Cout : 0.0),
             "convU*blur%s B%d %dx%d %d->%d", maskbits ? "*bits" : "", B, H, W, Cin, Cout);
    Conv2Args a{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(wc), nullptr, static_cast<bf16_t*>(y), nullptr, B, H, W, 2 * H, 2 * W, Cin, 64,
                SGX_ACT_NONE, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<const unsigned char*>(maskbits)};
    a.wcorr = static_cast<const bf16_t*>(wc) + (size_t)9 * 64 * 32;
    // 8-wave blocks once their 512-pixel tiles fill the chip, else the 4-wave block (256-pixel tiles); SGX_UPBLUR_BPC=2: two compact 4-wave
    // blocks per CU (round 6)
    static const int bpc2 = [] { const char* e = getenv("SGX_UPBLUR_BPC"); return e ? atoi(e) : 1; }();
    const long tiles8 = (long)B * ((H + 15) / 16) * (W / 32), tiles4 = (long)B * ((H + 7) / 8) * (W / 32);
    if (bpc2 == 2 && tiles4 >= 2L * sgx_ncu()) return launch_conv3<C2_S, 4, 2, 2, 2, 0, 2>(a, (hipStream_t)stream);
    return tiles8 >= sgx_ncu() ? launch_conv3<C2_S, 8, 2, 2, 2, 0, 1>(a, (hipStream_t)stream) : launch_conv3<C2_S, 4, 2, 2, 2, 0, 1>(a, (hipStream_t)stream);
}

// *launched = 1 if the second-generation kernel ran; 0 leaves the shape to the first-generation kernel.
// would sgx_conv2_try (variant -1) take this bf16 shape?  (the split-K plan of conv.hip only covers launches that stay with the first generation)
int sgx_conv2_takes(int geo, int B, int H, int W, int Cin, int Cout) { return conv2_pick(geo, B, H, W, Cin, Cout, -1).nw != 0; }
int sgx_conv2_try(int geo, const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int act,
                  const void* mask, int variant, hipStream_t st, int* launched) {
    *launched = 0;
    if (variant >= 30) {                          // conv3_kernel configurations (probes, tests): 30 + id of the table in conv3_variant
        Conv2Args a3{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), bias, static_cast<bf16_t*>(y), static_cast<const bf16_t*>(mask), B, H, W,
                     geo == C2_D ? H / 2 : H, geo == C2_D ? W / 2 : W, Cin, Cout, act, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
#ifdef SGX_PROBE_BUILD                            // DMA ablations (WRONG results by design): probe builds only (make PROBE=1)
        { const char* e = getenv("SGX_CONV3_DBG"); a3.dbg = e ? atoi(e) : 0; }
#endif
        const int rc = conv3_variant(geo, a3, variant - 30, st, launched);
        return rc;
    }
    if (variant == 20 || variant == 21) {        // probe: 128 output channels per block on 16-channel K-steps (one staged patch feeds 4 accumulator rows)
        if (geo != C2_S || Cin % 16 || Cout % 128 || W % 32) return 0;
        Conv2Args a{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), bias, static_cast<bf16_t*>(y), static_cast<const bf16_t*>(mask), B, H, W,
                    H, W, Cin, Cout, act, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
        *launched = 1;
        return variant == 20 ? launch_conv2<C2_S, 8, 4, 16>(a, st) : launch_conv2<C2_S, 4, 4, 16>(a, st);
    }
    if (variant == 22 || variant == 23) {        // probe: 32 output channels per block (MF = 1) whatever Cout: twice the blocks of the 64-channel tile
        if ((geo != C2_S && geo != C2_D) || Cin % 32 || Cout % 32 || (geo == C2_D ? W / 2 : W) % 32) return 0;
        Conv2Args a{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), bias, static_cast<bf16_t*>(y), static_cast<const bf16_t*>(mask), B, H, W,
                    geo == C2_D ? H / 2 : H, geo == C2_D ? W / 2 : W, Cin, Cout, act, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
        *launched = 1;
        if (geo == C2_D) return variant == 22 ? launch_conv2<C2_D, 4, 1>(a, st) : launch_conv2<C2_D, 8, 1>(a, st);
        return variant == 22 ? launch_conv2<C2_S, 4, 1>(a, st) : launch_conv2<C2_S, 8, 1>(a, st);
    }
    const Conv2Pick p = conv2_pick(geo, B, H, W, Cin, Cout, variant);
    if (!p.nw) return 0;
    Conv2Args a{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), bias, static_cast<bf16_t*>(y), static_cast<const bf16_t*>(mask), B, H, W,
                geo == C2_D ? H / 2 : (geo == C2_U ? 2 * H : H), geo == C2_D ? W / 2 : (geo == C2_U ? 2 * W : W), Cin, Cout, act, 0, 0, 0, 0, 0, 0,
                nullptr, nullptr, nullptr, nullptr, nullptr};
    *launched = 1;
    const int nw = p.nw;
    if (p.k16) {
        if (geo == C2_S) return nw == 8 ? launch_conv2<C2_S, 8, 1, 16, true>(a, st) : launch_conv2<C2_S, 4, 1, 16, true>(a, st);
        if (geo == C2_D) return nw == 8 ? launch_conv2<C2_D, 8, 1, 16, false>(a, st) : launch_conv2<C2_D, 4, 1, 16, false>(a, st);
        return nw == 8 ? launch_conv2<C2_U, 8, 1, 32, true>(a, st) : launch_conv2<C2_U, 4, 1, 32, true>(a, st);
    }
    if (geo == C2_S) {
        if (p.mf2) return nw == 8 ? launch_conv2<C2_S, 8, 2>(a, st) : launch_conv2<C2_S, 4, 2>(a, st);
        return nw == 8 ? launch_conv2<C2_S, 8, 1>(a, st) : launch_conv2<C2_S, 4, 1>(a, st);
    }
    if (geo == C2_D) {
        if (p.mf2) return nw == 8 ? launch_conv2<C2_D, 8, 2>(a, st) : launch_conv2<C2_D, 4, 2>(a, st);
        return nw == 8 ? launch_conv2<C2_D, 8, 1>(a, st) : launch_conv2<C2_D, 4, 1>(a, st);
    }
    return nw == 8 ? launch_conv2<C2_U, 8, 1>(a, st) : launch_conv2<C2_U, 4, 1>(a, st);
}

// ---- 3x3 convolution whose store also produces the instance-norm statistics of the LayerEpilogue that follows (generator
// conv1 -> epi2, models/Blocks.py:86-87 over models/CustomLayers.py:224-233): tiles per image covered by the second-generation
// kernel, 0 = this shape has no fused variant (the caller runs the plain convolution and the separate statistics pass).
// tile slots of the persistent grid for a plan (== Conv2Args::nslots as launch_conv2 sets it)
static int conv2_stats_slots(const Conv2Pick& p, int B, int H, int W, int Cout) {
    const int th = 2 * p.nw, ntiles = B * ((H + th - 1) / th) * ((W + 31) / 32);
    const int ncb = p.k16 ? 1 : Cout / (p.mf2 ? 64 : 32);
    int bpc = 1;
    if (p.k16) {
        static const int bpc_max = [] { const char* e = getenv("SGX_CONV2_BPC"); return e ? atoi(e) : 2; }();
        const int lds = (p.nw == 8 ? C2Lds<C2_S, 8, 1, 16>::TOTAL : C2Lds<C2_S, 4, 1, 16>::TOTAL) + 1024;
        bpc = (160 * 1024) / lds;
        if (bpc > 32 / p.nw) bpc = 32 / p.nw;
        if (bpc > bpc_max) bpc = bpc_max;
        if (bpc < 1) bpc = 1;
    }
    int per = conv2_ncu() * bpc / (8 * ncb);
    const int need = (ntiles + 7) / 8;
    if (per > need) per = need;
    if (per < 1) per = 1;
    return per * 8;
}
extern "C" int sgx_conv3x3_stats_nparts(int B, int H, int W, int Cin, int Cout, int dtype) {
    if (dtype != SGX_BF16) return 0;
    const Conv2Pick p = conv2_pick(C2_S, B, H, W, Cin, Cout, -1);
    if (!p.nw) return 0;
    return conv2_stats_slots(p, B, H, W, Cout);
}
extern "C" int sgx_conv3x3_stats(const void* x, const void* w, void* y, const float* ebias, const float* noise, const float* nw_,
                                 double* part, size_t part_bytes, int B, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    SGX_REQUIRE(dtype == SGX_BF16, SGX_EUNSUPPORTED, "conv3x3_stats: bf16 only");
    SGX_REQUIRE(x && w && y && noise && nw_ && part, SGX_EINVAL, "conv3x3_stats: null argument");
    const Conv2Pick p = conv2_pick(C2_S, B, H, W, Cin, Cout, -1);
    SGX_REQUIRE(p.nw, SGX_EUNSUPPORTED, "conv3x3_stats: shape B%d %dx%d %d->%d has no fused variant (sgx_conv3x3_stats_nparts == 0)", B, H, W, Cin, Cout);
    const int npart = conv2_stats_slots(p, B, H, W, Cout);
    SGX_REQUIRE(part_bytes >= (size_t)B * npart * Cout * 2 * sizeof(double), SGX_EWORKSPACE, "conv3x3_stats: partials buffer %zu < %zu", part_bytes,
                (size_t)B * npart * Cout * 2 * sizeof(double));
    SGX_NOTE(2.0 * 9 * Cin * Cout * B * H * W, 2.0 * ((double)B * H * W * (Cin + Cout) + 9.0 * Cin * Cout), "convS+stats B%d %dx%d %d->%d", B, H, W, Cin, Cout);
    Conv2Args a{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), nullptr, static_cast<bf16_t*>(y), nullptr, B, H, W, H, W, Cin, Cout,
                SGX_ACT_NONE, 0, 0, 0, 0, 0, 0, ebias, noise, nw_, part, nullptr};
    a.part_slots = npart;
    hipStream_t st = (hipStream_t)stream;
    const int nw = p.nw;
    if (p.k16) return nw == 8 ? launch_conv2<C2_S, 8, 1, 16, true, EPI_STATS>(a, st) : launch_conv2<C2_S, 4, 1, 16, true, EPI_STATS>(a, st);
    if (p.mf2) return nw == 8 ? launch_conv2<C2_S, 8, 2, 32, false, EPI_STATS>(a, st) : launch_conv2<C2_S, 4, 2, 32, false, EPI_STATS>(a, st);
    return nw == 8 ? launch_conv2<C2_S, 8, 1, 32, false, EPI_STATS>(a, st) : launch_conv2<C2_S, 4, 1, 32, false, EPI_STATS>(a, st);
}

// ---- transposed convolution + the blur that follows it (+ the LeakyReLU-backward mask of the discriminator's backward chain)
// in one kernel: 1 if this shape has the fused variant, 0 = the caller runs sgx_conv4x4s2_up and sgx_blur3x3(_act).
extern "C" int sgx_conv4x4s2_up_blur_ok(int B, int H, int W, int Cin, int Cout, int dtype) {
    if (dtype != SGX_BF16) return 0;
    static const int on = [] { const char* e = getenv("SGX_CONV_UP_BLUR"); return e ? atoi(e) : 1; }();   // A/B switch
    if (!on) return 0;
    return conv2_pick(C2_U, B, H, W, Cin, Cout, -1, true).nw ? 1 : 0;
}
static int conv_up_blur_launch(const void* x, const void* w, void* y, const void* mask, const void* maskbits, int B, int H, int W, int Cin, int Cout,
                               int dtype, void* stream);
extern "C" int sgx_conv4x4s2_up_blur(const void* x, const void* w, void* y, const void* mask, int B, int H, int W, int Cin, int Cout,
                                     int dtype, void* stream) {
    return conv_up_blur_launch(x, w, y, mask, nullptr, B, H, W, Cin, Cout, dtype, stream);
}
extern "C" int sgx_conv4x4s2_up_blur_bits(const void* x, const void* w, void* y, const void* bits, int B, int H, int W, int Cin, int Cout,
                                          int dtype, void* stream) {
    SGX_REQUIRE(bits && Cout % 8 == 0, SGX_EINVAL, "conv4x4s2_up_blur_bits: sign bits [B][2H][2W][Cout/8] expected");
    return conv_up_blur_launch(x, w, y, nullptr, bits, B, H, W, Cin, Cout, dtype, stream);
}
static int conv_up_blur_launch(const void* x, const void* w, void* y, const void* mask, const void* maskbits, int B, int H, int W, int Cin, int Cout,
                               int dtype, void* stream) {
    SGX_REQUIRE(dtype == SGX_BF16, SGX_EUNSUPPORTED, "conv4x4s2_up_blur: bf16 only");
    SGX_REQUIRE(x && w && y, SGX_EINVAL, "conv4x4s2_up_blur: null argument");
    const Conv2Pick p = conv2_pick(C2_U, B, H, W, Cin, Cout, -1, true);
    SGX_REQUIRE(p.nw, SGX_EUNSUPPORTED, "conv4x4s2_up_blur: shape B%d %dx%d %d->%d has no fused variant (sgx_conv4x4s2_up_blur_ok == 0)", B, H, W, Cin, Cout);
    SGX_NOTE(2.0 * 16 * Cin * Cout * B * H * W, 2.0 * ((double)B * H * W * (Cin + 4.0 * Cout * (mask ? 2 : 1)) + 16.0 * Cin * Cout) + (maskbits ? 0.5 * B * H * W * Cout : 0.0),
             "convU+blur%s B%d %dx%d %d->%d", mask ? "*mask" : (maskbits ? "*bits" : ""), B, H, W, Cin, Cout);
    Conv2Args a{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), nullptr, static_cast<bf16_t*>(y), static_cast<const bf16_t*>(mask), B, H, W,
                2 * H, 2 * W, Cin, Cout, SGX_ACT_NONE, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<const unsigned char*>(maskbits)};
    hipStream_t st = (hipStream_t)stream;
    if (p.k16) return p.nw == 8 ? launch_conv2<C2_U, 8, 1, 32, true, EPI_BLUR>(a, st) : launch_conv2<C2_U, 4, 1, 32, true, EPI_BLUR>(a, st);
    return p.nw == 8 ? launch_conv2<C2_U, 8, 1, 32, false, EPI_BLUR>(a, st) : launch_conv2<C2_U, 4, 1, 32, false, EPI_BLUR>(a, st);
}

// ---- 3x3 convolution that also writes the SIGN bits of its (bias-added, activated) output: one byte per pixel and 8 channels.
// The discriminator block's conv0 (models/Blocks.py:139-140): its pre-activation is the mask of the LeakyReLU backward, and the
// backward passes then read 1 bit instead of 16 per element.  1 = available for this shape (bf16, second-generation kernel).
extern "C" int sgx_conv3x3_signbits_ok(int B, int H, int W, int Cin, int Cout, int dtype) {
    if (dtype != SGX_BF16 || Cout % 8) return 0;
    static const int on = [] { const char* e = getenv("SGX_SIGNBITS"); return e ? atoi(e) : 1; }();   // A/B switch
    return on && conv2_pick(C2_S, B, H, W, Cin, Cout, -1).nw ? 1 : 0;
}
extern "C" int sgx_conv3x3_signbits(const void* x, const void* w, const float* bias, void* y, void* bits, int B, int H, int W, int Cin,
                                    int Cout, int act, const void* mask, int dtype, void* stream) {
    SGX_REQUIRE(dtype == SGX_BF16, SGX_EUNSUPPORTED, "conv3x3_signbits: bf16 only");
    SGX_REQUIRE(x && w && y && bits, SGX_EINVAL, "conv3x3_signbits: null argument");
    const Conv2Pick p = conv2_pick(C2_S, B, H, W, Cin, Cout, -1);
    SGX_REQUIRE(p.nw && Cout % 8 == 0, SGX_EUNSUPPORTED, "conv3x3_signbits: shape B%d %dx%d %d->%d has no variant (sgx_conv3x3_signbits_ok == 0)", B, H, W, Cin, Cout);
    SGX_NOTE(2.0 * 9 * Cin * Cout * B * H * W, 2.0 * ((double)B * H * W * (Cin + Cout) + 9.0 * Cin * Cout) + (double)B * H * W * Cout / 8.0,
             "convS+bits B%d %dx%d %d->%d", B, H, W, Cin, Cout);
    Conv2Args a{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), bias, static_cast<bf16_t*>(y), static_cast<const bf16_t*>(mask), B, H, W, H, W,
                Cin, Cout, act, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, static_cast<unsigned char*>(bits)};
    hipStream_t st = (hipStream_t)stream;
    const int nw = p.nw;
    if (p.k16) return nw == 8 ? launch_conv2<C2_S, 8, 1, 16, true>(a, st) : launch_conv2<C2_S, 4, 1, 16, true>(a, st);
    if (p.mf2) return nw == 8 ? launch_conv2<C2_S, 8, 2>(a, st) : launch_conv2<C2_S, 4, 2>(a, st);
    return nw == 8 ? launch_conv2<C2_S, 8, 1>(a, st) : launch_conv2<C2_S, 4, 1>(a, st);
}

// ---- stride-2 convolution whose store applies the fade-in lerp of the discriminator's newest block and writes the activation's sign
// bits instead of the activation (models/GAN.py:425-427 over models/Blocks.py:143-146):  y = alpha * lrelu(conv(x) + bias) + beta * resid.
extern "C" int sgx_conv4x4s2_down_fade_ok(int B, int H, int W, int Cin, int Cout, int dtype) {
    if (dtype != SGX_BF16 || Cout % 8) return 0;
    static const int on = [] { const char* e = getenv("SGX_FUSE_FADE"); return e ? atoi(e) : 1; }();   // A/B switch
    return on && conv2_pick(C2_D, B, H, W, Cin, Cout, -1).nw ? 1 : 0;
}
static int conv_down_fade_launch(const void* x, const void* w, const float* bias, const void* resid, const float* pimg, const float* wr, const float* rb,
                                 float ws, float bs1, float bs2, float alpha, float beta, const float* ab_dev, void* y, void* bits, int B, int H, int W, int Cin,
                                 int Cout, int dtype, void* stream);
extern "C" int sgx_conv4x4s2_down_fade(const void* x, const void* w, const float* bias, const void* resid, float alpha, float beta, const float* ab_dev,
                                       void* y, void* bits, int B, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    SGX_REQUIRE(resid, SGX_EINVAL, "conv4x4s2_down_fade: null argument");
    return conv_down_fade_launch(x, w, bias, resid, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, alpha, beta, ab_dev, y, bits, B, H, W, Cin, Cout, dtype, stream);
}
// ... with the residual branch from_rgb(pimg) computed in the store: pimg fp32 [B][H/2][W/2][3] (the down-sampled image), wr = from_rgb.weight
// [Cout][3][1][1], ws = its w_mul (x the prescale of the residual, if any), rb = from_rgb.bias or NULL with its two scales (b_mul, prescale)
extern "C" int sgx_conv4x4s2_down_fade_rgb(const void* x, const void* w, const float* bias, const float* pimg, const float* wr, float ws, const float* rb,
                                           float bs1, float bs2, float alpha, float beta, const float* ab_dev, void* y, void* bits, int B, int H, int W,
                                           int Cin, int Cout, int dtype, void* stream) {
    SGX_REQUIRE(pimg && wr, SGX_EINVAL, "conv4x4s2_down_fade_rgb: null argument");
    return conv_down_fade_launch(x, w, bias, nullptr, pimg, wr, rb, ws, bs1, bs2, alpha, beta, ab_dev, y, bits, B, H, W, Cin, Cout, dtype, stream);
}
static int conv_down_fade_launch(const void* x, const void* w, const float* bias, const void* resid, const float* pimg, const float* wr, const float* rb,
                                 float ws, float bs1, float bs2, float alpha, float beta, const float* ab_dev, void* y, void* bits, int B, int H, int W, int Cin,
                                 int Cout, int dtype, void* stream) {
    SGX_REQUIRE(dtype == SGX_BF16, SGX_EUNSUPPORTED, "conv4x4s2_down_fade: bf16 only");
    SGX_REQUIRE(x && w && y && bits, SGX_EINVAL, "conv4x4s2_down_fade: null argument");
    const Conv2Pick p = conv2_pick(C2_D, B, H, W, Cin, Cout, -1);
    SGX_REQUIRE(p.nw && Cout % 8 == 0, SGX_EUNSUPPORTED, "conv4x4s2_down_fade: shape B%d %dx%d %d->%d has no variant (sgx_conv4x4s2_down_fade_ok == 0)", B, H, W, Cin, Cout);
    const double opx = (double)B * (H / 2) * (W / 2);
    SGX_NOTE(2.0 * 16 * Cin * Cout * opx, 2.0 * ((double)B * H * W * Cin + (pimg ? 1.0 : 2.0) * opx * Cout + 16.0 * Cin * Cout) + opx * Cout / 8.0 + (pimg ? 12.0 * opx : 0.0),
             "convD+fade%s B%d %dx%d %d->%d", pimg ? "(rgb)" : "", B, H, W, Cin, Cout);
    Conv2Args a{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), bias, static_cast<bf16_t*>(y), nullptr, B, H, W, H / 2, W / 2, Cin, Cout,
                SGX_ACT_LRELU, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, static_cast<unsigned char*>(bits), nullptr,
                static_cast<const bf16_t*>(resid), alpha, beta, ab_dev, pimg, wr, rb, ws, bs1, bs2};
    hipStream_t st = (hipStream_t)stream;
    const int nw = p.nw;
    if (p.k16) return nw == 8 ? launch_conv2<C2_D, 8, 1, 16, false>(a, st) : launch_conv2<C2_D, 4, 1, 16, false>(a, st);
    if (p.mf2) return nw == 8 ? launch_conv2<C2_D, 8, 2>(a, st) : launch_conv2<C2_D, 4, 2>(a, st);
    return nw == 8 ? launch_conv2<C2_D, 8, 1>(a, st) : launch_conv2<C2_D, 4, 1>(a, st);
}
