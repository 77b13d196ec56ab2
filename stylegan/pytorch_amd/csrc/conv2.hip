// Second-generation bf16 convolution kernels for gfx950: v_mfma_f32_32x32x16_bf16, operands staged by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) into TWO LDS stages, one barrier per K-chunk.
//
// Why (round-1 profile, profiles/r01_*): the first-generation kernel (conv.hip) runs one wave per SIMD with a single LDS
// stage -- global load -> ds_write -> barrier -> MFMA -> barrier serialise, and at 64^2..256^2 (the MFMA-bound layers) it
// reaches 12-24 % of the matrix peak while neither LDS bandwidth (6 %) nor HBM is the limit.  Here a block is 8 waves
// (two per SIMD) on a 16 x 32 pixel tile x 64 output channels; while the waves run the 72 MFMAs of K-chunk c out of stage
// c&1, the DMA engine fills stage (c+1)&1 with the next chunk (or the next tile's first chunk: the block is persistent).
//
// GEMM view as in conv.hip: M = output channels (A = packed weights w[tap][n][k]), N = output pixels (B = activations),
// K = taps x input channels, K-chunks of 32 channels.  Wave tile 64 channels x 64 pixels (2 x 2 accumulators of 32x32).
//
// LDS image of one stage: [patch rows][weight rows], every row = 32 channels = 64 bytes = four 16-byte slots.  The DMA
// writes lane-linear (wave-uniform base + lane*16), so slot s of an instruction holds (row s/4, chunk (s%4) ^ swz(row)):
// the swizzle is applied on the per-lane SOURCE address and again on the fragment read (cdna guide rule 21).
//   patch row (pr, pc): swz = (pc >> 2) & 3   -- a fragment's 32 lanes read 32 consecutive pc of one patch row
//   weight row (tap, n): swz = (n >> 2) & 3   -- 32 consecutive n
// which makes every ds_read_b128 conflict-free (16-lane service groups {0-3,12-15,20-27}/{4-11,16-19,28-31} hit 16
// distinct 16-byte bank groups for any starting column).  Halo / out-of-image lanes read a 64-byte page of zeros.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __attribute__((aligned(64))) const unsigned sgx_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct Conv2Args {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
    int B, H, W, Cin, Cout, act;
    int tiles_x, tiles_y, ntiles;      // pixel tiles of one channel block: B * tiles_y * tiles_x
    int ncb, nslots;                   // channel blocks; persistent stride over tiles
};

// NW waves per block, each owning 2 rows x 32 pixels; MF 32-channel accumulator rows per wave (block: MF*32 channels).
template <int NW, int MF>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv3x3_v2_kernel(Conv2Args a) {
    constexpr int TH = 2 * NW, TW = 32, PH = TH + 2, PW = TW + 2, BCO = MF * 32;
    constexpr int PROWS = PH * PW;
    constexpr int P_INSTR = (PROWS * 4 + 63) / 64, P_BYTES = P_INSTR * 1024;
    constexpr int W_INSTR = 9 * BCO * 4 / 64, W_BYTES = W_INSTR * 1024;
    constexpr int STAGE = P_BYTES + W_BYTES;
    constexpr int NPI = (P_INSTR + NW - 1) / NW, NWI = (W_INSTR + NW - 1) / NW;
    constexpr int OROW = BCO * 2 + 16, VPR = BCO * 2 / 16;                 // epilogue scratch: pixel-major rows
    static_assert(NW * 32 * OROW <= P_BYTES, "epilogue scratch must fit the patch region");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (channel block, tile slot): blocks of one XCD (id % 8) that are neighbours in id/8 share a tile slot and
    // differ in the channel block, so the patches they share are served by that XCD's L2
    const int bid = blockIdx.x, xcd = bid & 7, j8 = bid >> 3;
    const int cb = j8 % a.ncb, slot = (j8 / a.ncb) * 8 + xcd;
    const int co0 = cb * BCO;
    if (slot >= a.ntiles) return;
    const int my_tiles = (a.ntiles - slot + a.nslots - 1) / a.nslots;
    const int nchunks = a.Cin >> 5;
    const int nsteps = my_tiles * nchunks;

    // ---- per-lane DMA descriptors (tile independent)
    int prel[NPI], ppos[NPI], wrel[NWI];
#pragma unroll
    for (int jj = 0; jj < NPI; ++jj) {
        const int s = (jj * NW + wave) * 64 + lane, row = s >> 2, c = s & 3;
        const int pr = row / PW, pc = row % PW;
        prel[jj] = (pr * a.W + pc) * a.Cin + ((c ^ ((pc >> 2) & 3)) << 3);
        ppos[jj] = (row < PROWS) ? ((pr << 8) | pc) : -1;
    }
#pragma unroll
    for (int jj = 0; jj < NWI; ++jj) {
        const int s = (jj * NW + wave) * 64 + lane, row = s >> 2, c = s & 3;
        const int tap = row / BCO, n = row % BCO;
        wrel[jj] = ((tap * a.Cout + co0 + n) * a.Cin) + ((c ^ ((n >> 2) & 3)) << 3);
    }
    // ---- per-lane fragment read offsets inside a stage
    int poff[3][2], woff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        woff[ks] = P_BYTES + l31 * 64 + (((hi + 2 * ks) ^ ((l31 >> 2) & 3)) << 4);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int pc = l31 + dx;
            poff[dx][ks] = (2 * wave * PW + pc) * 64 + (((hi + 2 * ks) ^ ((pc >> 2) & 3)) << 4);
        }
    }

    const bf16_t* __restrict__ xg = a.x;
    const bf16_t* __restrict__ wg = a.w;
    const unsigned long long zaddr = reinterpret_cast<unsigned long long>(sgx_zero_page) + (lane & 3) * 16;

    auto tile_coords = [&](int t, int& b, int& ty0, int& tx0) {
        const int tx_i = t % a.tiles_x; t /= a.tiles_x;
        const int ty_i = t % a.tiles_y;
        b = t / a.tiles_y; ty0 = ty_i * TH; tx0 = tx_i * TW;
    };
    // stage `step` (tile index it, K-chunk kc) into LDS stage buffer `buf`
    auto issue = [&](int step, char* buf) {
        const int it = step / nchunks, kc = step - it * nchunks;
        int b, ty0, tx0;
        tile_coords(slot + it * a.nslots, b, ty0, tx0);
        const int iy0 = ty0 - 1, ix0 = tx0 - 1;
        const bf16_t* base = xg + (((long)b * a.H + iy0) * a.W + ix0) * a.Cin + kc * 32;
#pragma unroll
        for (int jj = 0; jj < NPI; ++jj) {
            const int ii = jj * NW + wave;
            if (ii < P_INSTR) {
                const int pp = ppos[jj];
                const int gy = iy0 + (pp >> 8), gx = ix0 + (pp & 255);
                // branch-free select between the patch element and the page of zeros (a ?: on pointers compiles to
                // divergent branches around every DMA instruction)
                const unsigned long long ok = ((pp >= 0) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W)) ? ~0ull : 0ull;
                const unsigned long long pa = reinterpret_cast<unsigned long long>(base + prel[jj]);
                glds16(reinterpret_cast<const void*>(zaddr + ((pa - zaddr) & ok)), buf + ii * 1024);
            }
        }
        if (nchunks > 2 || step < 2) {               // <= 2 chunks: chunk kc's weights live in stage kc for the whole launch
            const bf16_t* w0 = wg + kc * 32;
#pragma unroll
            for (int jj = 0; jj < NWI; ++jj) {
                const int ii = jj * NW + wave;
                if (ii < W_INSTR) glds16(w0 + wrel[jj], buf + P_BYTES + ii * 1024);
            }
        }
    };

    f32x16 acc[MF][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][f][r] = 0.f;
    };
    zero_acc();

    issue(0, smem);
    for (int step = 0; step < nsteps; ++step) {
        char* cur = smem + (step & 1) * STAGE;
        __syncthreads();                         // (vmcnt(0) first) stage `step` landed; everyone is done with step-1
        if (step + 1 < nsteps) issue(step + 1, smem + ((step + 1) & 1) * STAGE);
        // ---- 72 MFMAs in 18 sub-steps (column shift dx, k-step ks, row shift dy).  Per (dx, ks) group the four patch rows
        // this wave touches are read once; fragments are software-pipelined one sub-step (weights) / one group (patch rows)
        // ahead so that the LDS latency hides under the MFMAs of the previous sub-step.
        {
            bf16x8 brow[2][4], af[2][MF];
            auto ld_brow = [&](int g, bf16x8 (&dst)[4]) {
                const int dx = g >> 1, ks = g & 1;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r] = *reinterpret_cast<const bf16x8*>(cur + poff[dx][ks] + r * PW * 64);
            };
            auto ld_af = [&](int sub, bf16x8 (&dst)[MF]) {
                const int g = sub / 3, dy = sub % 3, dx = g >> 1, ks = g & 1;
#pragma unroll
                for (int m = 0; m < MF; ++m)
                    dst[m] = *reinterpret_cast<const bf16x8*>(cur + woff[ks] + ((dy * 3 + dx) * BCO + m * 32) * 64);
            };
            ld_brow(0, brow[0]);
            ld_af(0, af[0]);
#pragma unroll
            for (int sub = 0; sub < 18; ++sub) {
                const int g = sub / 3, dy = sub % 3;
                if (sub + 1 < 18) ld_af(sub + 1, af[(sub + 1) & 1]);
                if (dy == 0 && g + 1 < 6) ld_brow(g + 1, brow[(g + 1) & 1]);
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sub & 1][m], brow[g & 1][f + dy], acc[m][f], 0, 0, 0);
            }
        }
        const int it = step / nchunks, kc = step - it * nchunks;
        if (kc == nchunks - 1) {
            // ---- epilogue: bias, activation, bf16; transposed through a wave-private LDS scratch (the patch region of
            // the stage just consumed) so that every lane stores 16 bytes and 8 lanes cover a 128-byte channel row
            int b, ty0, tx0;
            tile_coords(slot + it * a.nslots, b, ty0, tx0);
            float4 bv[MF][4];
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    bv[m][g] = a.bias ? *reinterpret_cast<const float4*>(a.bias + co0 + m * 32 + 8 * g + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();                     // every wave is done reading this stage's patch
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
                    const uint4 val = *reinterpret_cast<const uint4*>(scr + px * OROW + v * 16);
                    const int ox = tx0 + px;
                    if (oy < a.H && ox < a.W)
                        *reinterpret_cast<uint4*>(a.y + (((size_t)b * a.H + oy) * a.W + ox) * a.Cout + co0 + v * 8) = val;
                }
            }
            zero_acc();
        }
    }
}

static int conv2_ncu() {
    static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        return n;
    }();
    return ncu;
}

template <int NW, int MF>
static int launch_conv2(Conv2Args& a, hipStream_t st) {
    constexpr int TH = 2 * NW, PH = TH + 2, PW = 34, BCO = MF * 32;
    constexpr int P_BYTES = ((PH * PW * 4 + 63) / 64) * 1024, W_BYTES = (9 * BCO * 4 / 64) * 1024;
    constexpr int LDS = 2 * (P_BYTES + W_BYTES);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_v2_kernel<NW, MF>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)attr;
    a.tiles_x = (a.W + 31) / 32; a.tiles_y = (a.H + TH - 1) / TH;
    a.ntiles = a.B * a.tiles_y * a.tiles_x;
    a.ncb = a.Cout / BCO;
    int per = conv2_ncu() / (8 * a.ncb);                 // tile slots per XCD (one block per CU)
    const int need = (a.ntiles + 7) / 8;
    if (per > need) per = need;
    if (per < 1) per = 1;
    a.nslots = per * 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * a.ncb * per)), dim3(NW * 64), LDS, st, a);
    SGX_LAUNCH_CHECK("conv3x3_v2_kernel");
    return 0;
}

// Which layers take this kernel (SGX_CONV2=0 switches it off: A/B against conv.hip).  Returns 1 if launched, 0 if the
// shape is left to the first-generation kernel, <0 / >0 on error.
// ``variant``: -1 = choose (environment switch + heuristics), 4 / 8 = force the 4- / 8-wave block (A/B probes, tests).
int sgx_conv2_try_3x3(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int act,
                      int variant, hipStream_t st, int* launched) {
    static const int on = [] { const char* e = getenv("SGX_CONV2"); return e ? atoi(e) : 1; }();
    *launched = 0;
    if ((variant < 0 && !on) || Cin % 32 != 0 || Cout % 64 != 0 || W % 32 != 0 || Cout / 64 > 32 || H < 1) return 0;
    Conv2Args a{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), bias, static_cast<bf16_t*>(y), B, H, W, Cin, Cout, act, 0, 0, 0, 0, 0};
    const long blocks8 = (long)B * ((H + 15) / 16) * (W / 32) * (Cout / 64);
    static const int force_nw = [] { const char* e = getenv("SGX_CONV2_NW"); return e ? atoi(e) : 0; }();
    // measured (profiles/r02_conv2_probe.txt): the 8-wave block wins once its 512-pixel tiles fill the chip, the 4-wave
    // block (256-pixel tiles) down to one block per CU, below that the first-generation kernel's 64-pixel tiles do
    const long blocks4 = (long)B * ((H + 7) / 8) * (W / 32) * (Cout / 64);
    if (variant < 0 && !force_nw && blocks4 < conv2_ncu()) return 0;
    const int nw = variant > 0 ? variant : (force_nw ? force_nw : (blocks8 >= conv2_ncu() ? 8 : 4));
    *launched = 1;
    if (nw == 8) return launch_conv2<8, 2>(a, st);
    return launch_conv2<4, 2>(a, st);
}
