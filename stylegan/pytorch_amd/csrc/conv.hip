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
#include <stdlib.h>

enum { G3X3 = 0, GDOWN = 1, GUP = 2, GUPA = 3 };
// second-generation bf16 weight-gradient kernels (wgrad2.hip): plan = pixel splits (0: shape stays here); launch writes the
// same per-split partials as wgrad_kernel below
int sgx_wgrad2_plan(int geo, int B, int H, int W, int Ck, int Cn, int* nct_n, int* nct_k, int* kbw, int* ntiles);
int sgx_wgrad2_launch(int geo, const void* kside, const void* nside, float* ws, size_t ws_bytes, int B, int H, int W, int Ck, int Cn,
                      int want_bias, hipStream_t st, int* nsplit_out);
int sgx_conv2_try(int geo, const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int act,
                  const void* mask, int variant, hipStream_t st, int* launched);        // conv2.hip
int sgx_conv2_takes(int geo, int B, int H, int W, int Cin, int Cout);                   // conv2.hip   // GUPA: GUP with all four parity classes in one block (bf16)

// LDS operand tiles are arrays of rows (one pixel, or one (tap, output channel) weight row) holding KC channels.
// rowb_*: row pitch in bytes; load(): the lane's MFMA fragment (k = kk-step, q = lane>>4); lstore(): one 16-byte chunk.
// bf16/KC=32 rows are 64 B unpadded with the four 16-byte chunks XOR-swizzled by ((row>>1)&3): ds_read_b128 is served in
// 16-lane groups that mix rows {0-3,12-15} of one q with rows {4-11} of the neighbouring q, and this swizzle is
// conflict-free for any row offset (brute-forced; the 80-byte padded pitch is 2-way conflicted there, but it is the
// conflict-free choice when consecutive lanes are 2 rows apart, i.e. the stride-2 input tile).
template <typename T, int KC> struct Frag;
template <> struct Frag<float, 16> {
    static constexpr int NK = 4;
    static constexpr int rowb_w() { return 68; }                 // 17 dwords: conflict-free ds_read_b32 across 16 rows
    static constexpr int rowb_in(int) { return 68; }
    static constexpr bool swz_w() { return false; }
    static constexpr bool swz_in(int) { return false; }
    typedef float frag_t;
    static constexpr int pad_pw(int pw, int) { return pw; }
    __device__ static __forceinline__ int rd_off(int row, int rowb, bool, int q) { return row * rowb + q * 4; }
    __device__ static __forceinline__ frag_t ld(const char* p, int kk) { return *reinterpret_cast<const float*>(p + kk * 16); }
    __device__ static __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int lds_off(int row, int rowb, bool, int v) { return row * rowb + v * 16; }
    __device__ static __forceinline__ void lstore_at(char* p, uint4 val) {   // 68-byte rows: dword stores
        unsigned* d = reinterpret_cast<unsigned*>(p);
        d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
    }
};
template <> struct Frag<bf16_t, 32> {
    static constexpr int NK = 1;
    static constexpr int rowb_w() { return 64; }
    static constexpr int rowb_in(int is) { return is == 1 ? 64 : 80; }
    static constexpr bool swz_w() { return true; }
    static constexpr bool swz_in(int is) { return is == 1; }
    typedef bf16x8 frag_t;
    // swizzled tiles: a multiple-of-8 patch width keeps (row>>1)&3 unchanged when a tap moves one patch row down,
    // so every fragment address is (per-lane base for the tap's x shift) + compile-time immediate
    static constexpr int pad_pw(int pw, int is) { return is == 1 ? (pw + 7) / 8 * 8 : pw; }
    __device__ static __forceinline__ int rd_off(int row, int rowb, bool sw, int q) {
        return row * rowb + (sw ? (q ^ ((row >> 1) & 3)) : q) * 16;
    }
    __device__ static __forceinline__ frag_t ld(const char* p, int) { return *reinterpret_cast<const bf16x8*>(p); }
    __device__ static __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int lds_off(int row, int rowb, bool sw, int v) {
        return row * rowb + (sw ? (v ^ ((row >> 1) & 3)) : v) * 16;
    }
    __device__ static __forceinline__ void lstore_at(char* p, uint4 val) { *reinterpret_cast<uint4*>(p) = val; }
};
template <> struct Frag<bf16_t, 16> {
    // 16 channels = 32 B per row.  Stride-1 tiles: unpadded rows, the two 16-byte halves swapped for rows with bit 3
    // set, so a half-wave's ds_read_b64 (16 rows x {q=0,1}) touches every bank once and the 16-byte staging stores are
    // contiguous.  Stride-2 input tile: 48-byte pitch (conflict-free reads when lanes are two rows apart).
    static constexpr int NK = 1;
    static constexpr int rowb_w() { return 32; }
    static constexpr int rowb_in(int is) { return is == 1 ? 32 : 48; }
    static constexpr bool swz_w() { return true; }
    static constexpr bool swz_in(int is) { return is == 1; }
    typedef s16x4 frag_t;
    static constexpr int pad_pw(int pw, int is) { return is == 1 ? (pw + 15) / 16 * 16 : pw; }
    __device__ static __forceinline__ int rd_off(int row, int rowb, bool sw, int q) {
        const int half = sw ? ((q >> 1) ^ ((row >> 3) & 1)) : (q >> 1);
        return row * rowb + half * 16 + (q & 1) * 8;
    }
    __device__ static __forceinline__ frag_t ld(const char* p, int) { return *reinterpret_cast<const s16x4*>(p); }
    __device__ static __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int lds_off(int row, int rowb, bool sw, int v) {
        return row * rowb + (sw ? (v ^ ((row >> 3) & 1)) : v) * 16;
    }
    __device__ static __forceinline__ void lstore_at(char* p, uint4 val) { *reinterpret_cast<uint4*>(p) = val; }
};

struct ConvArgs {
    const void* x; const void* w; const float* bias; void* y;
    int B, H, W, OH, OW, OHc, OWc, Cin, Cout, act, tiles_x, tiles_y, ntiles;
    const void* mask;   // 3x3 only: y *= slope(mask) in the store (mask: a tensor shaped like y; the activation-backward of the layer below)
    int dbg;   // ablation switches (SGX_CONV_DBG, profiling only): 1 no MFMA, 2 no global loads, 4 no LDS stores, 8 no output stores
    int bands;          // XCD-aware tile order (grid x a multiple of 8): see conv_kernel
    // Split-K (round 6, bf16, the 512-channel layers at 4^2..32^2 of a small batch): ksplit > 1 = blockIdx.z / NCLS selects one of ksplit equal
    // ranges of the K-chunks; the block writes its RAW fp32 accumulators to kws[split][output pixel][Cout] and conv_splitk_finish sums the
    // splits in a fixed order and applies bias / activation / mask.  Why: these launches are 32-128 blocks that each stream their whole
    // 0.26-0.6 MB weight slice at ONE CU's load rate (~25 GB/s): 15-36 us for 0.5-5 GFLOP, whatever the K-chunk depth.
    int ksplit; float* kws;
};

template <int GEO> struct Geo;
template <> struct Geo<G3X3> { static constexpr int IS = 1, TK = 3, NCLS = 1; };
template <> struct Geo<GDOWN> { static constexpr int IS = 2, TK = 4, NCLS = 1; };
template <> struct Geo<GUP> { static constexpr int IS = 1, TK = 2, NCLS = 4; };
// all four output-parity classes of the transposed conv in ONE block: the coarse input patch (halo 1 on every side, i.e.
// the 3x3 patch geometry) is staged once instead of once per class, and the block owns complete 2x2 output quads, so its
// stores are whole contiguous rows of the fine image instead of every other pixel.
template <> struct Geo<GUPA> { static constexpr int IS = 1, TK = 3, NCLS = 1; };

// Small tiles of the HBM-bound layers (16/32 channels at 512^2..1024^2) want MANY resident blocks: a block has one
// tile's loads in flight, and bytes in flight per CU -- not MFMA rate -- set their speed.
#ifndef SGX_CONV_OCC_SMALL
#define SGX_CONV_OCC_SMALL 3
#endif
constexpr int conv_min_waves(int tsize, int CT, int BP, int GEO, int KC) {
    if (KC > 32) return 2;                                   // deep K-chunks: LDS allows two blocks per CU anyway
    if (GEO == GUPA) return CT * BP >= 512 ? 1 : 2;          // four classes of accumulators
    if (GEO == GDOWN && tsize == 2 && KC == 16) return 3;   // HBM-bound 16-channel stride-2 layers (measured: 69 -> 64 us)
    if (CT * BP >= 1024 || GEO == GDOWN) return 1;
    if (tsize == 2 && CT * BP <= 256) return SGX_CONV_OCC_SMALL;
    return 2;
}
// XCD-band tile order (see conv_kernel) is compiled only into the instantiations whose launches have enough tiles for it to
// matter: the 16x16-pixel tiles of the bf16 3x3 / all-class transposed layers (>= 256^2).  Everywhere else its index registers
// cost a wave of occupancy or spill (compiler resource remarks, tools/kernel_resources.py: 40 instantiations lost a wave per
// SIMD, 8 gained scratch; the 1024^2 stride-2 layer went from 64 to 89 us, the 64^2 one from 33 to 51), so those keep the
// plain walk and its single live tile index.
template <typename T>
constexpr bool conv_bands_ok(int GEO, int TH, int TW, int BP) {
    return sizeof(T) == 2 && (GEO == G3X3 || GEO == GUPA) && TH == 16 && TW == 16 && BP == 256;
}
template <typename T, int KC, int GEO, int TH, int TW, int BP, int CT, bool SK = false>
// Register budget: two waves per SIMD (<= 256 VGPR+AGPR) except for the 64-channel x 256-pixel tile, whose prefetch
// registers would spill at that bound -- and a spilled descriptor reload (scratch_load + s_waitcnt vmcnt) serialises
// the whole global prefetch behind it (seen in the ISA), which is far worse than one wave per SIMD.
__global__ __launch_bounds__(256, conv_min_waves(sizeof(T), CT, BP, GEO, KC)) void conv_kernel(ConvArgs a) {
    // bf16 K-chunks deeper than one MFMA k-step (KC = 64 / 128, the 512-channel layers at 4^2..16^2: 4x fewer
    // load -> LDS -> barrier -> MFMA stages per tile, which is what bounds those launches) are staged as KC/32 PLANES, each
    // laid out exactly like the KC = 32 tile (64-byte swizzled rows), so all fragment addressing stays as it is.
    constexpr int FK = (sizeof(T) == 2 && KC > 32) ? 32 : KC;
    constexpr int NPL = KC / FK;
    using F = Frag<T, FK>;
    constexpr int IS = Geo<GEO>::IS, TK = Geo<GEO>::TK, NT = TK * TK;
    constexpr int PH = (TH - 1) * IS + TK, PW = (TW - 1) * IS + TK, PWP = F::pad_pw(PW, IS);
    constexpr int NI = BP / (TH * TW), SPW = BP / 64, BCO = CT * 16;
    constexpr int VE = 16 / (int)sizeof(T), VPP = KC / VE, VPR = FK / VE;
    constexpr int ROWI = F::rowb_in(IS), ROWW = F::rowb_w();
    constexpr bool SWI = F::swz_in(IS), SWW = F::swz_w();
    constexpr bool UPA = GEO == GUPA;
    constexpr int NTW = UPA ? 16 : NT;                             // weight taps staged in LDS
    constexpr int NCL = UPA ? 4 : 1;                               // output-parity classes accumulated by this block
    constexpr int IN_PLANE = (NI * PH * PWP * ROWI + 15) / 16 * 16, W_PLANE = NTW * BCO * ROWW;
    constexpr int IN_BYTES = NPL * IN_PLANE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* in_lds = smem;
    char* w_lds = smem + IN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, l15 = lane & 15;
    // Persistent over pixel tiles: block x walks tiles x, x+gridDim.x, ... of its (output-channel block, parity class);
    // the first K-chunk of the NEXT tile is prefetched into registers while the current tile's MFMAs run.
    const int co0 = blockIdx.y * BCO;
    int py = 0, px = 0;
    constexpr int NCLS_Z = Geo<GEO>::NCLS;
    if (GEO == GUP) { py = (SK ? (int)(blockIdx.z % NCLS_Z) : (int)blockIdx.z) >> 1; px = blockIdx.z & 1; }
    // split-K (the SK instantiations only: the others are instruction for instruction what they were): this block's range of K-chunks
    int split = 0, k_lo = 0, k_hi = a.Cin;
    if constexpr (SK) {
        const int nkc = a.Cin / KC;
        split = (int)blockIdx.z / NCLS_Z;
        k_lo = (split * nkc / a.ksplit) * KC; k_hi = ((split + 1) * nkc / a.ksplit) * KC;
    }
    // Per-thread staging descriptors, computed ONCE: the index arithmetic of the global->LDS copy (which element of
    // the halo patch / weight tile this thread moves) does not depend on the tile or the K-chunk.  (Measured: doing
    // it per chunk cost 3-8 VALU instructions per MFMA and made the kernel issue-bound.)
    constexpr int NIN = (NI * PH * PW * VPP + 255) / 256, NWT = (NTW * BCO * VPP + 255) / 256;
    int in_rel[NIN], in_pos[NIN], in_dst[NIN], w_rel[NWT];
    // descriptor j sits 256/VPP rows further (swizzle-neutral)
    const int w_dst0 = ((tid % VPP) / VPR) * W_PLANE + F::lds_off(tid / VPP, ROWW, SWW, (tid % VPP) % VPR);
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
        const int idx = tid + j * 256;
        const int v = idx % VPP, pixel = idx / VPP;
        const int il = pixel / (PH * PW), rem = pixel % (PH * PW);
        const int pr = rem / PW, pc = rem % PW;
        const bool in_range = idx < NI * PH * PW * VPP;
        in_rel[j] = ((il * a.H + pr) * a.W + pc) * a.Cin + v * VE;
        in_pos[j] = in_range ? ((il << 20) | (pr << 10) | pc) : -1;
        in_dst[j] = (v / VPR) * IN_PLANE + F::lds_off((il * PH + pr) * PWP + pc, ROWI, SWI, v % VPR);
    }
#pragma unroll
    for (int j = 0; j < NWT; ++j) {
        const int idx = tid + j * 256;
        const int v = idx % VPP, row = idx / VPP;
        const int n = row % BCO, t = row / BCO;
        int tg = t;
        if (GEO == GUP) tg = (3 - py - 2 * (t >> 1)) * 4 + (3 - px - 2 * (t & 1));
        w_rel[j] = (idx < NTW * BCO * VPP) ? ((tg * a.Cout + co0 + n) * a.Cin + v * VE) : -1;
    }
    int img0, ty0, tx0;                                // coordinates of the tile being PREFETCHED (gload) ...
    long in_base = 0;                                  // element offset of its patch origin (may be negative: halo)
    unsigned in_ok = 0;                                // per-descriptor validity (inside image and batch)
    auto set_tile = [&](int tile) {
        const int tx_i = tile % a.tiles_x; tile /= a.tiles_x;
        const int ty_i = tile % a.tiles_y;
        img0 = (tile / a.tiles_y) * NI;
        ty0 = ty_i * TH; tx0 = tx_i * TW;
        const int iy0 = ty0 * IS + (GEO == GUP ? py - 1 : -1);
        const int ix0 = tx0 * IS + (GEO == GUP ? px - 1 : -1);
        in_base = (((long)img0 * a.H + iy0) * a.W + ix0) * a.Cin;
        in_ok = 0;
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            const int pp = in_pos[j];
            const int b = img0 + (pp >> 20), gy = iy0 + ((pp >> 10) & 1023), gx = ix0 + (pp & 1023);
            const bool ok = (pp >= 0) && (b < a.B) && ((unsigned)gy < (unsigned)a.H) && ((unsigned)gx < (unsigned)a.W);
            in_ok |= (ok ? 1u : 0u) << j;
        }
    };
    const T* __restrict__ xg = static_cast<const T*>(a.x);
    const T* __restrict__ wg = static_cast<const T*>(a.w);

    // LDS byte offsets of the lane's fragments, computed once: per pixel sub-tile and tap column for the input tile,
    // one for the weight tile; taps rows / weight rows add compile-time immediates in the MFMA loop.
    int inoff[SPW][TK];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int m = (wave * SPW + s) * 16 + l15;
        const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
#pragma unroll
        for (int tx = 0; tx < TK; ++tx) inoff[s][tx] = F::rd_off((il * PH + r * IS) * PWP + c * IS + tx, ROWI, SWI, q);
    }
    const int woff = F::rd_off(l15, ROWW, SWW, q);
    f32x4 acc[NCL * CT][SPW];                          // [class * CT + channel sub-tile][pixel sub-tile]
#pragma unroll
    for (int ct = 0; ct < NCL * CT; ++ct)
#pragma unroll
        for (int s = 0; s < SPW; ++s) acc[ct][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Staging is software pipelined through registers: the global loads of K-chunk c+1 (or of the next tile's first
    // chunk) are issued before the MFMAs of chunk c and written to LDS after them, so HBM/L2 latency hides under them.
    uint4 rin[NIN], rwt[NWT];
    // one K-chunk: the weight tile is the same for every pixel tile and is staged once -- unless this block's output
    // staging (GUPA: the whole fine tile) reaches into the weight region of the LDS, then it is re-staged per tile
    constexpr bool OUT_CLOBBERS_W = (UPA && 4 * BP * (BCO * 2 + 16) > IN_BYTES) ||
                                    (!UPA && sizeof(T) == 2 && CT >= 2 && BP * (BCO * 2 + 16) > IN_BYTES);
    const bool w_static = (a.Cin == KC) && !OUT_CLOBBERS_W;      // (never with split-K: >= 2 chunks)
    auto gload = [&](int k0, bool with_w) {
        const T* src0 = xg + in_base + k0;
#pragma unroll
        for (int j = 0; j < NIN; ++j)
            rin[j] = ((in_ok >> j) & 1u) ? *reinterpret_cast<const uint4*>(src0 + in_rel[j]) : make_uint4(0, 0, 0, 0);
        if (with_w) {
            const T* w0 = wg + k0;
#pragma unroll
            for (int j = 0; j < NWT; ++j)
                rwt[j] = (w_rel[j] >= 0) ? *reinterpret_cast<const uint4*>(w0 + w_rel[j]) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](bool with_w) {
#pragma unroll
        for (int j = 0; j < NIN; ++j)
            if (in_pos[j] >= 0) F::lstore_at(in_lds + in_dst[j], rin[j]);
        if (with_w) {
#pragma unroll
            for (int j = 0; j < NWT; ++j)
                if (w_rel[j] >= 0) F::lstore_at(w_lds + w_dst0 + j * (256 / VPP) * ROWW, rwt[j]);
        }
    };
    T* __restrict__ yg = static_cast<T*>(a.y);
    // Tile order.  Plain: block x walks tiles x, x + grid, ...  Bands (a.bands): the blocks of one XCD (block id % 8: the
    // hardware's round-robin, used as an affinity for speed only -- any placement is correct) walk ONE contiguous eighth of
    // the tile raster, so raster neighbours run on the same L2 close in time and the halo a tile shares with them is an L2
    // hit instead of a second HBM fetch (PMC: 336 MB fetched per launch against 268 MB algorithmic on the 1024^2 layer).
    constexpr bool BANDS = conv_bands_ok<T>(GEO, TH, TW, BP);
    int band, xcd, nb8, bslot;                         // band walk only: never read in the other instantiations
    if constexpr (BANDS) { band = (a.ntiles + 7) >> 3; xcd = blockIdx.x & 7; nb8 = gridDim.x >> 3; bslot = blockIdx.x >> 3; }
    auto tile_of = [&](int it) {
        if (!a.bands) { const int t = blockIdx.x + it * (int)gridDim.x; return t < a.ntiles ? t : -1; }
        const int j = bslot + it * nb8, t = xcd * band + j;
        return (j < band && t < a.ntiles) ? t : -1;
    };
    int iter = 0, tile, next_tile = -1;
    if constexpr (BANDS) {
        tile = tile_of(0); next_tile = tile_of(1);
        if (tile < 0) return;
    } else {
        tile = blockIdx.x;
        if (tile >= a.ntiles) return;
    }
    set_tile(tile);
    gload(k_lo, true);
    bool first = true;
    for (;;) {
    const int c_img0 = img0, c_ty0 = ty0, c_tx0 = tx0;   // ... and of the tile being COMPUTED
    for (int k0 = k_lo; k0 < k_hi; k0 += KC) {
        if (!first) __syncthreads();                  // every wave is done reading the previous stage
        const bool ww = first || !w_static;
        if (!(a.dbg & 4) || first) lstore(ww);
        first = false;
        __syncthreads();
        if constexpr (BANDS) {
        if (a.dbg & 2) { if (k0 + KC >= k_hi && next_tile >= 0) set_tile(next_tile); }
        else if (k0 + KC < k_hi) gload(k0 + KC, true);    // in flight during the MFMAs below
        else if (next_tile >= 0) { set_tile(next_tile); gload(k_lo, !w_static); }
        } else {
        if (a.dbg & 2) { if (k0 + KC >= k_hi && tile + (int)gridDim.x < a.ntiles) set_tile(tile + gridDim.x); }
        else if (k0 + KC < k_hi) gload(k0 + KC, true);    // in flight during the MFMAs below
        else if (tile + (int)gridDim.x < a.ntiles) { set_tile(tile + gridDim.x); gload(k_lo, !w_static); }
        }
        if constexpr (UPA) {
            // position-major: each of the 9 patch positions (dy, dx) is read once and feeds every class that has a tap
            // there -- class (py, px) uses tap (a, b) = (dy - py, dx - px) when both are 0 or 1, with the weight tap
            // (3 - py - 2a, 3 - px - 2b) of the 4x4 kernel (same mapping as the per-class GUP kernel).
#pragma unroll
            for (int pos = 0; pos < 9; ++pos) {
                const int dy = pos / 3, dx = pos % 3;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    typename F::frag_t fb[SPW];
#pragma unroll
                    for (int s = 0; s < SPW; ++s) fb[s] = F::ld(in_lds + pl * IN_PLANE + inoff[s][dx] + dy * PWP * ROWI, 0);
#pragma unroll
                    for (int cls = 0; cls < 4; ++cls) {
                        const int py_ = cls >> 1, px_ = cls & 1, ta = dy - py_, tb = dx - px_;
                        if (ta >= 0 && ta <= 1 && tb >= 0 && tb <= 1) {
                            const int tg = (3 - py_ - 2 * ta) * 4 + (3 - px_ - 2 * tb);
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) {
                                const typename F::frag_t fa = F::ld(w_lds + pl * W_PLANE + woff + (tg * BCO + ct * 16) * ROWW, 0);
#pragma unroll
                                for (int s = 0; s < SPW; ++s) acc[cls * CT + ct][s] = F::mma(fa, fb[s], acc[cls * CT + ct][s]);
                            }
                        }
                    }
                }
            }
        } else
        if (!(a.dbg & 1))
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int kk = 0; kk < F::NK; ++kk) {
                typename F::frag_t fa[CT], fb[SPW];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) fa[ct] = F::ld(w_lds + pl * W_PLANE + woff + (t * BCO + ct * 16) * ROWW, kk);
#pragma unroll
                for (int s = 0; s < SPW; ++s) fb[s] = F::ld(in_lds + pl * IN_PLANE + inoff[s][t % TK] + (t / TK) * PWP * ROWI, kk);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int s = 0; s < SPW; ++s) acc[ct][s] = F::mma(fa[ct], fb[s], acc[ct][s]);
            }
        }
    }
    // ---- epilogue: bias, activation, NHWC store.
    // bf16 with >= 32 output channels per block: a lane's natural store is 8 bytes and a wave instruction writes 32-byte
    // pieces (measured: 14 of 43 us on the 256^2 64->64 layer).  Transpose the tile through LDS instead and write whole
    // BCO*2-byte channel rows with 16 bytes per lane.
    if constexpr (SK) {
        // split-K: the raw fp32 accumulators of this K range, kws[split][output pixel][Cout]; a lane's 4 consecutive channels = one 16-byte store
        float* const dst0 = a.kws + (size_t)split * ((size_t)a.B * a.OH * a.OW) * a.Cout + co0 + q * 4;
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const int m = (wave * SPW + s) * 16 + l15;
            const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
            const int b = c_img0 + il, oyc = c_ty0 + r, oxc = c_tx0 + c;
            if (b >= a.B || oyc >= a.OHc || oxc >= a.OWc) continue;
#pragma unroll
            for (int cls = 0; cls < NCL; ++cls) {
                const int oy = UPA ? 2 * oyc + (cls >> 1) : ((GEO == GUP) ? 2 * oyc + py : oyc);
                const int ox = UPA ? 2 * oxc + (cls & 1) : ((GEO == GUP) ? 2 * oxc + px : oxc);
                float* d = dst0 + (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const f32x4 v = acc[cls * CT + ct][s];
                    *reinterpret_cast<float4*>(d + ct * 16) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    } else
    if constexpr (UPA) {
        // the block owns the complete (2 TH) x (2 TW) fine tile: stage it in LDS pixel-major and write whole rows
        static_assert(sizeof(T) == 2, "GUPA is the bf16 path");
        constexpr int OROW = BCO * 2 + 16, VPO = BCO * 2 / 16, FW = 2 * TW, FH = 2 * TH;
        __syncthreads();                                  // every wave is done reading this tile's operands
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const int m = (wave * SPW + s) * 16 + l15;
            const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
                const int fp = (il * FH + 2 * r + (cls >> 1)) * FW + 2 * c + (cls & 1);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const f32x4 v = acc[cls * CT + ct][s];
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(smem + fp * OROW + (ct * 16 + q * 4) * 2) = o;
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 4 * BP * VPO; idx += 256) {
            const int fp = idx / VPO, v = idx % VPO;
            const int il = fp / (FH * FW), fy = (fp / FW) % FH, fx = fp % FW;
            const int b = c_img0 + il, oy = 2 * c_ty0 + fy, ox = 2 * c_tx0 + fx;
            if (b >= a.B || oy >= a.OH || ox >= a.OW) continue;
            T* dst = yg + (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + v * 8;
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(smem + fp * OROW + v * 16);
        }
    } else
    if constexpr (sizeof(T) == 2 && CT >= 2) {
        constexpr int OROW = BCO * 2 + 16, VPR = BCO * 2 / 16;
        __syncthreads();                                  // every wave is done reading this tile's operands
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const int m = (wave * SPW + s) * 16 + l15;
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
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(smem + m * OROW + (ct * 16 + q * 4) * 2) = o;
            }
        }
        __syncthreads();
        if (!(a.dbg & 8))
        for (int idx = tid; idx < BP * VPR; idx += 256) {
            const int m = idx / VPR, v = idx % VPR;
            const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
            const int b = c_img0 + il, oyc = c_ty0 + r, oxc = c_tx0 + c;
            if (b >= a.B || oyc >= a.OHc || oxc >= a.OWc) continue;
            const int oy = (GEO == GUP) ? 2 * oyc + py : oyc, ox = (GEO == GUP) ? 2 * oxc + px : oxc;
            const size_t doff = (((size_t)b * a.OH + oy) * a.OW + ox) * a.Cout + co0 + v * 8;
            uint4 val = *reinterpret_cast<const uint4*>(smem + m * OROW + v * 16);
            if (GEO == G3X3 && a.mask) val = lrelu_mask_bf16x8(val, *reinterpret_cast<const uint4*>(static_cast<const T*>(a.mask) + doff));
            *reinterpret_cast<uint4*>(yg + doff) = val;
        }
    } else
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int m = (wave * SPW + s) * 16 + l15;
        const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
        const int b = c_img0 + il, oyc = c_ty0 + r, oxc = c_tx0 + c;
        if (b >= a.B || oyc >= a.OHc || oxc >= a.OWc || (a.dbg & 8)) continue;
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
                if (GEO == G3X3 && a.mask) {
                    const float4 mv = *reinterpret_cast<const float4*>(static_cast<const float*>(a.mask) + (dst - yg) + ct * 16);
                    v[0] *= lrelu_slope(mv.x); v[1] *= lrelu_slope(mv.y); v[2] *= lrelu_slope(mv.z); v[3] *= lrelu_slope(mv.w);
                }
                *reinterpret_cast<float4*>(dst + ct * 16) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                if (GEO == G3X3 && a.mask) {                               // on the ROUNDED value, as the separate pass would
                    const uint2 mv = *reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(a.mask) + (dst - yg) + ct * 16);
                    const uint4 r = lrelu_mask_bf16x8(make_uint4(o.x, o.y, 0u, 0u), make_uint4(mv.x, mv.y, 0u, 0u));
                    o.x = r.x; o.y = r.y;
                }
                *reinterpret_cast<uint2*>(dst + ct * 16) = o;
            }
        }
    }
    if constexpr (BANDS) {
        tile = next_tile; next_tile = tile_of(++iter + 1);
        if (tile < 0) break;
    } else {
        tile += gridDim.x;
        if (tile >= a.ntiles) break;
    }
#pragma unroll
    for (int ct = 0; ct < NCL * CT; ++ct)
#pragma unroll
        for (int s = 0; s < SPW; ++s) acc[ct][s] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

template <typename T, int KC, int GEO, int TH, int TW, int BP, int CT, bool SK = false>
static int launch_conv(ConvArgs& a, int ngroups, hipStream_t st) {
    constexpr int FK = (sizeof(T) == 2 && KC > 32) ? 32 : KC, NPL = KC / FK;
    using F = Frag<T, FK>;
    constexpr int IS = Geo<GEO>::IS, TK = Geo<GEO>::TK;
    constexpr int PH = (TH - 1) * IS + TK, PW = (TW - 1) * IS + TK, NI = BP / (TH * TW);
    constexpr int IN_BYTES = (NI * PH * F::pad_pw(PW, IS) * F::rowb_in(IS) + 15) / 16 * 16;
    constexpr int NTW = GEO == GUPA ? 16 : TK * TK;
    constexpr int OPER = NPL * (IN_BYTES + NTW * CT * 16 * F::rowb_w());
    constexpr int OUTB = GEO == GUPA ? 4 * BP * (CT * 32 + 16)                      // the whole fine tile
                                     : ((sizeof(T) == 2 && CT >= 2) ? BP * (CT * 32 + 16) : 0);   // LDS-transposed bf16 epilogue tile
    constexpr int LDS = OPER > OUTB ? OPER : OUTB;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv_kernel<T, KC, GEO, TH, TW, BP, CT, SK>;
    sgx_lds_opt_in<conv_kernel<T, KC, GEO, TH, TW, BP, CT, SK>>(LDS);
    a.ntiles = ngroups * a.tiles_y * a.tiles_x;
    // the ablation switches give WRONG results by design: only a probe build (make PROBE=1 -> -DSGX_PROBE_BUILD) reads them
#ifdef SGX_PROBE_BUILD
    static const int dbg = [] { const char* e = getenv("SGX_CONV_DBG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
#else
    a.dbg = 0;
#endif
    // persistent grid: exactly as many blocks as are resident at once (register- and LDS-limited; asked of the runtime
    // for this instantiation), never more than tiles.  More blocks than that would run as a second, under-occupied round.
    static const int resident = [] {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(conv_kernel<T, KC, GEO, TH, TW, BP, CT, SK>), 256, LDS) != hipSuccess || nb < 1) nb = 1;
        const char* e = getenv("SGX_CONV_PERCU");
        if (e && atoi(e) > 0 && atoi(e) < nb) nb = atoi(e);
        return nb;
    }();
    const int ncu = sgx_ncu();
    const int per_cu = resident;
    const int yz = (a.Cout / (CT * 16)) * Geo<GEO>::NCLS;
    int gx = (ncu * per_cu + yz - 1) / yz;
    if (gx > a.ntiles) gx = a.ntiles;
    if (gx < 1) gx = 1;
    static const int bands_on = [] { const char* e = getenv("SGX_TILE_BANDS"); return e ? atoi(e) : 1; }();   // measured (round 2): halo over-fetch gone (PMC), 80.8 vs 81.2 ms at batch 32, nothing at batch 4
    a.bands = 0;
    if (conv_bands_ok<T>(GEO, TH, TW, BP) && bands_on && gx >= 64 && a.ntiles >= 8 * gx) {   // enough tiles per band for the order to matter
        gx = gx / 8 * 8;
        a.bands = 1;
    }
    int ksp = 1;
    if constexpr (SK) {
        ksp = a.ksplit;
        SGX_REQUIRE(ksp > 1 && a.kws && a.Cin / KC >= ksp, SGX_EINVAL, "conv: %d K-chunks do not split %d ways", a.Cin / KC, ksp);
        gx = a.ntiles; a.bands = 0;                     // (small launches by construction: one tile per block)
    }
    dim3 grid((unsigned)gx, (unsigned)(a.Cout / (CT * 16)), Geo<GEO>::NCLS * ksp);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, st, a);
    SGX_LAUNCH_CHECK("conv_kernel");
    return 0;
}

// ---- launch configuration: (pixels per block, output-channel sub-tiles per block, tile shape).  Big tiles maximise
// operand reuse; when the problem is small (low resolution x batch 4) smaller tiles are chosen until the grid has at
// least two blocks per CU.  Exposed through sgx_conv_config so callers can name the instantiation a launch uses.
struct ConvCfg { int bp, ct, th, tw, ni; };
static void tile_shape(int bp, int ohc, int owc, ConvCfg& c) {
    if (ohc >= 16 && owc >= 16) { c.th = bp / 16; c.tw = 16; }
    else if (ohc >= 8 && owc >= 8) { c.th = 8; c.tw = 8; }
    else { c.th = 4; c.tw = 4; }
    c.ni = bp / (c.th * c.tw);
}
static ConvCfg pick_cfg(int geo, int B, int ohc, int owc, int Cout) {
    static const int cand_su[][2] = {{256, 4}, {256, 2}, {128, 4}, {256, 1}, {128, 2}, {128, 1}, {64, 2}, {64, 1}};
    static const int cand_d[][2] = {{128, 2}, {128, 1}, {64, 2}, {64, 1}};
    static const int cand_ua[][2] = {{256, 2}, {256, 1}, {128, 2}, {128, 1}, {64, 2}, {64, 1}};   // 4 classes of accumulators: ct <= 2
    const int (*cand)[2] = geo == GDOWN ? cand_d : (geo == GUPA ? cand_ua : cand_su);
    const int n = geo == GDOWN ? 4 : (geo == GUPA ? 6 : 8);
    ConvCfg best{0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        static const int max_ct = [] { const char* e = getenv("SGX_CONV_MAXCT"); return e ? atoi(e) : 4; }();   // tuning knob
        static const int max_bp = [] { const char* e = getenv("SGX_CONV_MAXBP"); return e ? atoi(e) : 256; }();
        if (cand[i][1] > max_ct || cand[i][0] > max_bp) continue;
        if (Cout % (16 * cand[i][1]) != 0) continue;
        ConvCfg c{cand[i][0], cand[i][1], 0, 0, 0};
        tile_shape(c.bp, ohc, owc, c);
        if (c.ni < 1) continue;
        best = c;
        const long blocks = (long)((B + c.ni - 1) / c.ni) * ((ohc + c.th - 1) / c.th) * ((owc + c.tw - 1) / c.tw) *
                            (Cout / (16 * c.ct)) * (geo == GUP ? 4 : 1);
        static const long min_blocks = [] { const char* e = getenv("SGX_CONV_MINBLOCKS"); return e && atoi(e) > 0 ? atol(e) : 512L; }();
        if (blocks >= min_blocks) break;
    }
    return best;
}

// Deep K-chunks (128 input channels per LDS stage) for the layers whose launch is 64-pixel tiles: many input channels, few
// pixels -- their time is the number of load/barrier/MFMA stages, not bytes or FLOPs.  SGX_CONV_DEEPK=0 switches it off.
static bool conv_deep_k(int geo, int B, int ohc, int owc, int Cin, int Cout) {
    static const int on = [] { const char* e = getenv("SGX_CONV_DEEPK"); return e ? atoi(e) : 1; }();
    if (!on || Cin % 128 != 0 || geo == GDOWN) return false;
    const ConvCfg c = pick_cfg(geo, B, ohc, owc, Cout);
    return c.bp == 64 && (c.ct == 1 || c.ct == 2);
}

template <typename T, int KC, int GEO, int BP, int CT, bool SK = false>
static int dispatch_tile(ConvArgs& a, const ConvCfg& c, hipStream_t st) {
    a.tiles_y = (a.OHc + c.th - 1) / c.th; a.tiles_x = (a.OWc + c.tw - 1) / c.tw;
    const int ngroups = (a.B + c.ni - 1) / c.ni;
    if (c.tw == 16) return launch_conv<T, KC, GEO, BP / 16, 16, BP, CT, SK>(a, ngroups, st);
    if (c.tw == 8) return launch_conv<T, KC, GEO, 8, 8, BP, CT, SK>(a, ngroups, st);
    return launch_conv<T, KC, GEO, 4, 4, BP, CT, SK>(a, ngroups, st);
}

// (SK: the split-K instantiations exist for the 64- and 128-pixel tiles with 16 / 32 output channels only -- conv_splitk_plan asks for nothing else)
template <typename T, int KC, int GEO, bool SK = false>
static int dispatch_cfg(ConvArgs& a, hipStream_t st) {
    const ConvCfg c = pick_cfg(GEO, a.B, a.OHc, a.OWc, a.Cout);
    SGX_REQUIRE(c.bp != 0, SGX_EUNSUPPORTED, "conv: no launch configuration for Cout=%d", a.Cout);
    if constexpr (!SK) {
    if constexpr (GEO != GDOWN) {
        if constexpr (GEO != GUPA) {
            if (c.bp == 256 && c.ct == 4) return dispatch_tile<T, KC, GEO, 256, 4>(a, c, st);
            if (c.bp == 128 && c.ct == 4) return dispatch_tile<T, KC, GEO, 128, 4>(a, c, st);
        }
        if (c.bp == 256 && c.ct == 2) return dispatch_tile<T, KC, GEO, 256, 2>(a, c, st);
        if (c.bp == 256 && c.ct == 1) return dispatch_tile<T, KC, GEO, 256, 1>(a, c, st);
    }
    } else {
        SGX_REQUIRE(c.bp <= 128 && c.ct <= 2, SGX_EUNSUPPORTED, "conv split-K: %d-pixel x %d-channel tile", c.bp, 16 * c.ct);
    }
    if (c.bp == 128 && c.ct == 2) return dispatch_tile<T, KC, GEO, 128, 2, SK>(a, c, st);
    if (c.bp == 128 && c.ct == 1) return dispatch_tile<T, KC, GEO, 128, 1, SK>(a, c, st);
    if (c.bp == 64 && c.ct == 2) return dispatch_tile<T, KC, GEO, 64, 2, SK>(a, c, st);
    return dispatch_tile<T, KC, GEO, 64, 1, SK>(a, c, st);
}

template <int GEO, bool SK = false>
static int dispatch_conv(ConvArgs& a, int dtype, hipStream_t st) {
    SGX_REQUIRE(a.Cout % 16 == 0 && a.Cin % 16 == 0, SGX_EUNSUPPORTED, "conv: channels must be multiples of 16 (Cin=%d Cout=%d)", a.Cin, a.Cout);
    SGX_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0, SGX_EINVAL, "conv: bad shape");
    if constexpr (!SK) {
        if (dtype == SGX_F32) return dispatch_cfg<float, 16, GEO>(a, st);
    }
    if (dtype == SGX_BF16) {
        if constexpr (GEO != GDOWN) {                          // (the stride-2 patch x 4 planes does not fit the LDS)
            if (conv_deep_k(GEO, a.B, a.OHc, a.OWc, a.Cin, a.Cout)) {
                const ConvCfg c = pick_cfg(GEO, a.B, a.OHc, a.OWc, a.Cout);
                if constexpr (!SK) {
                    if (c.ct == 2) return dispatch_tile<bf16_t, 128, GEO, 64, 2>(a, c, st);
                } else {                                       // (its split-K form spills 300-440 bytes per lane: not built, never planned)
                    SGX_REQUIRE(c.ct == 1, SGX_EUNSUPPORTED, "conv split-K: deep K-chunks with 32 output channels per block");
                }
                return dispatch_tile<bf16_t, 128, GEO, 64, 1, SK>(a, c, st);
            }
        }
        if constexpr (GEO == GUP) {
            static const int fuse = [] { const char* e = getenv("SGX_CONV_UPA"); return e ? atoi(e) : 1; }();   // A/B switch
            if (fuse) {
                if (a.Cin % 32 == 0) return dispatch_cfg<bf16_t, 32, GUPA, SK>(a, st);
                return dispatch_cfg<bf16_t, 16, GUPA, SK>(a, st);
            }
        }
        if (a.Cin % 32 == 0) return dispatch_cfg<bf16_t, 32, GEO, SK>(a, st);
        return dispatch_cfg<bf16_t, 16, GEO, SK>(a, st);
    }
    SGX_REQUIRE(false, SGX_EINVAL, "conv: bad dtype %d", dtype);
}

// ---------------------------------------------------------------------------------------------------------------- split-K (round 6)
// y[p][c] = mask(bf16(act(sum_s kws[s][p][c] + bias[c]))): the splits in a FIXED order (deterministic), then exactly the epilogue of conv_kernel
// (bias, activation, one rounding to bf16, the output mask on the rounded value).  8 channels per lane.
__global__ __launch_bounds__(256) void conv_splitk_finish(const float* __restrict__ kws, int ksplit, size_t npix, int Cout, const float* __restrict__ bias,
                                                          int act, const bf16_t* __restrict__ mask, bf16_t* __restrict__ y) {
    const size_t nvec = npix * Cout / 8, stride = npix * Cout;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int c0 = (int)((i * 8) % Cout);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    for (int s = 0; s < ksplit; ++s) {
        const float4 a0 = *reinterpret_cast<const float4*>(kws + s * stride + i * 8), a1 = *reinterpret_cast<const float4*>(kws + s * stride + i * 8 + 4);
        v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
    }
    if (bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bias[c0 + j];
    }
    if (act == SGX_ACT_LRELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lrelu(v[j]);
    }
    uint4 o = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    if (mask) o = lrelu_mask_bf16x8(o, *reinterpret_cast<const uint4*>(mask + i * 8));
    *reinterpret_cast<uint4*>(y + i * 8) = o;
}
// How many ways a bf16 launch splits its reduction (0 = not at all): only launches that stay with this file's kernel, only when their grid
// leaves most of the chip idle, a divisor of the K-chunk count, as many blocks as ~2 per CU, at most 8 (the finish pass reads ksplit partials).
static int conv_splitk_plan(int geo, int B, int H, int W, int Cin, int Cout, int dtype) {
    static const int on = [] { const char* e = getenv("SGX_CONV_SPLITK"); const int v = e ? atoi(e) : 1; return v < 0 ? 1 : v; }();      // A/B switch (0 = off; n > 1 = at most n ways)
    if (!on || dtype != SGX_BF16 || geo < 0 || geo > 2 || Cin % 16 || Cout % 16) return 0;
    // Measured alone on the MI355X (tools/splitk_probe.py, batch 4, 512 channels; both launches of the split form): stride-2 16^2 -> 8^2 23.1 ->
    // 17.0 us, 8^2 -> 4^2 21.2 -> 11.4 us; 3x3 at 4^2 / 8^2 11.3 -> 9.7 / 11.7 -> 10.8 us; transposed 4^2 -> 8^2 8.2 -> 9.1 us (worse).  The 3x3 and
    // transposed launches already run deep (128-channel) K-chunks -- 4 of them; the stride-2 geometry cannot (its patch x 4 planes does not
    // fit the LDS) and walks 16 chunks of 32 channels, each behind a global-load latency: that is the one split-K shortens enough to pay for
    // the finishing launch.  (The split-K forms of the other two geometries were built and measured -- profiles/r06_splitk_probe.txt, commit a6b45ff --
    // and are no longer compiled: 84 instantiations and two minutes of build time for nothing.)
    if (geo != GDOWN) return 0;
    if (sgx_conv2_takes(geo, B, H, W, Cin, Cout)) return 0;
    const int ohc = geo == GDOWN ? H / 2 : H, owc = geo == GDOWN ? W / 2 : W;
    // the instantiation dispatch_conv picks for this shape: deep K-chunks first, else (transposed) the all-class kernel
    const bool deep = geo != GDOWN && conv_deep_k(geo, B, ohc, owc, Cin, Cout);
    const ConvCfg c = pick_cfg((geo == GUP && !deep) ? GUPA : geo, B, ohc, owc, Cout);
    if (!c.bp || c.bp > 128 || c.ct > 2 || (deep && c.ct != 1)) return 0;       // (the tiles the split-K instantiations exist for)
    const int kc = deep ? 128 : (Cin % 32 == 0 ? 32 : 16);
    const int nk = Cin / kc;
    const long blocks = (long)((B + c.ni - 1) / c.ni) * ((ohc + c.th - 1) / c.th) * ((owc + c.tw - 1) / c.tw) * (Cout / (16 * c.ct)) * ((geo == GUP && deep) ? 4 : 1);
    const int ncu = sgx_ncu();
    if (nk < 4 || blocks > ncu) return 0;                                         // (at most one block per CU without the split)
    int want = (int)((2L * ncu + blocks - 1) / blocks);
    const int cap = on > 1 ? on : 8;
    if (want > cap) want = cap;
    if (want > nk / 2) want = nk / 2;                                             // (>= 2 chunks per block: the register prefetch still overlaps one)
    return want >= 2 ? want : 0;                                                  // (uneven ranges are fine: split s owns chunks [s nk / ks, (s + 1) nk / ks))
}
extern "C" size_t sgx_conv_splitk_ws_bytes(int geo, int B, int H, int W, int Cin, int Cout, int dtype) {
    const int ks = conv_splitk_plan(geo, B, H, W, Cin, Cout, dtype);
    if (!ks) return 0;
    const size_t opix = geo == GDOWN ? (size_t)B * (H / 2) * (W / 2) : (geo == GUP ? (size_t)B * 4 * H * W : (size_t)B * H * W);
    return (size_t)ks * opix * Cout * sizeof(float);
}
// sgx_conv3x3 / sgx_conv4x4s2_down / sgx_conv4x4s2_up (geo 0 / 1 / 2) of a shape for which sgx_conv_splitk_ws_bytes is not 0, with the
// reduction split over blocks: the partials go through `ws`, a second (tiny) launch finishes.  `mask`: geo 0 only.
extern "C" int sgx_conv_splitk(int geo, const void* x, const void* w, const float* bias, void* y, const void* mask, int B, int H, int W, int Cin, int Cout,
                               int act, int dtype, void* ws, size_t ws_bytes, void* stream) {
    const int ks = conv_splitk_plan(geo, B, H, W, Cin, Cout, dtype);
    SGX_REQUIRE(ks > 1, SGX_EUNSUPPORTED, "conv_splitk: this shape does not split (geo %d B%d %dx%d %d->%d)", geo, B, H, W, Cin, Cout);
    SGX_REQUIRE(x && w && y && ws && ws_bytes >= sgx_conv_splitk_ws_bytes(geo, B, H, W, Cin, Cout, dtype), SGX_EWORKSPACE, "conv_splitk: workspace");
    SGX_REQUIRE(geo == G3X3 || !mask, SGX_EINVAL, "conv_splitk: the output mask belongs to the 3x3 geometry");
    SGX_REQUIRE(geo != GDOWN || (H % 2 == 0 && W % 2 == 0), SGX_EINVAL, "conv_splitk: odd input size");
    hipStream_t st = (hipStream_t)stream;
    const int OH = geo == GDOWN ? H / 2 : (geo == GUP ? 2 * H : H), OW = geo == GDOWN ? W / 2 : (geo == GUP ? 2 * W : W);
    ConvArgs a{x, w, nullptr, y, B, H, W, OH, OW, geo == GUP ? H : OH, geo == GUP ? W : OW, Cin, Cout, SGX_ACT_NONE, 0, 0, 0, nullptr};
    a.ksplit = ks; a.kws = static_cast<float*>(ws);
    const double taps = geo == G3X3 ? 9.0 : 16.0, opix = (double)B * OH * OW, macs = geo == GUP ? taps / 4.0 : taps;
    SGX_NOTE(2.0 * macs * Cin * Cout * opix, 2.0 * ((double)B * H * W * Cin + opix * Cout + taps * Cin * Cout) + 4.0 * ks * opix * Cout, "conv%c/k%d B%d %dx%d %d->%d",
             geo == G3X3 ? 'S' : (geo == GDOWN ? 'D' : 'U'), ks, B, H, W, Cin, Cout);
    int rc;
    SGX_REQUIRE(geo == GDOWN, SGX_EUNSUPPORTED, "conv_splitk: only the stride-2 geometry has split-K kernels");
    rc = dispatch_conv<GDOWN, true>(a, dtype, st);
    if (rc) return rc;
    const size_t npix = (size_t)B * OH * OW, nvec = npix * Cout / 8;
    SGX_NOTE(0.0, 4.0 * ks * npix * Cout + 2.0 * npix * Cout, "conv_splitk_finish %zux%d", npix, Cout);
    hipLaunchKernelGGL(conv_splitk_finish, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, (const float*)ws, ks, npix, Cout, bias, act,
                       static_cast<const bf16_t*>(mask), static_cast<bf16_t*>(y));
    SGX_LAUNCH_CHECK("conv_splitk_finish");
    return 0;
}

extern "C" int sgx_conv_config(int geo, int B, int H, int W, int Cin, int Cout, int dtype, int* cfg5) {
    SGX_REQUIRE(geo >= 0 && geo <= 2 && cfg5, SGX_EINVAL, "conv_config: bad args");
    const int ohc = geo == GDOWN ? H / 2 : H, owc = geo == GDOWN ? W / 2 : W;
    const ConvCfg c = pick_cfg(geo, B, ohc, owc, Cout);
    cfg5[0] = dtype == SGX_F32 ? 16 : (conv_deep_k(geo, B, ohc, owc, Cin, Cout) ? 128 : (Cin % 32 == 0 ? 32 : 16));
    cfg5[1] = c.th; cfg5[2] = c.tw; cfg5[3] = c.bp; cfg5[4] = c.ct;
    return 0;
}

extern "C" int sgx_conv3x3(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin,
                           int Cout, int act, const void* mask, int dtype, void* stream) {
    ConvArgs a{x, w, bias, y, B, H, W, H, W, H, W, Cin, Cout, act, 0, 0, 0, mask};
    const double es = dtype == SGX_F32 ? 4 : 2;
    SGX_NOTE(2.0 * 9 * Cin * Cout * B * H * W, es * ((double)B * H * W * (Cin + Cout) + 9.0 * Cin * Cout), "convS B%d %dx%d %d->%d", B, H, W, Cin, Cout);
    if (dtype == SGX_BF16) {                                   // second-generation kernel (conv2.hip) where it applies
        int launched = 0;
        const int rc = sgx_conv2_try(G3X3, x, w, bias, y, B, H, W, Cin, Cout, act, mask, -1, (hipStream_t)stream, &launched);
        if (rc || launched) return rc;
    }
    return dispatch_conv<G3X3>(a, dtype, (hipStream_t)stream);
}

extern "C" int sgx_conv4x4s2_down(const void* x, const void* w, const float* bias, void* y, int B, int H, int W,
                                  int Cin, int Cout, int act, int dtype, void* stream) {
    SGX_REQUIRE(H % 2 == 0 && W % 2 == 0, SGX_EINVAL, "conv4x4s2_down: odd input size");
    ConvArgs a{x, w, bias, y, B, H, W, H / 2, W / 2, H / 2, W / 2, Cin, Cout, act, 0, 0};
    const double es = dtype == SGX_F32 ? 4 : 2;
    SGX_NOTE(2.0 * 16 * Cin * Cout * B * (H / 2) * (W / 2), es * ((double)B * H * W * (Cin + Cout / 4.0) + 16.0 * Cin * Cout), "convD B%d %dx%d %d->%d", B, H, W, Cin, Cout);
    if (dtype == SGX_BF16) {
        int launched = 0;
        const int rc = sgx_conv2_try(GDOWN, x, w, bias, y, B, H, W, Cin, Cout, act, nullptr, -1, (hipStream_t)stream, &launched);
        if (rc || launched) return rc;
    }
    return dispatch_conv<GDOWN>(a, dtype, (hipStream_t)stream);
}

extern "C" int sgx_conv4x4s2_up(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout,
                                int dtype, void* stream) {
    ConvArgs a{x, w, nullptr, y, B, H, W, 2 * H, 2 * W, H, W, Cin, Cout, SGX_ACT_NONE, 0, 0};
    const double es = dtype == SGX_F32 ? 4 : 2;
    SGX_NOTE(2.0 * 16 * Cin * Cout * B * H * W, es * ((double)B * H * W * (Cin + 4.0 * Cout) + 16.0 * Cin * Cout), "convU B%d %dx%d %d->%d", B, H, W, Cin, Cout);
    if (dtype == SGX_BF16) {
        int launched = 0;
        const int rc = sgx_conv2_try(GUP, x, w, nullptr, y, B, H, W, Cin, Cout, SGX_ACT_NONE, nullptr, -1, (hipStream_t)stream, &launched);
        if (rc || launched) return rc;
    }
    return dispatch_conv<GUP>(a, dtype, (hipStream_t)stream);
}

extern "C" int sgx_conv_variant(int geo, const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin,
                                int Cout, int act, int dtype, int variant, void* stream) {
    SGX_REQUIRE(geo >= 0 && geo <= 2, SGX_EINVAL, "conv_variant: bad geometry %d", geo);
    SGX_REQUIRE(geo != GDOWN || (H % 2 == 0 && W % 2 == 0), SGX_EINVAL, "conv_variant: odd input size");
    SGX_REQUIRE(geo != GUP || (!bias && act == SGX_ACT_NONE), SGX_EINVAL, "conv_variant: the transposed convolution has no fused bias / activation");
    const int OH = geo == GDOWN ? H / 2 : (geo == GUP ? 2 * H : H), OW = geo == GDOWN ? W / 2 : (geo == GUP ? 2 * W : W);
    ConvArgs a{x, w, bias, y, B, H, W, OH, OW, geo == GUP ? H : OH, geo == GUP ? W : OW, Cin, Cout, act, 0, 0};
    const double es = dtype == SGX_F32 ? 4 : 2, taps = geo == G3X3 ? 9 : 16, npix = (double)B * (geo == GDOWN ? OH * OW : H * W);
    SGX_NOTE(2.0 * taps * Cin * Cout * npix, es * ((double)B * (H * W * Cin + OH * OW * Cout) + taps * Cin * Cout), "conv%c/v%d B%d %dx%d %d->%d", "SDU"[geo], variant, B, H, W, Cin, Cout);
    if (variant == 0) {
        if (geo == G3X3) return dispatch_conv<G3X3>(a, dtype, (hipStream_t)stream);
        if (geo == GDOWN) return dispatch_conv<GDOWN>(a, dtype, (hipStream_t)stream);
        return dispatch_conv<GUP>(a, dtype, (hipStream_t)stream);
    }
    SGX_REQUIRE(dtype == SGX_BF16 && (variant == 4 || variant == 8 || (variant >= 20 && variant <= 29) || (variant >= 30 && variant <= 36)), SGX_EINVAL,
                "conv_variant: variant %d needs bf16 and 4 or 8 waves (or 20..29: a wide-block configuration, 30..36: a conv3_kernel configuration)", variant);
    int launched = 0;
    const int rc = sgx_conv2_try(geo, x, w, bias, y, B, H, W, Cin, Cout, act, nullptr, variant, (hipStream_t)stream, &launched);
    SGX_REQUIRE(rc || launched, SGX_EUNSUPPORTED, "conv_variant: shape not covered by the second-generation kernel");
    return rc;
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
    int want_bias;                                        // also emit column sums of the n side (the bias gradient)
    // direct epilogue (one pixel split): the block folds / scales / re-lays-out its own tile and writes dW (and db) itself
    int direct; float* dw; float* db; int O, I, mode, transposed, flip_t, accumulate; float scale;
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
    __device__ static __forceinline__ frag_t ones() { return 1.f; }
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
    __device__ static __forceinline__ frag_t ones() { return (s16x4){0x3f80, 0x3f80, 0x3f80, 0x3f80}; }
};

// LDS transpose read (gfx950 ds_read_b64_tr_b16): within a 16-lane group, lane i supplies the 8-byte address of row i/4,
// column block i%4 of a [4 rows][16 columns] bf16 tile and receives column i of it (4 rows) -- exactly the K-major MFMA
// operand that a pixel-major (NHWC) tile cannot provide with plain reads.
__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}
__device__ __forceinline__ bf16x8 cat8(s16x4 lo, s16x4 hi) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// LDS pixel pitch of an operand tile.  Transpose-read path with pixels ONE apart (a half-wave reads 8 consecutive pixels x
// 32 bytes): 16 channels unpadded (32 B: 256 contiguous bytes), 32 channels 96 B (chunk i at dword 24i: all 64 banks once);
// the padded 48 / 80-byte pitches of WFrag are the conflict-free choice for the scalar path and for lanes TWO pixels apart
// (the stride-2 fine side), and ran at 46-53 % bank conflicts on the one-apart transpose reads (r02_b_pmc_mfma_bf16_b4.json).
template <typename T>
constexpr int wgrad_pitch(int ch, bool tr, int is) {
    return (tr && is == 1) ? (ch == 16 ? 32 : ch * 2 + 32) : WFrag<T>::row_bytes(ch);
}

template <typename T, int GEO, int TH, int TW, int BP, int NSUB, int KSUB, bool TR>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    using F = WFrag<T>;
    constexpr int IS = Geo<GEO>::IS, TK = Geo<GEO>::TK, NT = TK * TK;
    constexpr int PH = (TH - 1) * IS + TK, PW = (TW - 1) * IS + TK;
    constexpr int NI = BP / (TH * TW);
    constexpr int NCH = NSUB * 16, KCH = KSUB * 16;
    constexpr int NROW = wgrad_pitch<T>(NCH, TR, 1), KROW = wgrad_pitch<T>(KCH, TR, IS);
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

    // Bias gradient = column sums of the n side: one extra MFMA per k-step against a tile of ones, on the wave with the
    // fewest taps, in the blocks of k-block 0 only.  Comes out of the same split partials, summed by the finish kernel.
    const bool do_bias = a.want_bias && (blockIdx.x % kblocks) == 0 && wave == 3;
    f32x4 bacc[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j) bacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Global -> LDS staging is software pipelined through registers: the loads of the NEXT tile are issued before the
    // MFMAs of the current one.
    constexpr int NNV = (BP * (NCH / VE) + 255) / 256, NKV = (NI * PH * PW * (KCH / VE) + 255) / 256;
    uint4 rn[NNV], rk[NKV];
    auto gload = [&](int tile) {
        int bx = tile;
        const int tx_i = bx % a.tiles_x; bx /= a.tiles_x;
        const int ty_i = bx % a.tiles_y;
        const int img0 = (bx / a.tiles_y) * NI;
        const int ty0 = ty_i * TH, tx0 = tx_i * TW;
        const int iy0 = ty0 * IS - 1, ix0 = tx0 * IS - 1;
#pragma unroll
        for (int j = 0; j < NNV; ++j) {                   // n-side tile: BP pixels x NCH channels
            const int idx = tid + j * 256;
            const int v = idx % (NCH / VE), m = idx / (NCH / VE);
            const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
            const int b = img0 + il, gy = ty0 + r, gx = tx0 + c;
            const bool ok = (idx < BP * (NCH / VE)) && (b < a.B) && (gy < a.Hn) && (gx < a.Wn);
            const T* src = ng + (((size_t)b * a.Hn + gy) * a.Wn + gx) * a.Cn + n0 + v * VE;
            rn[j] = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NKV; ++j) {                   // k-side patch with halo: NI*PH*PW pixels x KCH channels
            const int idx = tid + j * 256;
            const int v = idx % (KCH / VE), pixel = idx / (KCH / VE);
            const int il = pixel / (PH * PW), rem = pixel % (PH * PW);
            const int gy = iy0 + rem / PW, gx = ix0 + rem % PW, b = img0 + il;
            const bool ok = (idx < NI * PH * PW * (KCH / VE)) && (b < a.B) && ((unsigned)gy < (unsigned)a.Hk) && ((unsigned)gx < (unsigned)a.Wk);
            const T* src = kg + (((size_t)b * a.Hk + gy) * a.Wk + gx) * a.Ck + kc0 + v * VE;
            rk[j] = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < NNV; ++j) {
            const int idx = tid + j * 256;
            if (idx < BP * (NCH / VE)) *reinterpret_cast<uint4*>(n_lds + (idx / (NCH / VE)) * NROW + (idx % (NCH / VE)) * 16) = rn[j];
        }
#pragma unroll
        for (int j = 0; j < NKV; ++j) {
            const int idx = tid + j * 256;
            if (idx < NI * PH * PW * (KCH / VE)) *reinterpret_cast<uint4*>(k_lds + (idx / (KCH / VE)) * KROW + (idx % (KCH / VE)) * 16) = rk[j];
        }
    };
    if ((int)blockIdx.y < a.ntiles) gload(blockIdx.y);
    for (int tile = blockIdx.y; tile < a.ntiles; tile += gridDim.y) {
        __syncthreads();                                  // every wave is done reading the previous tile
        lstore();
        __syncthreads();
        if (tile + (int)gridDim.y < a.ntiles) gload(tile + gridDim.y);   // in flight during the MFMAs below
        if constexpr (TR) {
            // bf16, 32 pixels per MFMA (v_mfma_f32_16x16x32_bf16): two transpose reads per operand
            for (int ks = 0; ks < BP / 32; ++ks) {
                int noff2[2], koff2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m = ks * 32 + h * 16 + q * 4 + (l15 >> 2);
                    const int il = m / (TH * TW), r = (m / TW) % TH, c = m % TW;
                    noff2[h] = m * NROW + (l15 & 3) * 8;
                    koff2[h] = ((il * PH + r * IS) * PW + c * IS) * KROW + (l15 & 3) * 8;
                }
                bf16x8 fa8[NSUB];
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns)
                    fa8[ns] = cat8(lds_tr16(n_lds + noff2[0] + ns * 32), lds_tr16(n_lds + noff2[1] + ns * 32));
                if (do_bias) {
                    const bf16x8 one8 = cat8(WFrag<bf16_t>::ones(), WFrag<bf16_t>::ones());
#pragma unroll
                    for (int ns = 0; ns < NSUB; ++ns) bacc[ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa8[ns], one8, bacc[ns], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int t = wave + 4 * i;
                    if (t < NT) {
                        const int toff = ((t / TK) * PW + (t % TK)) * KROW;
#pragma unroll
                        for (int kk = 0; kk < KSUB; ++kk) {
                            const bf16x8 fb8 = cat8(lds_tr16(k_lds + toff + koff2[0] + kk * 32), lds_tr16(k_lds + toff + koff2[1] + kk * 32));
#pragma unroll
                            for (int ns = 0; ns < NSUB; ++ns)
                                acc[i][ns][kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa8[ns], fb8, acc[i][ns][kk], 0, 0, 0);
                        }
                    }
                }
            }
        } else
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
            if (do_bias) {
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns) bacc[ns] = F::mma(fa[ns], F::ones(), bacc[ns]);
            }
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
    if (a.direct) {
        // One pixel split: no partials.  The 4 waves hold different taps of the same NCH x KCH channel tile; gather them in
        // LDS, then every thread produces parameter-layout outputs: the 3x3 fold of the 4x4 taps (stride-2 pair), the
        // equalized-lr scale, the (o,i) orientation, optionally += into the existing gradient.  Same arithmetic and
        // association as wgrad_finish_kernel with one split, so both paths give the same bits.
        constexpr int KP = KCH + 1;
        float* accl = reinterpret_cast<float*>(smem);                 // [NT][NCH][KP]
        __syncthreads();                                              // operand tiles are dead
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + 4 * i;
            if (t < NT) {
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
                    for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
                        for (int r = 0; r < 4; ++r) accl[(t * NCH + ns * 16 + q * 4 + r) * KP + kk * 16 + l15] = acc[i][ns][kk][r];
            }
        }
        __syncthreads();
        const float c = a.scale * (a.mode == SGX_PACK_D ? 0.25f : 1.f);
        for (int idx = tid; idx < NCH * KCH * 9; idx += 256) {
            const int tap = idx % 9, kl = (idx / 9) % KCH, nl = idx / (9 * KCH);
            const int n = n0 + nl, k = kc0 + kl;
            const int o = a.transposed ? k : n, i = a.transposed ? n : k;
            if (i >= a.I || o >= a.O) continue;                       // padded channels carry no parameter
            float v;
            if (NT == 9) {
                v = accl[((a.flip_t ? 8 - tap : tap) * NCH + nl) * KP + kl];
            } else {
                const int y = tap / 3, x = tap % 3;
                const float* pp = accl + nl * KP + kl;
                v = (pp[(y * 4 + x) * NCH * KP] + pp[(y * 4 + x + 1) * NCH * KP]) +
                    (pp[((y + 1) * 4 + x) * NCH * KP] + pp[((y + 1) * 4 + x + 1) * NCH * KP]);
            }
            float* d = a.dw + ((size_t)o * a.I + i) * 9 + (a.mode == SGX_PACK_UF ? 8 - tap : tap);
            *d = (a.accumulate & 1) ? *d + c * v : c * v;
        }
        if (do_bias && l15 == 0) {
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* d = a.db + n0 + ns * 16 + q * 4 + r;
                    *d = (a.accumulate & 2) ? *d + bacc[ns][r] : bacc[ns][r];
                }
        }
        return;
    }
    // partial store: out[split][t][n][k]; lane holds rows n = q*4+r, column k = l15
    float* out = a.out + (size_t)blockIdx.y * ((size_t)NT * a.Cn * a.Ck + a.Cn);
    if (do_bias && l15 == 0) {                             // every column of bacc holds the same sums
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)NT * a.Cn * a.Ck + n0 + ns * 16 + q * 4 + r] = bacc[ns][r];
    }
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
struct FinishArgs { const float* ws; float* dw; float* db; int nsplit, O, I, Ip, mode, transposed, flip_t; float scale; int accumulate; };

// EPB consecutive workspace elements per block (consecutive in the partials' fastest dimension, so every read is
// coalesced) x 256/EPB split lanes.  Large weights with few splits use EPB=64; tiny weights with hundreds of splits
// (the 16/32-channel layers at 512^2..1024^2) use EPB=4 so that the launch still has many blocks and 64 split lanes.
template <int EPB>
__global__ __launch_bounds__(256) void wgrad_finish_kernel(FinishArgs f) {
    constexpr int NSL = 256 / EPB;
    __shared__ float red[NSL][EPB][9];
    const int pl = threadIdx.x % EPB, sl = threadIdx.x / EPB;
    const int A = f.transposed ? f.Ip : f.O, Bd = f.transposed ? f.O : f.Ip;
    const int e = blockIdx.x * EPB + pl;                              // workspace element (a, b) = (e / Bd, e % Bd)
    const int taps = f.mode == SGX_PACK_S ? 9 : 16;
    const size_t tstride = (size_t)f.O * f.Ip, total = (size_t)taps * tstride + A;   // + the n side's column sums
    if ((int)blockIdx.x >= (A * Bd + EPB - 1) / EPB) {                // trailing blocks: bias gradient db[o] (unscaled)
        const int o = ((int)blockIdx.x - (A * Bd + EPB - 1) / EPB) * EPB + pl;
        float b = 0.f;
        if (o < f.O)
            for (int s = sl; s < f.nsplit; s += NSL) b += f.ws[(size_t)s * total + (size_t)taps * tstride + o];
        red[sl][pl][0] = b;
        __syncthreads();
        if (sl == 0 && o < f.O) {
            float sum = 0.f;
            for (int l = 0; l < NSL; ++l) sum += red[l][pl][0];
            f.db[o] = (f.accumulate & 2) ? f.db[o] + sum : sum;
        }
        return;
    }
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    if (e < A * Bd) {
#pragma unroll 4
        for (int s = sl; s < f.nsplit; s += NSL) {
            const float* p = f.ws + (size_t)s * total + e;
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
    for (int t = threadIdx.x; t < EPB * 9; t += 256) {                // output (element, tap): fixed summation order
        const int el = t / 9, k = t % 9;
        const int ee = blockIdx.x * EPB + el;
        if (ee >= A * Bd) continue;
        const int a = ee / Bd, bb = ee % Bd;
        const int o = f.transposed ? bb : a, i = f.transposed ? a : bb;
        if (i >= f.I) continue;                                       // padded input channels carry no parameter
        float sum = 0.f;
#pragma unroll 8
        for (int l = 0; l < NSL; ++l) sum += red[l][el][k];
        const float c = f.scale * (f.mode == SGX_PACK_D ? 0.25f : 1.f);
        float* d = f.dw + ((size_t)o * f.I + i) * 9 + (f.mode == SGX_PACK_UF ? 8 - k : k);
        *d = (f.accumulate & 1) ? *d + c * sum : c * sum;
    }
}

// Pixel splits per channel pair.  ~512 blocks for the mid layers (measured 18.20 -> 18.05 ms/step against 1024: half the
// partial traffic); the 16/32-channel layers at 512^2..1024^2 (a handful of
// channel pairs, millions of pixels, partials of a few KB per split) take 2048 blocks / up to 1024 splits -- measured:
// 118 -> 87 us for the 1024^2 16x16 layer -- while for bigger weights more splits cost more in partial traffic than they
// gain (measured).  SGX_WGRAD_BLOCKS / SGX_WGRAD_MAXSPLIT override both for experiments.
static int wgrad_nsplit(int pairs, int ntiles, size_t elems_per_split) {
    static const int env_target = [] { const char* e = getenv("SGX_WGRAD_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    static const int env_cap = [] { const char* e = getenv("SGX_WGRAD_MAXSPLIT"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    const bool tiny = elems_per_split <= (size_t)16 * 32 * 32;
    static const int env_mid = [] { const char* e = getenv("SGX_WGRAD_BLOCKS_MID"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    const int target = env_target ? env_target : (tiny ? 2048 : (env_mid ? env_mid : 512)), cap = env_cap ? env_cap : (tiny ? 1024 : 512);
    int want = (target + pairs - 1) / pairs;
    // big weights at low resolution (512x512 channels at 4^2..32^2): with >= 512 channel pairs the grid is full without
    // pixel splits, and one split means no partials at all (the block writes the parameter gradient itself) -- partial
    // traffic was 4-8x the weight size there.
    if (!env_target && pairs >= 512 && ntiles <= 64) want = 1;
    if (want > cap) want = cap;
    if (want > ntiles) want = ntiles;
    if (want < 1) want = 1;
    return want;
}

template <typename T, int GEO, int TH, int TW, int BP, int NSUB, int KSUB, bool TR>
static int launch_wgrad(WgradArgs& a, void* ws, size_t ws_bytes, int* nsplit_out, hipStream_t st) {
    using F = WFrag<T>;
    constexpr int IS = Geo<GEO>::IS, TK = Geo<GEO>::TK, NT = TK * TK;
    constexpr int PH = (TH - 1) * IS + TK, PW = (TW - 1) * IS + TK, NI = BP / (TH * TW);
    constexpr int OPER = BP * wgrad_pitch<T>(NSUB * 16, TR, 1) + NI * PH * PW * wgrad_pitch<T>(KSUB * 16, TR, IS);
    constexpr int FOLD = NT * NSUB * 16 * (KSUB * 16 + 1) * 4;          // direct epilogue: all taps of the tile, fp32
    constexpr int LDS = OPER > FOLD ? OPER : FOLD;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    a.tiles_y = (a.Hn + TH - 1) / TH; a.tiles_x = (a.Wn + TW - 1) / TW;
    a.ntiles = ((a.B + NI - 1) / NI) * a.tiles_y * a.tiles_x;
    const int pairs = (a.Cn / (NSUB * 16)) * (a.Ck / (KSUB * 16));
    int nsplit = wgrad_nsplit(pairs, a.ntiles, (size_t)NT * a.Cn * a.Ck);
    const size_t total = (size_t)NT * a.Cn * a.Ck + a.Cn;             // per split: all taps + the n side's column sums
    size_t fit = ws_bytes / (total * sizeof(float));
    if ((size_t)nsplit > fit) nsplit = (int)fit;
    SGX_REQUIRE(nsplit >= 1, SGX_EWORKSPACE, "wgrad: workspace too small (%zu bytes, need >= %zu)", ws_bytes, total * sizeof(float));
    a.out = static_cast<float*>(ws);
    a.direct = (nsplit == 1 && a.dw != nullptr) ? 1 : 0;
    auto kern = wgrad_kernel<T, GEO, TH, TW, BP, NSUB, KSUB, TR>;
    sgx_lds_opt_in<wgrad_kernel<T, GEO, TH, TW, BP, NSUB, KSUB, TR>>(LDS);
    hipLaunchKernelGGL(kern, dim3(pairs, nsplit), dim3(256), LDS, st, a);
    SGX_LAUNCH_CHECK("wgrad_kernel");
    *nsplit_out = nsplit;
    return 0;
}

template <typename T, int GEO, int BP, int NSUB, int KSUB, bool TR>
static int wgrad_tile(WgradArgs& a, void* ws, size_t wsb, int* ns, hipStream_t st) {
    if (a.Hn >= 16 && a.Wn >= 16) return launch_wgrad<T, GEO, BP / 16, 16, BP, NSUB, KSUB, TR>(a, ws, wsb, ns, st);
    if (a.Hn >= 8 && a.Wn >= 8) return launch_wgrad<T, GEO, (BP >= 64 ? 8 : BP / 8), 8, BP, NSUB, KSUB, TR>(a, ws, wsb, ns, st);
    return launch_wgrad<T, GEO, (BP >= 16 ? 4 : BP / 4), 4, BP, NSUB, KSUB, TR>(a, ws, wsb, ns, st);
}

// SGX_WGRAD_TR=0/1 selects the plain / transpose-read bf16 operand path (A/B switch for parity tests and profiling)
static bool wgrad_use_tr() {
    static const int v = [] { const char* e = getenv("SGX_WGRAD_TR"); return e ? atoi(e) : 1; }();
    return v != 0;
}

template <typename T, int GEO, int BP, bool TR = false>
static int wgrad_ch(WgradArgs& a, void* ws, size_t wsb, int* ns, hipStream_t st) {
    SGX_REQUIRE(a.Cn % 16 == 0 && a.Ck % 16 == 0, SGX_EUNSUPPORTED, "wgrad: channels must be multiples of 16");
    const bool n32 = a.Cn % 32 == 0, k32 = a.Ck % 32 == 0;
    if (GEO == GDOWN) {                                       // fine patch is 4x larger: keep the k side at 16 channels
        if (n32) return wgrad_tile<T, GEO, BP, 2, 1, TR>(a, ws, wsb, ns, st);
        return wgrad_tile<T, GEO, BP, 1, 1, TR>(a, ws, wsb, ns, st);
    }
    const long npix = (long)a.B * a.Hn * a.Wn;
    const bool lowres_big = npix <= 64L * BP && (long)(a.Cn / 32) * (a.Ck / 32) < 512 && (long)(a.Cn / 32) * (a.Ck / 16) >= 512;
    if (n32 && k32 && !lowres_big) return wgrad_tile<T, GEO, BP, 2, 2, TR>(a, ws, wsb, ns, st);
    if (n32) return wgrad_tile<T, GEO, BP, 2, 1, TR>(a, ws, wsb, ns, st);
    if (k32) return wgrad_tile<T, GEO, BP, 1, 2, TR>(a, ws, wsb, ns, st);
    return wgrad_tile<T, GEO, BP, 1, 1, TR>(a, ws, wsb, ns, st);
}

// Pre-reduction for the many-split case (tiny weights at 512^2..1024^2: 256..1024 splits of a few KB): groups of splits are
// summed element-wise with fully coalesced 16-byte loads into 32 groups, which the finishing kernel then treats as
// its splits.  (The finishing kernel alone reads a 16..32-byte run per split: 70 us for 38 MB of partials.)
static int wgrad_pre_groups() { static const int v = [] { const char* e = getenv("SGX_WGRAD_PRE"); return e ? atoi(e) : 32; }(); return v; }
static int wgrad_pre_min() { static const int v = [] { const char* e = getenv("SGX_WGRAD_PRE_MIN"); return e ? atoi(e) : 256; }(); return v; }
__global__ __launch_bounds__(256) void wgrad_prereduce_kernel(const float* __restrict__ ws, float* __restrict__ out, size_t total4,
                                                              int nsplit) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;      // float4 index inside one split
    if (e >= total4) return;
    const int g = blockIdx.y, per = (nsplit + (int)gridDim.y - 1) / (int)gridDim.y;
    const int s0 = g * per, s1 = (s0 + per < nsplit) ? s0 + per : nsplit;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int sp = s0; sp < s1; ++sp) {
        const float4 v = reinterpret_cast<const float4*>(ws)[(size_t)sp * total4 + e];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(out)[(size_t)g * total4 + e] = acc;
}
static size_t wgrad_pre_bytes(size_t total_floats, int nsplit) {
    const int G = wgrad_pre_groups();                                  // SGX_WGRAD_PRE=0 switches the pass off (A/B)
    return (G > 0 && nsplit >= wgrad_pre_min() && nsplit > G && total_floats % 4 == 0) ? (size_t)G * total_floats * sizeof(float) : 0;
}

static int wgrad_finish(const void* ws, float* dw, float* db, int nsplit, int O, int I, int Ip, int mode, int transposed, int flip_t,
                        float scale, int accumulate, hipStream_t st, size_t ws_bytes = 0) {
    {
        const size_t total = (size_t)(mode == SGX_PACK_S ? 9 : 16) * O * Ip + (transposed ? Ip : O);
        const size_t pre = wgrad_pre_bytes(total, nsplit), used = (size_t)nsplit * total * sizeof(float);
        if (pre && ws_bytes >= used + pre) {
            float* out = reinterpret_cast<float*>(const_cast<char*>(static_cast<const char*>(ws)) + used);
            hipLaunchKernelGGL(wgrad_prereduce_kernel, dim3((unsigned)((total / 4 + 255) / 256), wgrad_pre_groups()), dim3(256), 0, st,
                               static_cast<const float*>(ws), out, total / 4, nsplit);
            SGX_LAUNCH_CHECK("wgrad_prereduce_kernel");
            ws = out;
            nsplit = wgrad_pre_groups();
        }
    }
    FinishArgs f{static_cast<const float*>(ws), dw, db, nsplit, O, I, Ip, mode, transposed, flip_t, scale, accumulate};
    SGX_NOTE(0.0, 4.0 * ((double)nsplit * (mode == SGX_PACK_S ? 9 : 16) * O * Ip + 9.0 * O * I), "finish %dx%d m%d tr%d ns%d", O, I, mode, transposed, nsplit);
    const int ne = O * Ip, nb = db ? O : 0;                            // bias blocks trail the weight blocks
    // widest contiguous run per split lane (coalescing: the partials are read once per split) that still leaves >= 128 blocks
    if (ne / 64 >= 128) hipLaunchKernelGGL(wgrad_finish_kernel<64>, dim3((unsigned)((ne + 63) / 64 + (nb + 63) / 64)), dim3(256), 0, st, f);
    else if (ne / 16 >= 128) hipLaunchKernelGGL(wgrad_finish_kernel<16>, dim3((unsigned)((ne + 15) / 16 + (nb + 15) / 16)), dim3(256), 0, st, f);
    else if (ne / 8 >= 128) hipLaunchKernelGGL(wgrad_finish_kernel<8>, dim3((unsigned)((ne + 7) / 8 + (nb + 7) / 8)), dim3(256), 0, st, f);
    else hipLaunchKernelGGL(wgrad_finish_kernel<4>, dim3((unsigned)((ne + 3) / 4 + (nb + 3) / 4)), dim3(256), 0, st, f);
    SGX_LAUNCH_CHECK("wgrad_finish_kernel");
    return 0;
}

extern "C" size_t sgx_wgrad_ws_bytes(int taps, int B, int H, int W, int Ck, int Cn) {
    size_t total = ((size_t)taps * Ck * Cn + Cn) * sizeof(float);
    int pairs = (Ck / 32 > 0 ? Ck / 32 : 1) * (Cn / 32 > 0 ? Cn / 32 : 1);
    size_t ntiles = (size_t)B * ((H + 3) / 4) * ((W + 3) / 4);
    size_t ns = (size_t)wgrad_nsplit(pairs, ntiles > (1u << 30) ? (1 << 30) : (int)ntiles, (size_t)taps * Ck * Cn);
    size_t need = total * ns + wgrad_pre_bytes(total / sizeof(float), (int)ns);
    int a0, a1, a2, a3;                                             // the second-generation kernel's splits (bf16 only; harmless for fp32)
    const int ns2 = taps == 9 ? sgx_wgrad2_plan(0, B, H, W, Ck, Cn, &a0, &a1, &a2, &a3) : sgx_wgrad2_plan(1, B, H / 2, W / 2, Ck, Cn, &a0, &a1, &a2, &a3);
    const size_t need2 = ns2 ? total * ns2 + wgrad_pre_bytes(total / sizeof(float), ns2) : 0;
    return need > need2 ? need : need2;
}

extern "C" int sgx_wgrad3x3_param(const void* x, const void* dy, float* dW, float* db, void* ws, size_t ws_bytes, int B, int H,
                                  int W, int Cx, int Cdy, int adjoint, float scale, int O, int I, int accumulate, int dtype,
                                  void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int Ip = adjoint ? Cdy : Cx;
    SGX_REQUIRE(!db || !adjoint, SGX_EINVAL, "wgrad3x3_param: bias gradient asked of the adjoint launch");
    SGX_REQUIRE((adjoint ? Cx : Cdy) == O && Ip >= I, SGX_EINVAL, "wgrad3x3_param: channel mismatch (Cx=%d Cdy=%d O=%d I=%d adj=%d)", Cx, Cdy, O, I, adjoint);
    WgradArgs a{x, dy, nullptr, B, H, W, H, W, Cx, Cdy, 0, 0, 0, db ? 1 : 0,
                0, dW, db, O, I, SGX_PACK_S, adjoint, adjoint, accumulate, scale};
    SGX_NOTE(2.0 * 9 * Cx * Cdy * B * H * W, (dtype == SGX_F32 ? 4.0 : 2.0) * B * H * W * (Cx + Cdy), "wgradS B%d %dx%d %dx%d", B, H, W, Cx, Cdy);
    int ns = 0, rc;
    if (dtype == SGX_BF16) {                                       // second-generation kernel (wgrad2.hip) where it applies
        rc = sgx_wgrad2_launch(0, x, dy, static_cast<float*>(ws), ws_bytes, B, H, W, Cx, Cdy, db ? 1 : 0, st, &ns);
        if (rc) return rc;
        if (ns) return wgrad_finish(ws, dW, db, ns, O, I, Ip, SGX_PACK_S, adjoint, adjoint, scale, accumulate, st, ws_bytes);
    }
    if (dtype == SGX_F32) rc = wgrad_ch<float, G3X3, 128>(a, ws, ws_bytes, &ns, st);
    else if (dtype == SGX_BF16) rc = wgrad_use_tr() ? wgrad_ch<bf16_t, G3X3, 128, true>(a, ws, ws_bytes, &ns, st) : wgrad_ch<bf16_t, G3X3, 128>(a, ws, ws_bytes, &ns, st);
    else { SGX_REQUIRE(false, SGX_EINVAL, "wgrad3x3_param: bad dtype"); }
    if (rc) return rc;
    if (a.direct) return 0;
    return wgrad_finish(ws, dW, db, ns, O, I, Ip, SGX_PACK_S, adjoint, adjoint, scale, accumulate, st, ws_bytes);
}

extern "C" int sgx_wgrad4x4s2_param(const void* fine, const void* coarse, float* dW, float* db, void* ws, size_t ws_bytes, int B,
                                    int H, int W, int Cfine, int Ccoarse, int mode, float scale, int O, int I, int accumulate,
                                    int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(H % 2 == 0 && W % 2 == 0, SGX_EINVAL, "wgrad4x4s2_param: odd fine size");
    SGX_REQUIRE(mode == SGX_PACK_D || mode == SGX_PACK_U || mode == SGX_PACK_UF, SGX_EINVAL, "wgrad4x4s2_param: bad mode %d", mode);
    const int transposed = mode != SGX_PACK_D;                 // kernel output is [t][coarse ch][fine ch]
    SGX_REQUIRE(!db || !transposed, SGX_EINVAL, "wgrad4x4s2_param: bias gradient only for mode D (dy is the coarse side)");
    SGX_REQUIRE(transposed ? (Ccoarse == I && Cfine == O) : (Ccoarse == O && Cfine == I), SGX_EINVAL,
                "wgrad4x4s2_param: channel mismatch (fine %d coarse %d O %d I %d mode %d)", Cfine, Ccoarse, O, I, mode);
    WgradArgs a{fine, coarse, nullptr, B, H, W, H / 2, W / 2, Cfine, Ccoarse, 0, 0, 0, db ? 1 : 0,
                0, dW, db, O, I, mode, transposed, 0, accumulate, scale};
    SGX_NOTE(2.0 * 16 * Cfine * Ccoarse * B * (H / 2) * (W / 2), (dtype == SGX_F32 ? 4.0 : 2.0) * B * H * W * (Cfine + Ccoarse / 4.0), "wgradD B%d fine%dx%d %dx%d", B, H, W, Cfine, Ccoarse);
    int ns = 0, rc;
    if (dtype == SGX_BF16) {
        rc = sgx_wgrad2_launch(1, fine, coarse, static_cast<float*>(ws), ws_bytes, B, H / 2, W / 2, Cfine, Ccoarse, db ? 1 : 0, st, &ns);
        if (rc) return rc;
        if (ns) return wgrad_finish(ws, dW, db, ns, O, I, I, mode, transposed, 0, scale, accumulate, st, ws_bytes);
    }
    if (dtype == SGX_F32) rc = wgrad_ch<float, GDOWN, 64>(a, ws, ws_bytes, &ns, st);
    else if (dtype == SGX_BF16) rc = wgrad_use_tr() ? wgrad_ch<bf16_t, GDOWN, 64, true>(a, ws, ws_bytes, &ns, st) : wgrad_ch<bf16_t, GDOWN, 64>(a, ws, ws_bytes, &ns, st);
    else { SGX_REQUIRE(false, SGX_EINVAL, "wgrad4x4s2_param: bad dtype"); }
    if (rc) return rc;
    if (a.direct) return 0;
    return wgrad_finish(ws, dW, db, ns, O, I, I, mode, transposed, 0, scale, accumulate, st, ws_bytes);
}

// =====================================================================================================
// Weight packing: parameter [O][I][3][3] fp32 -> the two MFMA operand packs of a layer, in the activation
// dtype, in one launch: fwd[t][O][Ip] for the layer's own convolution and adj[t'][Ip][O] for its data gradient
// (t' = 8-t for the 3x3 kernel, whose adjoint is spatially flipped; t' = t for the 4x4 stride-2 pair).
//   S : v = scale * w[o][i][ty][tx]                                         (models/CustomLayers.py:170-171)
//   D : v = 0.25*scale * sum_{a,b} w[o][i][ky-a][kx-b]                      (:159-162)
//   U : v = scale * sum_{a,b} w[o][i][ky-a][kx-b];  UF: same on the flipped 3x3 kernel   (:146-150)
// =====================================================================================================
// One block = one 32 (o) x 32 (i) tile of the parameter: the 32 x 288 fp32 source words are read coalesced into LDS, the
// forward pack is written with i fastest and the adjoint pack with o fastest (64-byte bf16 runs either way) -- instead
// of one 2-byte scattered store per element.
#define PACK_T 32
__device__ __forceinline__ float pack_tap(const float (*raw)[PACK_T][PACK_T + 1], int r, int il, int t, int mode, float scale) {
    if (mode == SGX_PACK_S) return scale * raw[t][r][il];
    const int ky = t >> 2, kx = t & 3;
    float v = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int y = ky - a, x = kx - b;
            if (y >= 0 && y < 3 && x >= 0 && x < 3) v += raw[mode == SGX_PACK_UF ? (2 - y) * 3 + (2 - x) : y * 3 + x][r][il];
        }
    return v * scale * (mode == SGX_PACK_D ? 0.25f : 1.f);
}
template <typename T>
__device__ __forceinline__ void pack_weight_tile(const float* __restrict__ w, T* __restrict__ fwd, T* __restrict__ adj, int O, int I,
                                                 int Ip, int mode, float scale, unsigned tile) {
    __shared__ float raw[9][PACK_T][PACK_T + 1];
    const int tiles_i = (Ip + PACK_T - 1) / PACK_T;
    const int o0 = (int)(tile / tiles_i) * PACK_T, i0 = (int)(tile % tiles_i) * PACK_T;
    const int ni = (I - i0 < PACK_T ? (I - i0 > 0 ? I - i0 : 0) : PACK_T);         // real (unpadded) input channels in this tile
    if (ni == PACK_T && (I * 9) % 4 == 0 && ((size_t)i0 * 9) % 4 == 0) {     // 16-byte loads: a tile row is 288 contiguous floats
        for (int idx = threadIdx.x; idx < PACK_T * (PACK_T * 9 / 4); idx += blockDim.x) {
            const int r = idx / (PACK_T * 9 / 4), c4 = (idx % (PACK_T * 9 / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o0 + r < O) v = *reinterpret_cast<const float4*>(w + ((size_t)(o0 + r) * I + i0) * 9 + c4);
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[(c4 + j) % 9][r][(c4 + j) / 9] = e[j];
        }
    } else
    for (int idx = threadIdx.x; idx < PACK_T * PACK_T * 9; idx += blockDim.x) {
        const int r = idx / (PACK_T * 9), c = idx % (PACK_T * 9);
        const int il = c / 9, k = c % 9;
        float v = 0.f;
        if (o0 + r < O && il < ni) v = w[((size_t)(o0 + r) * I + i0) * 9 + c];
        raw[k][r][il] = v;
    }
    __syncthreads();
    const int taps = mode == SGX_PACK_S ? 9 : 16;
    constexpr int VE = 16 / (int)sizeof(T), VPR = PACK_T / VE;                    // 16-byte stores: VE elements per lane
    const bool full = (o0 + PACK_T <= O) && (i0 + PACK_T <= Ip) && (O % VE == 0) && (Ip % VE == 0);
    if (full) {
        for (int e = threadIdx.x; e < taps * PACK_T * VPR; e += blockDim.x) {     // forward pack: i fastest
            const int t = e / (PACK_T * VPR), r = (e / VPR) % PACK_T, v = e % VPR;
            float val[VE];
#pragma unroll
            for (int j = 0; j < VE; ++j) val[j] = pack_tap(raw, r, v * VE + j, t, mode, scale);
            VecTraits<T>::store(fwd + ((size_t)t * O + o0 + r) * Ip + i0 + v * VE, val);
        }
        for (int e = threadIdx.x; e < taps * PACK_T * VPR; e += blockDim.x) {     // adjoint pack: o fastest
            const int t = e / (PACK_T * VPR), il = (e / VPR) % PACK_T, v = e % VPR;
            const int ta = mode == SGX_PACK_S ? 8 - t : t;
            float val[VE];
#pragma unroll
            for (int j = 0; j < VE; ++j) val[j] = pack_tap(raw, v * VE + j, il, t, mode, scale);
            VecTraits<T>::store(adj + ((size_t)ta * Ip + i0 + il) * O + o0 + v * VE, val);
        }
        return;
    }
    for (int e = threadIdx.x; e < taps * PACK_T * PACK_T; e += blockDim.x) {      // edge tiles: element-wise
        const int t = e / (PACK_T * PACK_T), r = (e / PACK_T) % PACK_T, il = e % PACK_T;
        if (o0 + r < O && i0 + il < Ip)
            fwd[((size_t)t * O + o0 + r) * Ip + i0 + il] = from_f<T>(pack_tap(raw, r, il, t, mode, scale));
    }
    for (int e = threadIdx.x; e < taps * PACK_T * PACK_T; e += blockDim.x) {
        const int t = e / (PACK_T * PACK_T), il = (e / PACK_T) % PACK_T, r = e % PACK_T;
        const int ta = mode == SGX_PACK_S ? 8 - t : t;
        if (o0 + r < O && i0 + il < Ip)
            adj[((size_t)ta * Ip + i0 + il) * O + o0 + r] = from_f<T>(pack_tap(raw, r, il, t, mode, scale));
    }
}
static inline unsigned pack_tiles(int O, int Ip) { return (unsigned)(((O + PACK_T - 1) / PACK_T) * ((Ip + PACK_T - 1) / PACK_T)); }
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ fwd, T* __restrict__ adj, int O,
                                                          int I, int Ip, int mode, float scale) {
    pack_weight_tile<T>(w, fwd, adj, O, I, Ip, mode, scale, blockIdx.x);
}
// Every stale weight of a network in ONE launch.  tab: n rows of SGX_PACK_ROW 64-bit words
//   [w, fwd, adj, O, I, Ipad, mode, scale (fp32 bits), first block, blocks];  blocks = sgx_pack_weight_blocks(O, Ipad)
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const long long* __restrict__ tab, int n) {
    int t = 0;
    while (t + 1 < n && (unsigned)tab[(t + 1) * SGX_PACK_ROW + 8] <= blockIdx.x) ++t;      // rows are sorted by first block
    const long long* r = tab + (size_t)t * SGX_PACK_ROW;
    pack_weight_tile<T>(reinterpret_cast<const float*>(r[0]), reinterpret_cast<T*>(r[1]), reinterpret_cast<T*>(r[2]), (int)r[3], (int)r[4],
                        (int)r[5], (int)r[6], __uint_as_float((unsigned)r[7]), blockIdx.x - (unsigned)r[8]);
}
extern "C" int sgx_pack_weight_blocks(int O, int Ipad) { return (int)pack_tiles(O, Ipad); }
extern "C" int sgx_pack_weight_multi(const void* table, int n, int total_blocks, int dtype, void* stream) {
    SGX_REQUIRE(table && n > 0 && total_blocks > 0, SGX_EINVAL, "pack_weight_multi: bad args");
    SGX_NOTE(0.0, 0.0, "pack_multi n%d", n);
    if (dtype == SGX_F32) hipLaunchKernelGGL(pack_weight_multi_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const long long*)table, n);
    else if (dtype == SGX_BF16) hipLaunchKernelGGL(pack_weight_multi_kernel<bf16_t>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const long long*)table, n);
    else { SGX_REQUIRE(false, SGX_EINVAL, "pack_weight_multi: bad dtype"); }
    SGX_LAUNCH_CHECK("pack_weight_multi_kernel");
    return 0;
}

extern "C" int sgx_pack_weight(const float* w, void* fwd, void* adj, int O, int I, int Ipad, int mode, float scale, int dtype,
                               void* stream) {
    SGX_REQUIRE(mode >= SGX_PACK_S && mode <= SGX_PACK_UF && Ipad >= I && O > 0 && I > 0, SGX_EINVAL, "pack_weight: bad args");
    const size_t n = (size_t)(mode == SGX_PACK_S ? 9 : 16) * O * Ipad;
    const size_t g = pack_tiles(O, Ipad);
    SGX_NOTE(0.0, 36.0 * O * I + 2.0 * n * (dtype == SGX_F32 ? 4 : 2), "pack %dx%d m%d", O, I, mode);
    if (dtype == SGX_F32) hipLaunchKernelGGL(pack_weight_kernel<float>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, (float*)fwd, (float*)adj, O, I, Ipad, mode, scale);
    else if (dtype == SGX_BF16) hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)fwd, (bf16_t*)adj, O, I, Ipad, mode, scale);
    else { SGX_REQUIRE(false, SGX_EINVAL, "pack_weight: bad dtype"); }
    SGX_LAUNCH_CHECK("pack_weight_kernel");
    return 0;
}
