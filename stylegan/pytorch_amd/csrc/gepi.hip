// Generator layer epilogue (models/CustomLayers.py:219-248) as four HBM passes instead of ~12:
//   p = x + bias[c] + nw[c]*noise[b,hw];  a = lrelu(p);  xh = (a - mean[b,c]) * rstd[b,c];  y = xh*(s0+1) + s1
// forward : stats pass (per-(b,c) sum / sum of squares; fp32 per lane over <= 64 elements, double across lanes,
//           chunks and the final mean/variance: no float atomics, bit-reproducible) + apply pass.
// backward: reduction pass (sum dy, sum dy*xh per (b,c)) + apply pass that writes dx and the per-channel partials
//           of d(noise weight) and d(bias).
// Layout per block: image b, pixel chunk; thread = (pixel row tr, channel vector tc); channel vectors are 16 bytes.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define GEPI_EPS 1e-5f
#define GEPI_ROWS_PER_THREAD 64
#ifndef GEPI_MIN_BLOCKS
#define GEPI_MIN_BLOCKS 512        // (2048 blocks were tried for the read-only reduction passes: the per-block reduction tail costs more than the extra loads in flight gain)
#endif

struct GepiGeom { int cvt, rows, chunk, nchunk; };
// Rows per thread: 64 for the big layers (few partials), fewer for the small ones so that a launch still has
// GEPI_MIN_BLOCKS blocks (a 4x4..64x64 layer with 64 rows per thread is a handful of blocks walking a long serial loop).
static GepiGeom gepi_geom(int B, int HW, int C, int ve) {
    GepiGeom g;
    int cv = C / ve;
    g.cvt = cv < 256 ? cv : 256;
    g.rows = 256 / g.cvt;
    static const int rpt_max = [] { const char* e = getenv("SGX_GEPI_RPT"); const int v = e ? atoi(e) : 0; return v > 0 ? v : GEPI_ROWS_PER_THREAD; }();
    int rpt = rpt_max;
    while (rpt > 8 && (long)B * ((HW + g.rows * rpt - 1) / (g.rows * rpt)) < GEPI_MIN_BLOCKS) rpt >>= 1;
    g.chunk = g.rows * rpt;
    g.nchunk = (HW + g.chunk - 1) / g.chunk;
    return g;
}

static size_t gepi_ws_head(int B, int HW, int C) {
    GepiGeom g4 = gepi_geom(B, HW, C, 4), g8 = gepi_geom(B, HW, C, 8);
    int nchunk = g4.nchunk > g8.nchunk ? g4.nchunk : g8.nchunk;
    return ((size_t)2 * B * nchunk * C * 2 * sizeof(double) + (size_t)B * C * 2 * sizeof(float) + 255) / 256 * 256;
}
// behind the head: [B][7][C] floats (the short-lived apply passes' coefficient tables), then the per-block partials of gepi_bwd2s
static size_t gepi_ws_tab(int B, int C) { return ((size_t)B * 7 * C * sizeof(float) + 255) / 256 * 256; }
extern "C" size_t sgx_gepi_ws_bytes(int B, int HW, int C) {
    const size_t nblk = ((size_t)HW * (C / 4) + 1023) / 1024;                     // (the fp32 vector width gives the larger count)
    return gepi_ws_head(B, HW, C) + gepi_ws_tab(B, C) + (size_t)B * nblk * C * 2 * sizeof(double) + 256;
}
static float* gepi_ctab(void* ws, int B, int HW, int C) { return reinterpret_cast<float*>(static_cast<char*>(ws) + gepi_ws_head(B, HW, C)); }
static double* gepi_part2s(void* ws, int B, int HW, int C) { return reinterpret_cast<double*>(static_cast<char*>(ws) + gepi_ws_head(B, HW, C) + gepi_ws_tab(B, C)); }

// sum over 16 consecutive lanes (the finalizers give every output 16 lanes that stride over the chunk partials)
__device__ __forceinline__ double sum16(double v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// VE consecutive per-channel coefficients with 16-byte loads (c0 is a multiple of VE, the arrays are 16-byte aligned)
template <int VE>
__device__ __forceinline__ void load_coef(const float* __restrict__ p, float (&k)[VE]) {
#pragma unroll
    for (int j = 0; j < VE; j += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + j);
        k[j] = t.x; k[j + 1] = t.y; k[j + 2] = t.z; k[j + 3] = t.w;
    }
}

// raw 16-byte vector <-> VE floats (bf16: the pack is v_cvt_pk_bf16_f32, round to nearest even, as every store of the library)
template <typename T> __device__ __forceinline__ void unpack16(const uint4& q, float (&v)[VecTraits<T>::VE]);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& q, float (&v)[4]) {
    v[0] = __uint_as_float(q.x); v[1] = __uint_as_float(q.y); v[2] = __uint_as_float(q.z); v[3] = __uint_as_float(q.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& q, float (&v)[8]) {
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float (&v)[VecTraits<T>::VE]);
template <> __device__ __forceinline__ uint4 pack16<float>(const float (&v)[4]) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float (&v)[8]) {
    return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

// Block-wide, fixed-order sum of the per-chunk partial pairs of image b: part_b[k][C][2], k < npart -> for every channel the two
// totals, handed to ``fin(c, total0, total1)`` by ONE thread per channel.  256 threads; red: 512 doubles of LDS scratch.  This is
// the work of the former gepi_fin_stats / gepi_fin_bwd1 launches, done by every block of the pass that consumes the result
// (a few KB of L2 reads per block) instead of a 7 us launch of its own between the two passes.
template <typename Fin>
__device__ __forceinline__ void gepi_block_totals(const double* __restrict__ part_b, int npart, int C, double* red, Fin fin) {
    const int Cc = C < 256 ? C : 256, L = 256 / Cc;               // L lanes per channel stride over the chunks
    const int cl = threadIdx.x % Cc, ln = threadIdx.x / Cc;
    for (int cb = 0; cb < C; cb += Cc) {                          // uniform trip count
        const int c = cb + cl;
        double t0 = 0.0, t1 = 0.0;
        if (ln < L) {
            // eight independent 16-byte loads in flight per lane (a dependent chain of L2 misses here sits on the critical path of
            // every block of the pass: measured +10 us per launch with a rolled loop)
            for (int k0 = ln; k0 < npart; k0 += 8 * L) {
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * L;
                    v[u] = k < npart ? *reinterpret_cast<const double2*>(part_b + ((size_t)k * C + c) * 2) : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { t0 += v[u].x; t1 += v[u].y; }
            }
        }
        red[threadIdx.x * 2] = t0; red[threadIdx.x * 2 + 1] = t1;
        __syncthreads();
        if (ln == 0) {
            double a0 = 0.0, a1 = 0.0;
            for (int l = 0; l < L; ++l) { a0 += red[(l * Cc + cl) * 2]; a1 += red[(l * Cc + cl) * 2 + 1]; }
            fin(c, a0, a1);
        }
        __syncthreads();
    }
}

// mode 0: (sum a, sum a^2)                       [forward statistics]
// mode 1: (sum dy, sum dy*xh)                    [backward reduction]
// mode 2: writes dx, (sum dp*noise, sum dp)      [backward apply]
template <typename T, int MODE, bool NT = false>
__global__ __launch_bounds__(256) void gepi_pass(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                 const float* __restrict__ bias, const float* __restrict__ noise,
                                                 const float* __restrict__ nw, const float* __restrict__ style,
                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                 const float* __restrict__ coef, double* __restrict__ part, int HW, int C,
                                                 int cvt, int rows, int chunk, int act,
                                                 const double* __restrict__ fpart, int fnpart, int fnorm, float* __restrict__ dstyle) {
    constexpr int VE = VecTraits<T>::VE;
    extern __shared__ double sh[];                                 // [256][2*VE]  (+ MODE 2 with fpart: float [2][C] after it)
    const int b = blockIdx.y, ch = blockIdx.x;
    float* const scoef = reinterpret_cast<float*>(sh + 256 * 2 * VE);   // [C] k1, [C] k2 of image b
    if (MODE == 2 && fpart) {
        // what gepi_fin_bwd1 computed: dstyle (written by the image's first block) and the two statistics-gradient coefficients
        gepi_block_totals(fpart + (size_t)b * fnpart * C * 2, fnpart, C, sh, [&](int c, double s1, double s0) {
            if (ch == 0 && dstyle) {
                dstyle[(size_t)b * 2 * C + c] = (float)s0;         // d/d style[:,0] = sum dy*xh
                dstyle[(size_t)b * 2 * C + C + c] = (float)s1;     // d/d style[:,1] = sum dy
            }
            const double sc = (double)style[(size_t)b * 2 * C + c] + 1.0;
            scoef[c] = fnorm ? (float)(sc * s1 / HW) : 0.f;
            scoef[C + c] = fnorm ? (float)(sc * s0 / HW) : 0.f;
        });
    }
    const int tc = threadIdx.x % cvt, tr = threadIdx.x / cvt;
    const int cv = C / VE;
    const int p0 = ch * chunk, p1 = (p0 + chunk < HW) ? p0 + chunk : HW;
    for (int vb = 0; vb < cv; vb += cvt) {                         // uniform trip count (cv is a multiple of cvt)
        const int v = vb + tc, c0 = v * VE;
        float s0[VE], s1[VE], kb[VE], kw[VE], km[VE], kr[VE], ks[VE], k1[VE], k2[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) { s0[j] = 0.f; s1[j] = 0.f; kb[j] = 0.f; }
        if (bias) load_coef<VE>(bias + c0, kb);
        load_coef<VE>(nw + c0, kw);
        if (MODE >= 1) {
            load_coef<VE>(mean + (size_t)b * C + c0, km);
            load_coef<VE>(rstd + (size_t)b * C + c0, kr);
            load_coef<VE>(style + (size_t)b * 2 * C + c0, ks);
#pragma unroll
            for (int j = 0; j < VE; ++j) ks[j] += 1.f;
        }
        if (MODE == 2) {
            if (fpart) {
#pragma unroll
                for (int j = 0; j < VE; ++j) { k1[j] = scoef[c0 + j]; k2[j] = scoef[C + c0 + j]; }
            } else {
                float kk[2 * VE];
                load_coef<2 * VE>(coef + ((size_t)b * C + c0) * 2, kk);
#pragma unroll
                for (int j = 0; j < VE; ++j) { k1[j] = kk[2 * j]; k2[j] = kk[2 * j + 1]; }
            }
        }
        if (tr < rows) {
            // loads in flight per lane: 8 rows for the read-only statistics pass (2 blocks per CU: it is latency-bound, 2.6 ->
            // 3.2 TB/s); 4 for the two-tensor passes (8 costs them the second wave per SIMD: 284 VGPRs, measured 2x slower)
            // (round 6, tools/gepi_probe.py with 1 / 2 / 4 / 8 in flight for both read-only passes: no difference beyond noise)
            constexpr int UNR = MODE == 0 ? 8 : 4;
#pragma unroll UNR
            for (int p = p0 + tr; p < p1; p += rows) {
                const size_t off = (((size_t)b * HW + p) * cv + v) * VE;
                const float nz = noise[(size_t)b * HW + p];
                float xv[VE];
                if constexpr (NT) { const uint4 q = ld16(x + off, true); unpack16<T>(q, xv); }
                else VecTraits<T>::load(x + off, xv);
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < VE; ++j) {
                        const float a = act_apply(xv[j] + kb[j] + kw[j] * nz, act);
                        s0[j] += a; s1[j] += a * a;
                    }
                } else {
                    float gv[VE];
                    if constexpr (NT) { const uint4 q = ld16(dy + off, true); unpack16<T>(q, gv); }
                    else VecTraits<T>::load(dy + off, gv);
                    if (MODE == 1) {
#pragma unroll
                        for (int j = 0; j < VE; ++j) {
                            const float xh = (act_apply(xv[j] + kb[j] + kw[j] * nz, act) - km[j]) * kr[j];
                            s0[j] += gv[j]; s1[j] += gv[j] * xh;
                        }
                    } else {
                        float ov[VE];
#pragma unroll
                        for (int j = 0; j < VE; ++j) {
                            const float pp = xv[j] + kb[j] + kw[j] * nz;
                            const float xh = (act_apply(pp, act) - km[j]) * kr[j];
                            const float da = kr[j] * (gv[j] * ks[j] - k1[j] - xh * k2[j]);
                            const float dp = act ? da * lrelu_slope(pp) : da;
                            ov[j] = dp;
                            s0[j] += dp * nz; s1[j] += dp;
                        }
                        if constexpr (NT) st16(dx + off, pack16<T>(ov), true);
                        else VecTraits<T>::store(dx + off, ov);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < VE; ++j) { sh[threadIdx.x * 2 * VE + j] = (double)s0[j]; sh[threadIdx.x * 2 * VE + VE + j] = (double)s1[j]; }
        __syncthreads();
        // Sum over the block's pixel rows in a fixed order.  NO = outputs of the block (channel vectors x 2*VE sums).  Few
        // channels (C = 16: 32 outputs, 128 rows): every thread sums one output over a slice of 2*VE rows, then the first NO
        // threads add the 256/NO slices -- instead of `cvt` threads walking all rows while 250 idle.  Many channels: one
        // thread per channel vector walks the (few) rows.
        const int NO = cvt * 2 * VE;
        if (NO < 256 && rows > 1) {
            const int o = threadIdx.x % NO, slice = threadIdx.x / NO, otc = o / (2 * VE), oj = o % (2 * VE);
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < 2 * VE; ++r) acc += sh[((slice * 2 * VE + r) * cvt + otc) * 2 * VE + oj];
            __syncthreads();                                         // every read of the per-thread sums is done
            sh[threadIdx.x] = acc;                                   // [slice][output]
            __syncthreads();
            if ((int)threadIdx.x < NO) {
                double a = 0.0;
                for (int sl = 0; sl < 256 / NO; ++sl) a += sh[sl * NO + threadIdx.x];
                part[(((size_t)b * gridDim.x + ch) * C + (size_t)(vb + otc) * VE + (oj % VE)) * 2 + (oj / VE)] = a;
            }
        } else if (tr == 0) {
            double* o = part + (((size_t)b * gridDim.x + ch) * C + c0) * 2;
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                double a0 = 0.0, a1 = 0.0;
                for (int r = 0; r < rows; ++r) {
                    a0 += sh[(r * cvt + tc) * 2 * VE + j];
                    a1 += sh[(r * cvt + tc) * 2 * VE + VE + j];
                }
                o[j * 2] = a0; o[j * 2 + 1] = a1;
            }
        }
        __syncthreads();
    }
}

// forward finalize: mean / rstd per (b,c); 16 lanes per output
// ``tab`` (optional): the apply pass's per-(image, channel) coefficients packed as [B][6][C] floats -- bias, noise weight, mean, rstd, style
// scale + 1, style shift -- so that a short-lived block of gepi_apply1 fetches its whole table with ONE coalesced load per thread
__global__ void gepi_fin_stats(const double* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd, int B, int C,
                               int nchunk, int HW, int norm, float* __restrict__ tab = nullptr, const float* __restrict__ bias = nullptr,
                               const float* __restrict__ nw = nullptr, const float* __restrict__ style = nullptr) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l = threadIdx.x & 15;
    const bool ok = i < B * C;
    const int b = ok ? i / C : 0, c = ok ? i % C : 0;
    auto put = [&](float m, float r) {
        mean[i] = m; rstd[i] = r;
        if (tab) {
            float* t = tab + (size_t)b * 6 * C + c;
            t[0] = bias ? bias[c] : 0.f; t[C] = nw[c]; t[2 * C] = m; t[3 * C] = r;
            t[4 * C] = style[(size_t)b * 2 * C + c] + 1.f; t[5 * C] = style[(size_t)b * 2 * C + C + c];
        }
    };
    if (!norm) {                                                   // no instance norm: xh = a
        if (ok && !l) put(0.f, 1.f);
        return;
    }
    double s = 0.0, ss = 0.0;
    if (ok)
        for (int k = l; k < nchunk; k += 16) {
            const double* p = part + (((size_t)b * nchunk + k) * C + c) * 2;
            s += p[0]; ss += p[1];
        }
    s = sum16(s); ss = sum16(ss);
    if (!ok || l) return;
    const double m = s / HW;
    double var = ss / HW - m * m;
    if (var < 0.0) var = 0.0;
    put((float)m, (float)(1.0 / sqrt(var + (double)GEPI_EPS)));
}

// backward finalize 1: dstyle and the two per-(b,c) coefficients of the apply pass
// ``tab`` (optional): the short-lived backward apply pass's per-(image, channel) coefficients packed as [B][7][C] floats -- bias, noise weight, mean,
// rstd, style scale + 1, and the two statistics-gradient coefficients -- one coalesced load per thread of a block (gepi_bwd2s)
__global__ void gepi_fin_bwd1(const double* __restrict__ part, const float* __restrict__ style, float* __restrict__ dstyle,
                              float* __restrict__ coef, int B, int C, int nchunk, int HW, int norm, float* __restrict__ tab = nullptr,
                              const float* __restrict__ bias = nullptr, const float* __restrict__ nw = nullptr, const float* __restrict__ mean = nullptr,
                              const float* __restrict__ rstd = nullptr) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l = threadIdx.x & 15;
    const bool ok = i < B * C;
    const int b = ok ? i / C : 0, c = ok ? i % C : 0;
    double s1 = 0.0, s0 = 0.0;
    if (ok)
        for (int k = l; k < nchunk; k += 16) {
            const double* p = part + (((size_t)b * nchunk + k) * C + c) * 2;
            s1 += p[0]; s0 += p[1];
        }
    s1 = sum16(s1); s0 = sum16(s0);
    if (!ok || l) return;
    dstyle[(size_t)b * 2 * C + c] = (float)s0;              // d/d style[:,0] = sum dy*xh
    dstyle[(size_t)b * 2 * C + C + c] = (float)s1;          // d/d style[:,1] = sum dy
    const double sc = (double)style[(size_t)b * 2 * C + c] + 1.0;
    const float k1 = norm ? (float)(sc * s1 / HW) : 0.f, k2 = norm ? (float)(sc * s0 / HW) : 0.f;   // the statistics' own gradient terms (instance norm only)
    coef[(size_t)i * 2] = k1;
    coef[(size_t)i * 2 + 1] = k2;
    if (tab) {
        float* t = tab + (size_t)b * 7 * C + c;
        t[0] = bias ? bias[c] : 0.f; t[C] = nw[c]; t[2 * C] = mean[i]; t[3 * C] = rstd[i];
        t[4 * C] = style[(size_t)b * 2 * C + c] + 1.f; t[5 * C] = k1; t[6 * C] = k2;
    }
}

// backward finalize 2: d noise-weight and d bias per channel (sum over images and chunks); one wave per channel (batch 32 at
// 1024^2 has 8192 partials per channel: with 16 lanes per channel and one block for 16 channels this was a 32 us serial tail)
__global__ void gepi_fin_bwd2(const double* __restrict__ part, float* __restrict__ dnw, float* __restrict__ dbias, int B, int C,
                              int nchunk) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
    const bool ok = c < C;
    double a = 0.0, d = 0.0;
    if (ok)
        for (int k = l; k < B * nchunk; k += 64) {
            const double2 p = *reinterpret_cast<const double2*>(part + ((size_t)k * C + c) * 2);
            a += p.x; d += p.y;
        }
    a = wave_sum_d(a); d = wave_sum_d(d);
    if (!ok || l) return;
    dnw[c] = (float)a;
    if (dbias) dbias[c] = (float)d;
}

// forward apply pass, same block geometry as gepi_pass: image b and channel vector are fixed per thread, so the six
// per-(b,c) coefficients live in registers and the pixel loop is pure streaming (4 rows in flight per lane).
template <typename T>
__global__ __launch_bounds__(256) void gepi_apply(const T* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ noise,
                                                  const float* __restrict__ nw, const float* __restrict__ style,
                                                  const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ y,
                                                  int HW, int C, int cvt, int rows, int chunk, int act,
                                                  const double* __restrict__ spart, int snpart, int snorm,
                                                  float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    constexpr int VE = VecTraits<T>::VE;
    extern __shared__ double sha[];                                // spart: 512 doubles of scratch, then float [2][C] mean / rstd
    const int b = blockIdx.y, ch = blockIdx.x;
    const int tc = threadIdx.x % cvt, tr = threadIdx.x / cvt;
    const int cv = C / VE;
    const int p0 = ch * chunk, p1 = (p0 + chunk < HW) ? p0 + chunk : HW;
    float* const sstat = reinterpret_cast<float*>(sha + 512);
    if (spart) {
        // what gepi_fin_stats computed (mean / rstd per channel of image b), by every block for itself; the image's first block
        // also writes them out (the backward reads them)
        gepi_block_totals(spart + (size_t)b * snpart * C * 2, snpart, C, sha, [&](int c, double sum, double sq) {
            float m = 0.f, r = 1.f;
            if (snorm) {
                const double mm = sum / HW;
                double var = sq / HW - mm * mm;
                if (var < 0.0) var = 0.0;
                m = (float)mm; r = (float)(1.0 / sqrt(var + (double)GEPI_EPS));
            }
            sstat[c] = m; sstat[C + c] = r;
            if (ch == 0) { mean_out[(size_t)b * C + c] = m; rstd_out[(size_t)b * C + c] = r; }
        });
    }
    if (tr >= rows) return;
    for (int vb = 0; vb < cv; vb += cvt) {
        const int v = vb + tc, c0 = v * VE;
        float kb[VE], kw[VE], km[VE], kr[VE], ks[VE], k1[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) kb[j] = 0.f;
        if (bias) load_coef<VE>(bias + c0, kb);
        load_coef<VE>(nw + c0, kw);
        if (spart) {
#pragma unroll
            for (int j = 0; j < VE; ++j) { km[j] = sstat[c0 + j]; kr[j] = sstat[C + c0 + j]; }
        } else {
            load_coef<VE>(mean + (size_t)b * C + c0, km);
            load_coef<VE>(rstd + (size_t)b * C + c0, kr);
        }
        load_coef<VE>(style + (size_t)b * 2 * C + c0, ks);
        load_coef<VE>(style + (size_t)b * 2 * C + C + c0, k1);
#pragma unroll
        for (int j = 0; j < VE; ++j) ks[j] += 1.f;
#pragma unroll 4
        for (int p = p0 + tr; p < p1; p += rows) {
            const size_t off = (((size_t)b * HW + p) * cv + v) * VE;
            const float nz = noise[(size_t)b * HW + p];
            float xv[VE];
            VecTraits<T>::load(x + off, xv);
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                const float a = act_apply(xv[j] + kb[j] + kw[j] * nz, act);
                const float xh = (a - km[j]) * kr[j];
                xv[j] = xh * ks[j] + k1[j];
            }
            VecTraits<T>::store(y + off, xv);
        }
    }
}


// ---- the apply pass as SHORT-LIVED blocks (round 6).  tools/stream_probe.hip on the MI355X: a read + write stream of 537 MB runs at 5.1 TB/s
// in this file's long-loop block shape (64 pixel rows per thread), at 5.0 as a capped grid-stride loop -- and at 6.1-6.2 TB/s when every
// thread moves ONE 16-byte vector and exits (aten's elementwise kernels sit at 6.3 the same way); more work per thread only loses (x2 5.9, x4
// 5.7, x8 5.4).  What kept the loop here was the 6 x VE per-(image, channel) coefficients a thread needs: loaded per thread from global for one
// vector they cost 2.7 TB/s.  So: the block's coefficient table (6 x C floats) goes through LDS once per block, requested AFTER the thread's
// own data and noise loads so that the three latencies overlap, and every thread then does the arithmetic of gepi_apply on one vector --
// same operations in the same order: bit-identical output (probe: 6.05 TB/s for this shape).  mean / rstd come from gepi_fin_stats.
template <typename T>
__global__ __launch_bounds__(256) void gepi_apply1(const T* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ ctab,
                                                   T* __restrict__ y, int HW, int C, int act) {
    constexpr int VE = VecTraits<T>::VE;
    extern __shared__ float tab[];                                 // [6][C]: bias, noise weight, mean, rstd, style scale + 1, style shift (gepi_fin_stats)
    const int b = blockIdx.y, cv = C / VE;
    const unsigned nvi = (unsigned)HW * (unsigned)cv;              // vectors per image (< 2^31: 32-bit index arithmetic, one 32-bit divide)
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const bool live = i < nvi;
    const unsigned p = live ? i / (unsigned)cv : 0u;
    const int c0 = (int)(i - p * (unsigned)cv) * VE;
    const size_t off = ((size_t)b * nvi + (live ? i : 0u)) * VE;
    uint4 raw = make_uint4(0u, 0u, 0u, 0u);
    float nz = 0.f;
    if (live) {
        raw = ld16(x + off, true);                                 // (these launches are tensors of >= 192 MB: nontemporal both ways)
        nz = noise[(size_t)b * HW + p];
    }
    const float4* src = reinterpret_cast<const float4*>(ctab + (size_t)b * 6 * C);
    for (int j = threadIdx.x; j < 6 * C / 4; j += 256) reinterpret_cast<float4*>(tab)[j] = src[j];
    __syncthreads();
    if (!live) return;
    float kb[VE], kw[VE], km[VE], kr[VE], ks[VE], k1[VE], xv[VE];
    load_coef<VE>(tab + c0, kb); load_coef<VE>(tab + C + c0, kw); load_coef<VE>(tab + 2 * C + c0, km);
    load_coef<VE>(tab + 3 * C + c0, kr); load_coef<VE>(tab + 4 * C + c0, ks); load_coef<VE>(tab + 5 * C + c0, k1);
    unpack16<T>(raw, xv);
#pragma unroll
    for (int j = 0; j < VE; ++j) {
        const float a = act_apply(xv[j] + kb[j] + kw[j] * nz, act);
        const float xh = (a - km[j]) * kr[j];
        xv[j] = xh * ks[j] + k1[j];
    }
    st16(y + off, pack16<T>(xv), true);
}
// ---- the backward apply pass as short-lived blocks (round 6; as gepi_apply1): two inputs, one output -- 4.0 TB/s through a capped grid-stride
// loop, 6.0 as blocks that move a vector or four per thread and exit (tools/stream_probe.hip T0).  A thread owns U = 4 vectors of ONE channel vector
// (pixels 256 / cv apart), so the per-channel sums of d noise-weight / d bias are 2 x VE fp32 partials per thread; lanes of a wave with the same
// channel vector are summed by xor shuffles, the four waves through LDS (fp64), one partial per (block, channel) for gepi_fin_bwd2 -- a fixed
// order, no atomics.  dx: the arithmetic of gepi_pass<T, 2> on the coefficients gepi_fin_bwd1 computed: bit-identical.
template <typename T>
__global__ __launch_bounds__(256) void gepi_bwd2s(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, const float* __restrict__ noise,
                                                  const float* __restrict__ ctab, double* __restrict__ part, int HW, int C, int act) {
    constexpr int VE = VecTraits<T>::VE, U = 4;
    extern __shared__ float tab[];                                 // [7][C] coefficients, then [4 waves][cv][2 * VE] doubles
    const int b = blockIdx.y, cv = C / VE, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lcv = 31 - __builtin_clz((unsigned)cv);              // cv is a power of two (gepi_bwd2s_on): shifts, no division (a 64-bit divide per
                                                                   // vector made the first version of this kernel 45 % SLOWER than the loop it replaces)
    const unsigned nvi = (unsigned)HW * (unsigned)cv;              // vectors per image (< 2^31: gepi_check)
    const unsigned i0 = blockIdx.x * (256u * U) + threadIdx.x;
    const int c0 = (int)(i0 & (unsigned)(cv - 1)) * VE;            // (256 % cv == 0: the same channel vector for every u)
    const T* xb = x + (size_t)b * nvi * VE; const T* gb = dy + (size_t)b * nvi * VE; T* db = dx + (size_t)b * nvi * VE;
    const float* nzb = noise + (size_t)b * HW;
    uint4 rx[U], rg[U];
    float nz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned i = i0 + u * 256u;
        rx[u] = make_uint4(0u, 0u, 0u, 0u); rg[u] = rx[u]; nz[u] = 0.f;
        if (i < nvi) {
            rx[u] = ld16(xb + (size_t)i * VE, true);
            rg[u] = ld16(gb + (size_t)i * VE, true);
            nz[u] = nzb[i >> lcv];
        }
    }
    const float4* src = reinterpret_cast<const float4*>(ctab + (size_t)b * 7 * C);
    for (int j = threadIdx.x; j < 7 * C / 4; j += 256) reinterpret_cast<float4*>(tab)[j] = src[j];
    __syncthreads();
    float kb[VE], kw[VE], km[VE], kr[VE], ks[VE], k1[VE], k2[VE], s0[VE], s1[VE];
    load_coef<VE>(tab + c0, kb); load_coef<VE>(tab + C + c0, kw); load_coef<VE>(tab + 2 * C + c0, km); load_coef<VE>(tab + 3 * C + c0, kr);
    load_coef<VE>(tab + 4 * C + c0, ks); load_coef<VE>(tab + 5 * C + c0, k1); load_coef<VE>(tab + 6 * C + c0, k2);
#pragma unroll
    for (int j = 0; j < VE; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned i = i0 + u * 256u;
        if (i < nvi) {
            float xv[VE], gv[VE], ov[VE];
            unpack16<T>(rx[u], xv); unpack16<T>(rg[u], gv);
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                const float pp = xv[j] + kb[j] + kw[j] * nz[u];
                const float xh = (act_apply(pp, act) - km[j]) * kr[j];
                const float da = kr[j] * (gv[j] * ks[j] - k1[j] - xh * k2[j]);
                const float dp = act ? da * lrelu_slope(pp) : da;
                ov[j] = dp;
                s0[j] += dp * nz[u]; s1[j] += dp;
            }
            st16(db + (size_t)i * VE, pack16<T>(ov), true);
        }
    }
    // lanes of a 16-lane ROW with the same channel vector (lane % cv; cv a power of two <= 16): DPP row rotations -- one v_add per step and value
    // (the first version summed over the whole wave with 80 ds_bpermute shuffles per thread and was 75 % slower than the loop it replaces)
    auto ror_add = [](float v, auto N) {
        constexpr int n = decltype(N)::value;
        return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + n, 0xf, 0xf, false));
    };
#pragma unroll
    for (int j = 0; j < VE; ++j) {
        if (cv <= 8) { s0[j] = ror_add(s0[j], std::integral_constant<int, 8>{}); s1[j] = ror_add(s1[j], std::integral_constant<int, 8>{}); }
        if (cv <= 4) { s0[j] = ror_add(s0[j], std::integral_constant<int, 4>{}); s1[j] = ror_add(s1[j], std::integral_constant<int, 4>{}); }
        if (cv <= 2) { s0[j] = ror_add(s0[j], std::integral_constant<int, 2>{}); s1[j] = ror_add(s1[j], std::integral_constant<int, 2>{}); }
        if (cv <= 1) { s0[j] = ror_add(s0[j], std::integral_constant<int, 1>{}); s1[j] = ror_add(s1[j], std::integral_constant<int, 1>{}); }
    }
    // the 16 rows of the block (4 waves x 4 rows) through LDS in fp64, summed in a fixed order by one thread per (channel vector, value)
    double* wtot = reinterpret_cast<double*>(tab + 7 * C + (7 * C & 1));       // [16 rows][cv][2 * VE] (8-byte aligned: 7 C is even for C % 8 == 0; kept general)
    const int row = wave * 4 + (lane >> 4), l15 = lane & 15;
    if (l15 < cv) {
#pragma unroll
        for (int j = 0; j < VE; ++j) { wtot[(row * cv + l15) * 2 * VE + j] = (double)s0[j]; wtot[(row * cv + l15) * 2 * VE + VE + j] = (double)s1[j]; }
    }
    __syncthreads();
    if ((int)threadIdx.x < cv * 2 * VE) {
        const int v = threadIdx.x / (2 * VE), k = threadIdx.x % (2 * VE);
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += wtot[(r * cv + v) * 2 * VE + k];
        part[(((size_t)b * gridDim.x + blockIdx.x) * C + (size_t)v * VE + (k % VE)) * 2 + (k / VE)] = a;
    }
}
// gepi_bwd2s leaves one partial row of 2 C doubles per BLOCK (65 536 rows at 1024^2 x 16, batch 32): gepi_fin_bwd2 walks a channel's rows with one
// wave -- 16 waves on the whole chip, 500 us for 17 MB.  This pre-reduction sums contiguous ranges of rows with every lane loading its own
// value of consecutive rows (coalesced): out[j][w] = sum of rows [j K / NB, (j + 1) K / NB), NB <= 256 rows left for gepi_fin_bwd2.
__global__ __launch_bounds__(256) void gepi_rows_prereduce(const double* __restrict__ part, double* __restrict__ out, int K, int Wd) {
    extern __shared__ double shr[];                                // [256]
    const int NB = gridDim.x, j = blockIdx.x;
    const int k0 = (int)((long)j * K / NB), k1 = (int)((long)(j + 1) * K / NB);
    const int lanes = 256 / Wd * Wd;                               // threads in use: whole rows per sweep (Wd <= 256)
    const int w = threadIdx.x % Wd, r = threadIdx.x / Wd, rows = 256 / Wd;
    double a = 0.0;
    if ((int)threadIdx.x < lanes)
        for (int k = k0 + r; k < k1; k += rows) a += part[(size_t)k * Wd + w];
    shr[threadIdx.x] = (int)threadIdx.x < lanes ? a : 0.0;
    __syncthreads();
    if ((int)threadIdx.x < Wd) {
        double t = 0.0;
        for (int q = 0; q < rows; ++q) t += shr[q * Wd + threadIdx.x];
        out[(size_t)j * Wd + threadIdx.x] = t;
    }
}
// Measured alone (tools/gepi_probe.py, both backward passes + finalisers, batch 32): 1024^2 x 16 1043..1083 -> 1002 us, 512^2 x 32 531 -> 526,
// 256^2 x 64 284 -> 319 (worse): the noise load, the coefficient table, the in-block sums and the extra pre-reduction launch eat most of what the
// two-input stream gains in the probe (267 vs 348 us for 1.6 GB).  So: tensors of >= 800 MB only (the 1024^2 layer from batch 24 up).
static bool gepi_bwd2s_on(int C, int VE, double tensor_bytes) {
    static const int on = [] { const char* e = getenv("SGX_GEPI_BWD2S"); return e ? atoi(e) : 1; }();       // A/B switch; 2 = whatever the size
    const int cv = C / VE;
    return on != 0 && C <= 128 && cv >= 1 && cv <= 16 && (cv & (cv - 1)) == 0 && (on == 2 || tensor_bytes >= 800e6);
}
// partials of the short-lived backward apply pass: blocks per image
static int gepi_bwd2s_blocks(int HW, int C, int VE) { return (int)(((size_t)HW * (C / VE) + 1023) / 1024); }

// the launch of the apply pass: short-lived blocks (SGX_GEPI_APPLY1, default 1) or the long-loop kernel with the statistics finalize folded in
// Measured alone (tools/gepi_probe.py, statistics + apply, batch 32): 1024^2 x 16: 664 -> 599 us, 512^2 x 32: 345 -> 310, 256^2 x 64: 175 -> 170,
// 128^2 x 128: no change; batch 4 (tensors of <= 134 MB): +2..3 us for the separate finalize launch and nothing back.  So: tensors of >= 192 MB.
static bool gepi_apply1_on(int C, double tensor_bytes) {
    static const int on = [] { const char* e = getenv("SGX_GEPI_APPLY1"); return e ? atoi(e) : 1; }();      // 2 = whatever the size
    // (a block's table is 24 C bytes next to 4 KB of data: the 256 / 512-channel layers keep the loop)
    return on != 0 && C <= 128 && (on == 2 || tensor_bytes >= 192e6);
}

// ---------------------------------------------------------------------------------------------------------------
// Small layers (round 6): the whole epilogue of one (image, 16-channel group) in ONE block, one launch per direction.
// At 4x4 .. 64x64 a layer's tensor is a few hundred KB to 8 MB (L2 resident) and the two-pass structure above costs two
// (forward) / three (backward) launches of 8-20 us each that are bound by launch latency and by the round trip of the
// per-chunk partials through memory -- 40 of the 126 epilogue launches of a batch-4 step.  Here a block owns ALL pixels of
// its 16 channels of image b: the statistics are block-local (fp32 per lane over <= IT values, fp64 across lanes and waves in
// a fixed order: deterministic), and the block goes straight on to the apply pass from the SAME registers: every lane keeps
// its <= IT pixels (IT x 16 bytes) from the first pass, nothing is read twice.  thread = (pixel row r, 16-byte channel vector
// v); NT threads: 256 up to 16x16, 1024 above (a 64x64 layer is 8 pixels per lane; the first version walked 32 pixels per lane
// twice through L2 with 256 threads and lost to the two-pass kernels' 512 blocks: 48 -> 96 us at batch 32, 64x64).
#define GS_CG 16                                                   // channels per block
// Fixed-order block sum of the per-thread partial vectors (q0, q1)[VE]: lanes of a wave with the same channel vector by xor
// shuffles (fp64), the waves' results through LDS -> tot[v * 2 * VE + j] for j < VE (q0) and VE + j (q1).  NV * 2 * VE = 32 outputs.
template <int VE, int NT>
__device__ __forceinline__ void gs_block_sum(const float (&q0)[VE], const float (&q1)[VE], double* sh, double* tot) {
    constexpr int NV = GS_CG / VE, W2 = 2 * VE, NO = NV * W2, NWV = NT / 64;
    static_assert(NO == 32, "32 outputs per block");
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
#pragma unroll
    for (int j = 0; j < W2; ++j) {                                 // one value at a time: two live registers, not 4 * VE
        double d = (double)(j < VE ? q0[j < VE ? j : 0] : q1[j < VE ? 0 : j - VE]);
#pragma unroll
        for (int o = 32; o >= NV; o >>= 1) d += __shfl_xor(d, o, 64);
        if (lane < NV) sh[(wv * NV + lane) * W2 + j] = d;
    }
    __syncthreads();
    if (t < NO) {
        const int v = t / W2, j = t % W2;
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) a += sh[(w * NV + v) * W2 + j];
        tot[t] = a;
    }
    __syncthreads();
}

template <typename T, int NT, int IT>
__global__ __launch_bounds__(NT) void gepi_small_fwd(const T* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ noise,
                                                     const float* __restrict__ nw, const float* __restrict__ style, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int HW, int C, int act, int norm) {
    constexpr int VE = VecTraits<T>::VE, NV = GS_CG / VE, R = NT / NV, W2 = 2 * VE;
    __shared__ double sh[(NT / 64) * NV * W2 + 32];               // per-wave partials, [32] totals
    __shared__ float sstat[2 * GS_CG];                            // mean / rstd of the block's channels
    double* const tot = sh + (NT / 64) * NV * W2;
    const int b = blockIdx.y, t = threadIdx.x, v = t % NV, r = t / NV;
    const int cv = C / VE, vg = blockIdx.x * NV + v, c0 = vg * VE;
    float kb[VE], kw[VE], s0[VE], s1[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { kb[j] = 0.f; s0[j] = 0.f; s1[j] = 0.f; }
    if (bias) load_coef<VE>(bias + c0, kb);
    load_coef<VE>(nw + c0, kw);
    const T* xb = x + ((size_t)b * HW * cv + vg) * VE;
    const float* nzb = noise + (size_t)b * HW;
    uint4 xq[IT];
    float nz[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {                                 // all of the lane's loads in flight at once
        const int p = r + i * R;
        const bool ok = p < HW;
        const uint4 xl = *reinterpret_cast<const uint4*>(xb + (size_t)(ok ? p : 0) * cv * VE);     // (clamped address, value select: see the backward)
        const float nl = nzb[ok ? p : 0];
        xq[i] = make_uint4(ok ? xl.x : 0u, ok ? xl.y : 0u, ok ? xl.z : 0u, ok ? xl.w : 0u);
        nz[i] = ok ? nl : 0.f;
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        if (r + i * R < HW) {
            float xv[VE];
            unpack16<T>(xq[i], xv);
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                const float a = act_apply(xv[j] + kb[j] + kw[j] * nz[i], act);
                s0[j] += a; s1[j] += a * a;
            }
        }
    }
    gs_block_sum<VE, NT>(s0, s1, sh, tot);
    if (t < GS_CG) {
        float m = 0.f, rs = 1.f;
        if (norm) {                                                // as gepi_fin_stats
            const int vv = t / VE, j = t % VE;
            const double mm = tot[vv * W2 + j] / HW;
            double var = tot[vv * W2 + VE + j] / HW - mm * mm;
            if (var < 0.0) var = 0.0;
            m = (float)mm; rs = (float)(1.0 / sqrt(var + (double)GEPI_EPS));
        }
        sstat[t] = m; sstat[GS_CG + t] = rs;
        mean_out[(size_t)b * C + blockIdx.x * GS_CG + t] = m;
        rstd_out[(size_t)b * C + blockIdx.x * GS_CG + t] = rs;
    }
    __syncthreads();
    float km[VE], kr[VE], ks[VE], k1[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { km[j] = sstat[v * VE + j]; kr[j] = sstat[GS_CG + v * VE + j]; }
    load_coef<VE>(style + (size_t)b * 2 * C + c0, ks);
    load_coef<VE>(style + (size_t)b * 2 * C + C + c0, k1);
#pragma unroll
    for (int j = 0; j < VE; ++j) ks[j] += 1.f;
    T* yb = y + ((size_t)b * HW * cv + vg) * VE;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int p = r + i * R;
        if (p < HW) {
            float xv[VE];
            unpack16<T>(xq[i], xv);
#pragma unroll
            for (int j = 0; j < VE; ++j) {                         // the arithmetic of gepi_apply
                const float a = act_apply(xv[j] + kb[j] + kw[j] * nz[i], act);
                const float xh = (a - km[j]) * kr[j];
                xv[j] = xh * ks[j] + k1[j];
            }
            VecTraits<T>::store(yb + (size_t)p * cv * VE, xv);
        }
    }
}

// backward of the same: reduction (sum dy, sum dy*xh) -> the statistics-gradient coefficients and d style, then dx and the block's
// (image's) share of d noise-weight / d bias: part[b][C][2] = (sum dp*noise, sum dp), summed over images by gepi_fin_bwd2
template <typename T, int NT, int IT>
__global__ __launch_bounds__(NT) void gepi_small_bwd(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                     const float* __restrict__ bias, const float* __restrict__ noise, const float* __restrict__ nw,
                                                     const float* __restrict__ style, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     float* __restrict__ dstyle, double* __restrict__ part, int HW, int C, int act, int norm) {
    constexpr int VE = VecTraits<T>::VE, NV = GS_CG / VE, R = NT / NV, W2 = 2 * VE;
    __shared__ double sh[(NT / 64) * NV * W2 + 32];
    __shared__ float scoef[2 * GS_CG];                            // [16] k1, [16] k2
    double* const tot = sh + (NT / 64) * NV * W2;
    const int b = blockIdx.y, t = threadIdx.x, v = t % NV, r = t / NV;
    const int cv = C / VE, vg = blockIdx.x * NV + v, c0 = vg * VE;
    float kb[VE], kw[VE], km[VE], kr[VE], ks[VE], s0[VE], s1[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { kb[j] = 0.f; s0[j] = 0.f; s1[j] = 0.f; }
    if (bias) load_coef<VE>(bias + c0, kb);
    load_coef<VE>(nw + c0, kw);
    load_coef<VE>(mean + (size_t)b * C + c0, km);
    load_coef<VE>(rstd + (size_t)b * C + c0, kr);
    load_coef<VE>(style + (size_t)b * 2 * C + c0, ks);
#pragma unroll
    for (int j = 0; j < VE; ++j) ks[j] += 1.f;
    const size_t base = ((size_t)b * HW * cv + vg) * VE;
    const float* nzb = noise + (size_t)b * HW;
    uint4 xq[IT], gq[IT];
    float nz[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int p = r + i * R;
        const bool ok = p < HW;
        // (clamped address + select of the VALUE: a ?: between a load and an addressable zero vector compiles to a flat load from a
        // scratch copy of the zeros -- seen in the ISA of the first version, 32 bytes of scratch per lane)
        const size_t po = (size_t)(ok ? p : 0) * cv * VE;
        const uint4 xl = *reinterpret_cast<const uint4*>(x + base + po), gl = *reinterpret_cast<const uint4*>(dy + base + po);
        const float nl = nzb[ok ? p : 0];
        xq[i] = make_uint4(ok ? xl.x : 0u, ok ? xl.y : 0u, ok ? xl.z : 0u, ok ? xl.w : 0u);
        gq[i] = make_uint4(ok ? gl.x : 0u, ok ? gl.y : 0u, ok ? gl.z : 0u, ok ? gl.w : 0u);
        nz[i] = ok ? nl : 0.f;
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        if (r + i * R < HW) {
            float xv[VE], gv[VE];
            unpack16<T>(xq[i], xv); unpack16<T>(gq[i], gv);
#pragma unroll
            for (int j = 0; j < VE; ++j) {                         // gepi_pass<T, 1>
                const float xh = (act_apply(xv[j] + kb[j] + kw[j] * nz[i], act) - km[j]) * kr[j];
                s0[j] += gv[j]; s1[j] += gv[j] * xh;
            }
        }
    }
    gs_block_sum<VE, NT>(s0, s1, sh, tot);
    if (t < GS_CG) {                                               // gepi_fin_bwd1
        const int vv = t / VE, j = t % VE, c = blockIdx.x * GS_CG + t;
        const double sdy = tot[vv * W2 + j], sdx = tot[vv * W2 + VE + j];
        if (dstyle) {
            dstyle[(size_t)b * 2 * C + c] = (float)sdx;            // d/d style[:,0] = sum dy*xh
            dstyle[(size_t)b * 2 * C + C + c] = (float)sdy;        // d/d style[:,1] = sum dy
        }
        const double sc = (double)style[(size_t)b * 2 * C + c] + 1.0;
        scoef[t] = norm ? (float)(sc * sdy / HW) : 0.f;
        scoef[GS_CG + t] = norm ? (float)(sc * sdx / HW) : 0.f;
    }
    __syncthreads();
    float k1[VE], k2[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { k1[j] = scoef[v * VE + j]; k2[j] = scoef[GS_CG + v * VE + j]; s0[j] = 0.f; s1[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int p = r + i * R;
        if (p < HW) {
            float xv[VE], gv[VE], ov[VE];
            unpack16<T>(xq[i], xv); unpack16<T>(gq[i], gv);
#pragma unroll
            for (int j = 0; j < VE; ++j) {                         // gepi_pass<T, 2>
                const float pp = xv[j] + kb[j] + kw[j] * nz[i];
                const float xh = (act_apply(pp, act) - km[j]) * kr[j];
                const float da = kr[j] * (gv[j] * ks[j] - k1[j] - xh * k2[j]);
                const float dp = act ? da * lrelu_slope(pp) : da;
                ov[j] = dp;
                s0[j] += dp * nz[i]; s1[j] += dp;
            }
            VecTraits<T>::store(dx + base + (size_t)p * cv * VE, ov);
        }
    }
    gs_block_sum<VE, NT>(s0, s1, sh, tot);
    if (t < GS_CG) {
        const int vv = t / VE, j = t % VE;
        double* o = part + ((size_t)b * C + blockIdx.x * GS_CG + t) * 2;
        o[0] = tot[vv * W2 + j]; o[1] = tot[vv * W2 + VE + j];
    }
}
// block shape by layer size: 256 threads x 2 pixels per lane up to 16x16 (bf16), 512 threads x 4 up to 32x32 (1024 threads x 2 caps the
// kernel at 128 registers: the bf16 backward spilled 40 bytes); a 64x64 layer would be 16 cached vectors per lane and stays two-pass
template <typename T> struct GsPlan {
    static constexpr int NV = GS_CG / VecTraits<T>::VE;
    static constexpr int MAX_HW = 4 * (512 / NV);                  // bf16: 1024, fp32: 512
    static int pick(int HW) { return HW <= 2 * (256 / NV) ? 0 : 1; }
};
// which layers take the one-launch kernels (SGX_GEPI_SMALL=0: A/B against the two-pass structure)
template <typename T> static bool gepi_small_ok(int B, int HW, int C) {
    static const int on = [] { const char* e = getenv("SGX_GEPI_SMALL"); return e ? atoi(e) : 1; }();
    // measured (tools/gepi_probe.py, kernel time forward / backward, two-pass -> one launch): batch 4: 4x4..16x16 18-21 / 30-34 -> 8-9 / 19-20 us,
    // 32x32 26 / 39 -> 16 / 36; batch 32: 16x16 26 / 39 -> 19 / 37, but 32x32 39 / 59 -> 60 / 125: a block reads 32 of every 128-byte line, and
    // once the layer (33 MB) no longer sits in L2 the other three quarters are fetched again by the blocks of the other channel groups
    const long bytes = (long)B * HW * C * (long)sizeof(T);
    return on && C % GS_CG == 0 && (long)B * (C / GS_CG) >= 64 && (HW <= GsPlan<T>::MAX_HW / 4 || (HW <= GsPlan<T>::MAX_HW && bytes <= (8L << 20)));
}

template <typename T>
static int gepi_fwd_t(const void* x, const float* bias, const float* noise, const float* nw, const float* style, void* y,
                      float* mean, float* rstd, void* ws, const double* pre_part, int pre_npart, int B, int HW, int C, int flags,
                      hipStream_t st) {
    const int act = (flags & SGX_EPI_ACT) ? SGX_ACT_LRELU : SGX_ACT_NONE, norm = (flags & SGX_EPI_NORM) ? 1 : 0;
    constexpr int VE = VecTraits<T>::VE;
    GepiGeom g = gepi_geom(B, HW, C, VE);
    double* part = static_cast<double*>(ws);
    float* ctab = gepi_ctab(ws, B, HW, C);                         // gepi_apply1's coefficient tables, behind the partials of both directions
    const double nb = (double)sizeof(T) * B * HW * C;
    if (norm && pre_part) {
        // the statistics pass already happened in the kernel that PRODUCED x (sgx_blur3x3_stats / the convolution's store
        // epilogue): pre_npart partial (sum a, sum a^2) pairs per (image, channel), same layout as gepi_pass<T, 0> writes
        if (gepi_apply1_on(C, nb)) {
            hipLaunchKernelGGL(gepi_fin_stats, dim3((B * C + 15) / 16), dim3(256), 0, st, pre_part, mean, rstd, B, C, pre_npart, HW, norm, ctab, bias, nw, style);
            SGX_LAUNCH_CHECK("gepi_fin_stats");
            SGX_NOTE(0.0, 2.0 * nb, "gepi_apply B%d HW%d C%d", B, HW, C);
            hipLaunchKernelGGL(gepi_apply1<T>, dim3((unsigned)(((size_t)HW * (C / VE) + 255) / 256), B), dim3(256), 6 * C * sizeof(float), st, (const T*)x, noise,
                               (const float*)ctab, (T*)y, HW, C, act);
            SGX_LAUNCH_CHECK("gepi_apply1");
            return 0;
        }
        hipLaunchKernelGGL(gepi_fin_stats, dim3((B * C + 15) / 16), dim3(256), 0, st, pre_part, mean, rstd, B, C, pre_npart, HW, norm, (float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
        SGX_LAUNCH_CHECK("gepi_fin_stats");
        SGX_NOTE(0.0, 2.0 * nb, "gepi_apply B%d HW%d C%d", B, HW, C);
        hipLaunchKernelGGL(gepi_apply<T>, dim3(g.nchunk, B), dim3(256), 0, st, (const T*)x, bias, noise, nw, style, mean, rstd,
                           (T*)y, HW, C, g.cvt, g.rows, g.chunk, act, (const double*)nullptr, 0, 0, (float*)nullptr, (float*)nullptr);
        SGX_LAUNCH_CHECK("gepi_apply");
        return 0;
    }
    if (gepi_small_ok<T>(B, HW, C)) {                   // (with or without the normalisation: one launch)
        SGX_NOTE(0.0, 2.0 * nb, "gepi_fwd1 B%d HW%d C%d", B, HW, C);
        const dim3 grid(C / GS_CG, B);
        switch (GsPlan<T>::pick(HW)) {
            case 0: hipLaunchKernelGGL((gepi_small_fwd<T, 256, 2>), grid, dim3(256), 0, st, (const T*)x, bias, noise, nw, style, (T*)y, mean, rstd, HW, C, act, norm); break;
            default: hipLaunchKernelGGL((gepi_small_fwd<T, 512, 4>), grid, dim3(512), 0, st, (const T*)x, bias, noise, nw, style, (T*)y, mean, rstd, HW, C, act, norm);
        }
        SGX_LAUNCH_CHECK("gepi_small_fwd");
        return 0;
    }
    if (norm) {
        SGX_NOTE(0.0, nb, "gepi_stats B%d HW%d C%d", B, HW, C);
        hipLaunchKernelGGL((gepi_pass<T, 0>), dim3(g.nchunk, B), dim3(256), 256 * 2 * VE * sizeof(double), st, (const T*)x,
                           (const T*)nullptr, (T*)nullptr, bias, noise, nw, style, mean, rstd, (const float*)nullptr, part, HW, C,
                           g.cvt, g.rows, g.chunk, act, (const double*)nullptr, 0, 0, (float*)nullptr);
        SGX_LAUNCH_CHECK("gepi_stats");
    }
    // the statistics' finalize (mean / rstd from the per-chunk partials) rides in the apply pass: every block sums the partials of
    // its image itself (SGX_GEPI_FOLD=0: the separate gepi_fin_stats launch, A/B)
    static const int fold = [] { const char* e = getenv("SGX_GEPI_FOLD"); return e ? atoi(e) : 1; }();
    if (gepi_apply1_on(C, nb)) {
        hipLaunchKernelGGL(gepi_fin_stats, dim3((B * C + 15) / 16), dim3(256), 0, st, (const double*)part, mean, rstd, B, C, g.nchunk, HW, norm, ctab, bias, nw, style);
        SGX_LAUNCH_CHECK("gepi_fin_stats");
        SGX_NOTE(0.0, 2.0 * nb, "gepi_apply B%d HW%d C%d", B, HW, C);
        hipLaunchKernelGGL(gepi_apply1<T>, dim3((unsigned)(((size_t)HW * (C / VE) + 255) / 256), B), dim3(256), 6 * C * sizeof(float), st, (const T*)x, noise,
                           (const float*)ctab, (T*)y, HW, C, act);
        SGX_LAUNCH_CHECK("gepi_apply1");
        return 0;
    }
    if (fold) {
        SGX_NOTE(0.0, 2.0 * nb, "gepi_apply B%d HW%d C%d", B, HW, C);
        hipLaunchKernelGGL(gepi_apply<T>, dim3(g.nchunk, B), dim3(256), 512 * sizeof(double) + 2 * C * sizeof(float), st, (const T*)x, bias, noise,
                           nw, style, mean, rstd, (T*)y, HW, C, g.cvt, g.rows, g.chunk, act, (const double*)part, norm ? g.nchunk : 0, norm, mean, rstd);
        SGX_LAUNCH_CHECK("gepi_apply");
        return 0;
    }
    hipLaunchKernelGGL(gepi_fin_stats, dim3((B * C + 15) / 16), dim3(256), 0, st, (const double*)part, mean, rstd, B, C, g.nchunk, HW, norm, (float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
    SGX_LAUNCH_CHECK("gepi_fin_stats");
    SGX_NOTE(0.0, 2.0 * nb, "gepi_apply B%d HW%d C%d", B, HW, C);
    hipLaunchKernelGGL(gepi_apply<T>, dim3(g.nchunk, B), dim3(256), 0, st, (const T*)x, bias, noise, nw, style, mean, rstd,
                       (T*)y, HW, C, g.cvt, g.rows, g.chunk, act, (const double*)nullptr, 0, 0, (float*)nullptr, (float*)nullptr);
    SGX_LAUNCH_CHECK("gepi_apply");
    return 0;
}

template <typename T>
static int gepi_bwd_t(const void* dy, const void* x, const float* bias, const float* noise, const float* nw,
                      const float* style, const float* mean, const float* rstd, void* dx, float* dstyle, float* dnw,
                      float* dbias, void* ws, int B, int HW, int C, int flags, hipStream_t st) {
    constexpr int VE = VecTraits<T>::VE;
    const int act = (flags & SGX_EPI_ACT) ? SGX_ACT_LRELU : SGX_ACT_NONE, norm = (flags & SGX_EPI_NORM) ? 1 : 0;
    GepiGeom g = gepi_geom(B, HW, C, VE);
    double* partA = static_cast<double*>(ws);
    double* partB = partA + (size_t)B * g.nchunk * C * 2;
    float* coef = reinterpret_cast<float*>(partB + (size_t)B * g.nchunk * C * 2);
    const size_t shb = 256 * 2 * VE * sizeof(double);
    const double nb = (double)sizeof(T) * B * HW * C;
    if (gepi_small_ok<T>(B, HW, C)) {
        SGX_NOTE(0.0, 3.0 * nb, "gepi_bwd B%d HW%d C%d", B, HW, C);
        const dim3 grid(C / GS_CG, B);
        switch (GsPlan<T>::pick(HW)) {
            case 0: hipLaunchKernelGGL((gepi_small_bwd<T, 256, 2>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, (T*)dx, bias, noise, nw, style, mean, rstd, dstyle, partB, HW, C, act, norm); break;
            default: hipLaunchKernelGGL((gepi_small_bwd<T, 512, 4>), grid, dim3(512), 0, st, (const T*)x, (const T*)dy, (T*)dx, bias, noise, nw, style, mean, rstd, dstyle, partB, HW, C, act, norm);
        }
        SGX_LAUNCH_CHECK("gepi_small_bwd");
        hipLaunchKernelGGL(gepi_fin_bwd2, dim3((C + 3) / 4), dim3(256), 0, st, partB, dnw, dbias, B, C, 1);
        SGX_LAUNCH_CHECK("gepi_fin_bwd2");
        return 0;
    }
    SGX_NOTE(0.0, 2.0 * nb, "gepi_bwd1 B%d HW%d C%d", B, HW, C);
    hipLaunchKernelGGL((gepi_pass<T, 1>), dim3(g.nchunk, B), dim3(256), shb, st, (const T*)x, (const T*)dy, (T*)nullptr, bias,
                       noise, nw, style, mean, rstd, (const float*)nullptr, partA, HW, C, g.cvt, g.rows, g.chunk, act,
                       (const double*)nullptr, 0, 0, (float*)nullptr);
    SGX_LAUNCH_CHECK("gepi_bwd1");
    static const int fold = [] { const char* e = getenv("SGX_GEPI_FOLD"); return e ? atoi(e) : 1; }();
    if (gepi_bwd2s_on(C, VE, nb)) {
        float* ctab = gepi_ctab(ws, B, HW, C);
        double* part2 = gepi_part2s(ws, B, HW, C);
        const int nblk = gepi_bwd2s_blocks(HW, C, VE), cv = C / VE;
        hipLaunchKernelGGL(gepi_fin_bwd1, dim3((B * C + 15) / 16), dim3(256), 0, st, (const double*)partA, style, dstyle, coef, B, C, g.nchunk, HW, norm, ctab, bias, nw,
                           mean, rstd);
        SGX_LAUNCH_CHECK("gepi_fin_bwd1");
        SGX_NOTE(0.0, 3.0 * nb, "gepi_bwd2 B%d HW%d C%d", B, HW, C);
        const size_t shs = (size_t)(7 * C + 2) * sizeof(float) + (size_t)16 * cv * 2 * VE * sizeof(double);
        hipLaunchKernelGGL(gepi_bwd2s<T>, dim3((unsigned)nblk, B), dim3(256), shs, st, (const T*)x, (const T*)dy, (T*)dx, noise, (const float*)ctab, part2, HW, C, act);
        SGX_LAUNCH_CHECK("gepi_bwd2s");
        // B * nblk partial rows -> <= 256 (into the head's partB region, sized for B * nchunk >= 256 rows at these tensor sizes) -> the two gradients
        const int K = B * nblk, NB = K < 256 ? K : 256;
        if ((size_t)NB <= (size_t)B * g.nchunk && 2 * C <= 256) {
            hipLaunchKernelGGL(gepi_rows_prereduce, dim3(NB), dim3(256), 256 * sizeof(double), st, (const double*)part2, partB, K, 2 * C);
            SGX_LAUNCH_CHECK("gepi_rows_prereduce");
            hipLaunchKernelGGL(gepi_fin_bwd2, dim3((C + 3) / 4), dim3(256), 0, st, (const double*)partB, dnw, dbias, 1, C, NB);
        } else {
            hipLaunchKernelGGL(gepi_fin_bwd2, dim3((C + 3) / 4), dim3(256), 0, st, (const double*)part2, dnw, dbias, B, C, nblk);
        }
        SGX_LAUNCH_CHECK("gepi_fin_bwd2");
        return 0;
    }
    if (fold) {          // gepi_fin_bwd1's work (dstyle, the statistics-gradient coefficients) in the prologue of the apply pass
        SGX_NOTE(0.0, 3.0 * nb, "gepi_bwd2 B%d HW%d C%d", B, HW, C);
        if (sgx_nt_for(nb))
            hipLaunchKernelGGL((gepi_pass<T, 2, true>), dim3(g.nchunk, B), dim3(256), shb + 2 * C * sizeof(float), st, (const T*)x, (const T*)dy, (T*)dx, bias,
                               noise, nw, style, mean, rstd, (const float*)nullptr, partB, HW, C, g.cvt, g.rows, g.chunk, act,
                               (const double*)partA, g.nchunk, norm, dstyle);
        else
        hipLaunchKernelGGL((gepi_pass<T, 2>), dim3(g.nchunk, B), dim3(256), shb + 2 * C * sizeof(float), st, (const T*)x, (const T*)dy, (T*)dx, bias,
                           noise, nw, style, mean, rstd, (const float*)nullptr, partB, HW, C, g.cvt, g.rows, g.chunk, act,
                           (const double*)partA, g.nchunk, norm, dstyle);
        SGX_LAUNCH_CHECK("gepi_bwd2");
    } else {
        hipLaunchKernelGGL(gepi_fin_bwd1, dim3((B * C + 15) / 16), dim3(256), 0, st, (const double*)partA, style, dstyle, coef, B, C, g.nchunk, HW, norm, (float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
        SGX_LAUNCH_CHECK("gepi_fin_bwd1");
        SGX_NOTE(0.0, 3.0 * nb, "gepi_bwd2 B%d HW%d C%d", B, HW, C);
        hipLaunchKernelGGL((gepi_pass<T, 2>), dim3(g.nchunk, B), dim3(256), shb, st, (const T*)x, (const T*)dy, (T*)dx, bias, noise,
                           nw, style, mean, rstd, (const float*)coef, partB, HW, C, g.cvt, g.rows, g.chunk, act,
                           (const double*)nullptr, 0, 0, (float*)nullptr);
        SGX_LAUNCH_CHECK("gepi_bwd2");
    }
    hipLaunchKernelGGL(gepi_fin_bwd2, dim3((C + 3) / 4), dim3(256), 0, st, partB, dnw, dbias, B, C, g.nchunk);
    SGX_LAUNCH_CHECK("gepi_fin_bwd2");
    return 0;
}

static int gepi_check(int B, int HW, int C, int dtype, size_t ws_bytes) {
    const int ve = dtype == SGX_F32 ? 4 : 8;
    SGX_REQUIRE(dtype == SGX_F32 || dtype == SGX_BF16, SGX_EINVAL, "gepi: bad dtype");
    SGX_REQUIRE(C % ve == 0, SGX_EUNSUPPORTED, "gepi: C=%d", C);
    const int cv = C / ve;
    SGX_REQUIRE(cv <= 256 ? (256 % cv == 0) : (cv % 256 == 0), SGX_EUNSUPPORTED, "gepi: C=%d", C);
    SGX_REQUIRE(ws_bytes >= sgx_gepi_ws_bytes(B, HW, C), SGX_EWORKSPACE, "gepi: workspace %zu < %zu", ws_bytes,
                sgx_gepi_ws_bytes(B, HW, C));
    return 0;
}

extern "C" int sgx_gepi_fwd(const void* x, const float* bias, const float* noise, const float* nw, const float* style, void* y,
                            float* mean, float* rstd, void* ws, size_t ws_bytes, const double* pre_part, int pre_npart, int B, int HW,
                            int C, int flags, int dtype, void* stream) {
    int rc = gepi_check(B, HW, C, dtype, ws_bytes);
    if (rc) return rc;
    SGX_REQUIRE(!pre_part || pre_npart > 0, SGX_EINVAL, "gepi_fwd: %d producer partials", pre_npart);
    if (dtype == SGX_F32)
        return gepi_fwd_t<float>(x, bias, noise, nw, style, y, mean, rstd, ws, pre_part, pre_npart, B, HW, C, flags, (hipStream_t)stream);
    return gepi_fwd_t<bf16_t>(x, bias, noise, nw, style, y, mean, rstd, ws, pre_part, pre_npart, B, HW, C, flags, (hipStream_t)stream);
}

// The statistics half of sgx_gepi_fwd alone: mean / rstd of a = act(x + bias + nw * noise) per (image, channel), from the pass
// over x or from the partials the producer of x already wrote -- for a consumer that applies the normalisation and style itself
// (sgx_rgb_out_epi: the LAST epilogue of the generator, whose output only feeds to_rgb).
template <typename T>
static int gepi_stats_t(const void* x, const float* bias, const float* noise, const float* nw, float* mean, float* rstd, void* ws,
                        const double* pre_part, int pre_npart, int B, int HW, int C, int flags, hipStream_t st) {
    const int act = (flags & SGX_EPI_ACT) ? SGX_ACT_LRELU : SGX_ACT_NONE, norm = (flags & SGX_EPI_NORM) ? 1 : 0;
    constexpr int VE = VecTraits<T>::VE;
    GepiGeom g = gepi_geom(B, HW, C, VE);
    double* part = static_cast<double*>(ws);
    if (norm && !pre_part) {
        SGX_NOTE(0.0, (double)sizeof(T) * B * HW * C, "gepi_stats B%d HW%d C%d", B, HW, C);
        hipLaunchKernelGGL((gepi_pass<T, 0>), dim3(g.nchunk, B), dim3(256), 256 * 2 * VE * sizeof(double), st, (const T*)x,
                           (const T*)nullptr, (T*)nullptr, bias, noise, nw, (const float*)nullptr, mean, rstd, (const float*)nullptr, part, HW, C,
                           g.cvt, g.rows, g.chunk, act, (const double*)nullptr, 0, 0, (float*)nullptr);
        SGX_LAUNCH_CHECK("gepi_stats");
    }
    hipLaunchKernelGGL(gepi_fin_stats, dim3((B * C + 15) / 16), dim3(256), 0, st, pre_part ? pre_part : (const double*)part, mean, rstd, B, C,
                       pre_part ? pre_npart : g.nchunk, HW, norm, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
    SGX_LAUNCH_CHECK("gepi_fin_stats");
    return 0;
}
extern "C" int sgx_gepi_stats(const void* x, const float* bias, const float* noise, const float* nw, float* mean, float* rstd, void* ws,
                              size_t ws_bytes, const double* pre_part, int pre_npart, int B, int HW, int C, int flags, int dtype, void* stream) {
    int rc = gepi_check(B, HW, C, dtype, ws_bytes);
    if (rc) return rc;
    SGX_REQUIRE(x && noise && nw && mean && rstd, SGX_EINVAL, "gepi_stats: null argument");
    SGX_REQUIRE(!pre_part || pre_npart > 0, SGX_EINVAL, "gepi_stats: %d producer partials", pre_npart);
    if (dtype == SGX_F32) return gepi_stats_t<float>(x, bias, noise, nw, mean, rstd, ws, pre_part, pre_npart, B, HW, C, flags, (hipStream_t)stream);
    return gepi_stats_t<bf16_t>(x, bias, noise, nw, mean, rstd, ws, pre_part, pre_npart, B, HW, C, flags, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Blur with the epilogue's statistics pass folded into its store: y = blur3x3(x) (the generator's conv0_up -> blur,
// models/CustomLayers.py:176-177) and, from the values just written (rounded to the storage type, exactly what gepi_apply
// will read back), the per-(image, channel) partial sums of a = act(y + bias[c] + nw[c]*noise[b,p]) and a^2 that
// gepi_pass<T, 0> would otherwise re-read the tensor for.  Block = image b, strip of rt (<= 64) output rows, 256/cvt columns;
// thread = (column, channel vector) walking the strip with the separable sliding window of blur3x3_kernel (pointwise.hip).
// Partials: part[((b * npart + blk) * C + c) * 2 + {0, 1}], npart = strips * column chunks (sgx_blur3x3_stats_nparts).
#ifndef BS_WAVES
#define BS_WAVES 2        // waves per SIMD the register allocation is asked for (175 VGPRs without spills; 128 spills 23)
#define BS_GROUP 4        // rows whose loads are issued together
#endif
template <typename T>
__global__ __launch_bounds__(256, BS_WAVES) void blur_stats_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ bias,
                                                         const float* __restrict__ noise, const float* __restrict__ nw,
                                                         double* __restrict__ part, int H, int W, int C, int cvt, int wchunks, int act,
                                                         int rt) {
    constexpr int VE = VecTraits<T>::VE;
    extern __shared__ double sh[];                                 // [256][2*VE]
    const int b = blockIdx.y, blk = blockIdx.x, sidx = blk / wchunks, wc = blk % wchunks;
    const int cols = 256 / cvt;                                    // columns per block
    const int tc = threadIdx.x % cvt, tr = threadIdx.x / cvt;
    const int cv = C / VE;
    const int w = wc * cols + tr, h0 = sidx * rt;
    const int h1 = h0 + rt < H ? h0 + rt : H;                      // output rows [h0, h1)
    const bool live = w < W;
    const bool hasl = w > 0, hasr = w + 1 < W;
    const size_t rstride = (size_t)W * cv * VE;
    for (int vb = 0; vb < cv; vb += cvt) {                         // uniform trip count (cv is a multiple of cvt)
        const int v = vb + tc, c0 = v * VE;
        float s0[VE], s1[VE], kb[VE], kw[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) { s0[j] = 0.f; s1[j] = 0.f; kb[j] = 0.f; }
        if (bias) load_coef<VE>(bias + c0, kb);
        load_coef<VE>(nw + c0, kw);
        if (live) {
            const size_t col = (((size_t)b * H * W + w) * cv + v) * VE;      // (b, row 0, w, v)
            const float* nzp = noise + (size_t)b * H * W + w;
            const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
            // raw 16-byte loads of input row r: left / centre / right pixel (zero outside the image)
            auto ldrow = [&](int r, uint4 (&q)[3]) {
                const bool ok = (unsigned)r < (unsigned)H;
                const T* src = x + col + (size_t)(ok ? r : 0) * rstride;
                q[1] = ok ? *reinterpret_cast<const uint4*>(src) : zero4;
                q[0] = (ok && hasl) ? *reinterpret_cast<const uint4*>(src - cv * VE) : zero4;
                q[2] = (ok && hasr) ? *reinterpret_cast<const uint4*>(src + cv * VE) : zero4;
            };
            auto hsum = [&](const uint4 (&q)[3], float (&h)[VE]) {           // horizontal [1,2,1]
                float l[VE], m[VE], rr[VE];
                unpack16<T>(q[0], l); unpack16<T>(q[1], m); unpack16<T>(q[2], rr);
#pragma unroll
                for (int j = 0; j < VE; ++j) h[j] = l[j] + 2.f * m[j] + rr[j];
            };
            float ha[VE], hb[VE], hc[VE];                                     // horizontal sums of rows ro-1, ro, ro+1
            {
                uint4 q[3];
                ldrow(h0 - 1, q); hsum(q, ha);
                ldrow(h0, q); hsum(q, hb);
            }
            constexpr int G = BS_GROUP;                                       // rows per group: all 3*G loads (+G noise) issued, then the math
            for (int rg = h0; rg < h1; rg += G) {
                uint4 q[G][3];
                float nz[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    ldrow(rg + g + 1 < h1 + 1 ? rg + g + 1 : -1, q[g]);
                    nz[g] = rg + g < h1 ? nzp[(size_t)(rg + g) * W] : 0.f;
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int ro = rg + g;                                    // output row ro from input rows ro-1, ro, ro+1
                    if (ro < h1) {
                        hsum(q[g], hc);
                        float o[VE];
#pragma unroll
                        for (int j = 0; j < VE; ++j) o[j] = (ha[j] + 2.f * hb[j] + hc[j]) * (1.f / 16.f);
                        const uint4 packed = pack16<T>(o);
                        *reinterpret_cast<uint4*>(y + col + (size_t)ro * rstride) = packed;
                        float st[VE];
                        unpack16<T>(packed, st);                              // the stored (rounded) values: what the apply pass reads back
#pragma unroll
                        for (int j = 0; j < VE; ++j) {
                            const float a = act_apply(st[j] + kb[j] + kw[j] * nz[g], act);
                            s0[j] += a; s1[j] += a * a;
                            ha[j] = hb[j]; hb[j] = hc[j];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < VE; ++j) { sh[threadIdx.x * 2 * VE + j] = (double)s0[j]; sh[threadIdx.x * 2 * VE + VE + j] = (double)s1[j]; }
        __syncthreads();
        // fixed-order sum over the block's columns (same two schemes as gepi_pass)
        const int NO = cvt * 2 * VE;
        double* outp = part + ((size_t)b * gridDim.x + blk) * C * 2;
        if (NO < 256 && cols > 1) {
            const int o = threadIdx.x % NO, slice = threadIdx.x / NO, otc = o / (2 * VE), oj = o % (2 * VE);
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < 2 * VE; ++r) acc += sh[((slice * 2 * VE + r) * cvt + otc) * 2 * VE + oj];
            __syncthreads();
            sh[threadIdx.x] = acc;
            __syncthreads();
            if ((int)threadIdx.x < NO) {
                double a = 0.0;
                for (int sl = 0; sl < 256 / NO; ++sl) a += sh[sl * NO + threadIdx.x];
                outp[((size_t)(vb + otc) * VE + (oj % VE)) * 2 + (oj / VE)] = a;
            }
        } else if (tr == 0) {
#pragma unroll
            for (int j = 0; j < VE; ++j) {
                double a0 = 0.0, a1 = 0.0;
                for (int r = 0; r < cols; ++r) {
                    a0 += sh[(r * cvt + tc) * 2 * VE + j];
                    a1 += sh[(r * cvt + tc) * 2 * VE + VE + j];
                }
                outp[(size_t)(c0 + j) * 2] = a0; outp[(size_t)(c0 + j) * 2 + 1] = a1;
            }
        }
        __syncthreads();
    }
}

// rows per thread: 64 (fp32 per-lane partials over <= 64 values, as in gepi_pass; the block reduction is amortised over a
// 64-row strip and the vertical halo is 2 rows in 66), halved while the launch would have fewer than 1024 blocks
struct BlurStatsGeom { int cvt, wchunks, rt, npart; };
static BlurStatsGeom blur_stats_geom(int B, int H, int W, int C, int ve) {
    BlurStatsGeom g;
    const int cv = C / ve;
    g.cvt = cv < 256 ? cv : 256;
    const int cols = 256 / g.cvt;
    g.wchunks = (W + cols - 1) / cols;
    g.rt = 64;
    while (g.rt > 8 && (long)B * ((H + g.rt - 1) / g.rt) * g.wchunks < 1024) g.rt >>= 1;
    g.npart = ((H + g.rt - 1) / g.rt) * g.wchunks;
    return g;
}
extern "C" int sgx_blur3x3_stats_nparts(int B, int H, int W, int C, int dtype) {
    return blur_stats_geom(B, H, W, C, dtype == SGX_F32 ? 4 : 8).npart;
}
extern "C" int sgx_blur3x3_stats(const void* x, void* y, const float* bias, const float* noise, const float* nw, double* part,
                                 size_t part_bytes, int B, int H, int W, int C, int act, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SGX_REQUIRE(dtype == SGX_F32 || dtype == SGX_BF16, SGX_EINVAL, "blur3x3_stats: bad dtype");
    const int ve = dtype == SGX_F32 ? 4 : 8;
    SGX_REQUIRE(C % ve == 0 && ((C / ve) <= 256 ? 256 % (C / ve) == 0 : (C / ve) % 256 == 0), SGX_EUNSUPPORTED, "blur3x3_stats: C=%d", C);
    SGX_REQUIRE(x && y && noise && nw && part, SGX_EINVAL, "blur3x3_stats: null argument");
    SGX_REQUIRE(act == SGX_ACT_NONE || act == SGX_ACT_LRELU, SGX_EINVAL, "blur3x3_stats: activation %d", act);
    const BlurStatsGeom g = blur_stats_geom(B, H, W, C, ve);
    SGX_REQUIRE(part_bytes >= (size_t)B * g.npart * C * 2 * sizeof(double), SGX_EWORKSPACE, "blur3x3_stats: partials buffer %zu < %zu",
                part_bytes, (size_t)B * g.npart * C * 2 * sizeof(double));
    SGX_NOTE(0.0, 2.0 * (dtype == SGX_F32 ? 4.0 : 2.0) * B * H * W * C, "blur_stats B%d %dx%d C%d", B, H, W, C);
    const size_t shb = 256 * 2 * ve * sizeof(double);
    if (dtype == SGX_F32)
        hipLaunchKernelGGL(blur_stats_kernel<float>, dim3(g.npart, B), dim3(256), shb, st, (const float*)x, (float*)y, bias, noise, nw, part, H, W, C, g.cvt, g.wchunks, act, g.rt);
    else
        hipLaunchKernelGGL(blur_stats_kernel<bf16_t>, dim3(g.npart, B), dim3(256), shb, st, (const bf16_t*)x, (bf16_t*)y, bias, noise, nw, part, H, W, C, g.cvt, g.wchunks, act, g.rt);
    SGX_LAUNCH_CHECK("blur_stats_kernel");
    return 0;
}

extern "C" int sgx_gepi_bwd(const void* dy, const void* x, const float* bias, const float* noise, const float* nw,
                            const float* style, const float* mean, const float* rstd, void* dx, float* dstyle, float* dnw,
                            float* dbias, void* ws, size_t ws_bytes, int B, int HW, int C, int flags, int dtype, void* stream) {
    int rc = gepi_check(B, HW, C, dtype, ws_bytes);
    if (rc) return rc;
    if (dtype == SGX_F32)
        return gepi_bwd_t<float>(dy, x, bias, noise, nw, style, mean, rstd, dx, dstyle, dnw, dbias, ws, B, HW, C, flags, (hipStream_t)stream);
    return gepi_bwd_t<bf16_t>(dy, x, bias, noise, nw, style, mean, rstd, dx, dstyle, dnw, dbias, ws, B, HW, C, flags, (hipStream_t)stream);
}
