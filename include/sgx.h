/* sgx.h -- C ABI of libsgx_hip.so: the MI355X (gfx950) kernels behind the StyleGAN G+D training hot path.
 *
 * The reference (huangzh13/StyleGAN.pytorch) is pure Python and has no FFI: its seam for this path is the
 * nn.Module surface of models/CustomLayers.py, models/Blocks.py and models/GAN.py.  Every entry point below
 * replaces one composition of PyTorch ops that those modules issue; the comment above each one cites it
 * (file:line relative to the reference root).  stylegan/pytorch_amd/native.py binds them with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers.
 *   - activations are NHWC ("channels last"): x[b][h][w][c].  dtype: SGX_F32 or SGX_BF16 storage
 *     (accumulation is always fp32).  Parameters, statistics, RGB images and gradients of parameters are fp32.
 *   - `stream` is a hipStream_t passed as void*; the library never allocates, frees or synchronises.
 *     Scratch memory is supplied by the caller (`ws`, `ws_bytes`; query with the *_ws_bytes functions).
 *   - every function returns 0 on success, a negative SGX_E* code for argument errors, or a positive
 *     hipError_t; sgx_last_error() returns a thread-local message.  Nothing throws or aborts.
 *   - re-entrant: no global mutable state (backward runs on autograd worker threads).
 */
#ifndef SGX_H
#define SGX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { SGX_F32 = 0, SGX_BF16 = 1 };
/* LeakyReLU(0.2) / ReLU: the 'lrelu' | 'relu' nonlinearity of models/GAN.py:67-68,150-151,346-347.  The fused epilogues of
 * the convolution / linear kernels implement NONE and LRELU (what every shipped config uses); RELU is served by
 * sgx_bias_act + sgx_lrelu_bwd(slope 0) after the un-activated kernel.                                               */
enum { SGX_ACT_NONE = 0, SGX_ACT_LRELU = 1, SGX_ACT_RELU = 2 };
enum { SGX_EINVAL = -1, SGX_EUNSUPPORTED = -2, SGX_EWORKSPACE = -3 };

int sgx_version(void);
const char* sgx_last_error(void);
int sgx_clear_error(void);          /* forget the message and HIP's sticky last error (returns it); after a failed capture */
/* Stream ordering without a host object per fork: `waiter` waits for all work enqueued on `signaler` so far (pooled
 * events inside the library; valid during stream capture, where it becomes a graph edge).  Replaces the
 * torch.cuda.Stream.wait_stream calls a multi-stream implementation of models/GAN.py:595-655 would make.            */
int sgx_stream_wait_stream(void* waiter, void* signaler);
/* Hardware self-test of ds_read_b64_tr_b16 (the transpose read of the bf16 weight-gradient kernel): with lds[e] = e and
 * lane l addressing elements [4l,4l+4), writes the 4 int16 values each of the 64 lanes received to out256.          */
int sgx_selftest_tr16(void* out256, void* stream);

/* ---- per-launch profiler.  Every kernel the library launches can be bracketed by two HIP events recorded on the launch
 * stream itself.  sgx_prof_start(1, 0): record every launch (clears earlier records); (2, i): record only launches of
 * the kernel that record i belongs to; (0, 0): stop, keeping the records.  sgx_prof_get waits for record i and returns
 * the demangled kernel name (as rocprofv3 prints it), the milliseconds between its events, and the flops, algorithmic
 * bytes and layer description attached by the entry point that launched it (0 / "" when it attached none). */
int sgx_prof_start(int mode, int only_of);
int sgx_prof_count(void);
int sgx_prof_get(int i, char* name, int name_cap, float* ms, double* flops, double* bytes, char* desc, int desc_cap);

/* ---------------------------------------------------------------- convolutions (MFMA implicit GEMM)
 * Packed weight layout for all three: w[tap][n][k], n = output channel of THIS launch, k = reduction
 * channel, k contiguous; dtype = activation dtype.  bias (fp32, may be NULL) and act are fused in the store.
 *
 * sgx_conv3x3: EqualizedConv2d plain path, F.conv2d(x, W*w_mul, b, padding=1) -- models/CustomLayers.py:170-171;
 *   also its data gradient (call with the flipped/transposed pack).
 *   y[b,h,w,n] = act(bias[n] + sum_{ty,tx,k} x[b,h+ty-1,w+tx-1,k] * w[ty*3+tx][n][k])
 *   mask (may be NULL; a tensor shaped and typed like y): y *= (mask > 0 ? 1 : 0.2) in the store.  As a DATA-GRADIENT launch
 *   of a layer whose input is the LeakyReLU output of the layer below (DiscriminatorBlock conv0 after the previous block's
 *   conv1_down + act, models/Blocks.py:140-146), that is the activation's autograd applied where the gradient is produced
 *   instead of in a pass of its own; bit-identical to sgx_lrelu_bwd on the stored result.                              */
int sgx_conv3x3(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                int act, const void* mask, int dtype, void* stream);
/* sgx_conv4x4s2_down: fused conv+downscale, F.conv2d(x, W4, stride=2, padding=1) -- models/CustomLayers.py:158-165
 *   (== conv3x3 -> avg_pool2 of :166-168); also the data gradient of sgx_conv4x4s2_up.
 *   H,W = input size.  y[b,oy,ox,n] = act(bias[n] + sum_{ky,kx,k} x[b,2oy+ky-1,2ox+kx-1,k] * w[ky*4+kx][n][k])    */
int sgx_conv4x4s2_down(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin,
                       int Cout, int act, int dtype, void* stream);
/* sgx_conv4x4s2_up: fused upscale+conv, F.conv_transpose2d(x, W4, stride=2, padding=1) -- models/CustomLayers.py:143-152
 *   (== nearest-up -> conv3x3 of :153-154 with the kernel flipped); also the data gradient of
 *   sgx_conv4x4s2_down.  H,W = input (coarse) size, output is 2H x 2W.
 *   y[b,iy,ix,n] = sum over (oy,ky),(ox,kx) with 2oy+ky-1=iy, 2ox+kx-1=ix of x[b,oy,ox,k] * w[ky*4+kx][n][k]      */
int sgx_conv4x4s2_up(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout, int dtype,
                     void* stream);
/* Any of the three convolutions with the kernel generation named by the caller (A/B probes and parity tests of both).
 *   geo 0: sgx_conv3x3, 1: sgx_conv4x4s2_down, 2: sgx_conv4x4s2_up (H, W = input size; geo 2 takes no bias / activation);
 *   variant 0: first-generation kernel (16x16 MFMA tiles, register-staged single LDS stage; any shape / dtype);
 *   variant 4 / 8: second-generation bf16 kernel (32x32x16 MFMA, LDS-DMA double-buffered stages; the stride-2 convolution
 *   as four polyphase 2x2 convolutions, the transposed one with its four output-parity classes in one block) with 4- / 8-wave
 *   blocks; needs Cin % 32 == 0, Cout % 32 == 0 (64 for geo 0) and a tile-grid width % 32 == 0 (SGX_EUNSUPPORTED otherwise).
 *   The three entry points above choose by themselves.                                                               */
int sgx_conv_variant(int geo, const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                     int act, int dtype, int variant, void* stream);
/* Round 6: split-K for the bf16 stride-2 launches that leave most of the chip idle (the 512-channel conv1_down layers at 16x16 -> 8x8 and
 * 8x8 -> 4x4 of a small batch: 32-128 blocks, each walking 16 K-chunks behind a global-load latency).  sgx_conv_splitk_ws_bytes: 0 = the shape
 * does not split (call sgx_conv3x3 / sgx_conv4x4s2_down / sgx_conv4x4s2_up; always 0 for geo 0 and 2, whose split forms were measured no better
 * and are not built), else the bytes of fp32 partials [ksplit][output pixel][Cout] sgx_conv_splitk needs.  sgx_conv_splitk: geo 1 =
 * sgx_conv4x4s2_down's convolution (models/CustomLayers.py:160-171; bias and act as there, mask NULL), the reduction over input channels split
 * over ksplit <= 8 blocks, the partials summed in a fixed order by a second launch.  Other geometries: SGX_EUNSUPPORTED. */
size_t sgx_conv_splitk_ws_bytes(int geo, int B, int H, int W, int Cin, int Cout, int dtype);
int sgx_conv_splitk(int geo, const void* x, const void* w, const float* bias, void* y, const void* mask, int B, int H, int W, int Cin, int Cout,
                    int act, int dtype, void* ws, size_t ws_bytes, void* stream);

/* Host-only query: the launch configuration a convolution of this shape resolves to (geo 0: 3x3, 1: 4x4s2 down,
 * 2: 4x4s2 up; H,W = input size).  cfg5 = {KC, TH, TW, pixels per block, output-channel sub-tiles}: the template
 * arguments of conv_kernel<T, KC, geo, TH, TW, BP, CT> as rocprofv3 prints them.                                    */
int sgx_conv_config(int geo, int B, int H, int W, int Cin, int Cout, int dtype, int* cfg5);
/* sgx_pack_weight: runtime weight scaling + operand packing (`self.weight * self.w_mul` and the 3x3 -> 4x4 kernel
 * synthesis of models/CustomLayers.py:146-150,159-162) in one launch.  w: parameter [O][I][3][3] fp32.
 * Writes BOTH operand packs of the layer in the activation dtype: fwd[taps][O][Ipad] (the layer's convolution) and
 * adj[taps][Ipad][O] (its data gradient; 3x3 taps reversed).  Ipad >= I zero-pads the input-channel axis.
 *   S : scale*w                      -> sgx_conv3x3 / adj: sgx_conv3x3
 *   D : 0.25*scale*sum4shifts(w)     -> sgx_conv4x4s2_down / adj: sgx_conv4x4s2_up
 *   U : scale*sum4shifts(w)          -> sgx_conv4x4s2_up / adj: sgx_conv4x4s2_down    (fused upscale, input >= 64)
 *   UF: as U on the spatially flipped kernel (== nearest-up -> conv3x3 of the non-fused branch)       */
enum { SGX_PACK_S = 0, SGX_PACK_D = 1, SGX_PACK_U = 2, SGX_PACK_UF = 3 };
int sgx_pack_weight(const float* w, void* fwd, void* adj, int O, int I, int Ipad, int mode, float scale, int dtype,
                    void* stream);
/* the same for n parameters in ONE launch (all stale weights of a network after its optimizer step).  table: device
 * array of n rows of SGX_PACK_ROW 64-bit words  [w, fwd, adj, O, I, Ipad, mode, scale as fp32 bits, first block, blocks],
 * rows sorted by first block, blocks = sgx_pack_weight_blocks(O, Ipad), total_blocks = their sum; all packs in `dtype`. */
#define SGX_PACK_ROW 10
int sgx_pack_weight_blocks(int O, int Ipad);          /* blocks one parameter needs (32 x 32 channel tiles) */
int sgx_pack_weight_multi(const void* table, int n, int total_blocks, int dtype, void* stream);
/* weight gradients in the PARAMETER layout dW[O][I][3][3] (fp32): MFMA pixel-reduction into split partials (ws), then
 * one finishing kernel that sums the splits and applies the adjoint of sgx_pack_weight (autograd of F.conv2d /
 * F.conv_transpose2d w.r.t. weight composed with the reference's weight arithmetic).
 *   sgx_wgrad3x3_param : y = conv3x3(x, pack_S(w)):  x has Cx channels, dy has Cdy.  adjoint=1: the launch was the
 *                        layer's DATA-GRADIENT convolution (x: O channels, dy: Ipad channels) -- second-order path (R1).
 *   sgx_wgrad4x4s2_param: fine/coarse = the stride-2 pair's high/low resolution tensors (H,W = fine size);
 *                        mode D: coarse has O channels, fine has I;  mode U/UF: coarse has I, fine has O.
 *   db (nullable, fp32 [O]): also the bias gradient sum_{b,h,w} dy[.,o] (EqualizedConv2d bias, CustomLayers.py:178),
 *                        from the same pass over dy (one extra MFMA against a tile of ones) -- only where dy is the
 *                        O-channel side: adjoint=0 / mode D; otherwise SGX_EINVAL.
 *   accumulate           bit 0: dW += result, bit 1: db += result (a parameter used by several passes of one backward --
 *                        D on real and fake images, the R1 double backward -- accumulates in the finishing kernel
 *                        instead of a separate add per contribution).                                               */
size_t sgx_wgrad_ws_bytes(int taps, int B, int H, int W, int Ck, int Cn);
int sgx_wgrad3x3_param(const void* x, const void* dy, float* dW, float* db, void* ws, size_t ws_bytes, int B, int H, int W,
                       int Cx, int Cdy, int adjoint, float scale, int O, int I, int accumulate, int dtype, void* stream);
int sgx_wgrad4x4s2_param(const void* fine, const void* coarse, float* dW, float* db, void* ws, size_t ws_bytes, int B, int H,
                         int W, int Cfine, int Ccoarse, int mode, float scale, int O, int I, int accumulate, int dtype,
                         void* stream);

/* ---------------------------------------------------------------- memory-bound layer pieces
 * y = act(x + bscale*bias[c])               bias after blur / avgpool: models/CustomLayers.py:178-179; Blocks.py:142,146
 *   bscale = the layer's b_mul (lrmul; 0.01 in the mapping network, models/CustomLayers.py:94-95,101-102)           */
int sgx_bias_act(const void* x, const float* bias, float bscale, void* y, size_t npix, int C, int act, int dtype,
                 void* stream);
/* dx = scale * dy * (y > 0 ? 1 : slope)     autograd of nn.LeakyReLU(0.2) (slope 0.2) / torch.relu (slope 0); y is the
 * activation OUTPUT.  scale (1 = none; scale_dev != NULL: read from device memory instead) folds the fade-in coefficient of
 * the branch the activation sits on (x = alpha*straight + (1-alpha)*residual, models/GAN.py:427) into the same pass    */
int sgx_lrelu_bwd(const void* dy, const void* y, void* dx, size_t n, float slope, float scale, const float* scale_dev, int dtype,
                  void* stream);
/* sgx_lrelu_bwd with y given as sign bits (1 bit per element, bits[i] = the 8 channels of 16-byte vector i); bf16 */
int sgx_lrelu_bwd_bits(const void* dy, const void* bits, void* dx, size_t n, float slope, float scale, const float* scale_dev, int dtype,
                       void* stream);
/* The newest discriminator block's tail in one kernel (models/Blocks.py:143-146 + models/GAN.py:425-427): the stride-2 convolution,
 * its bias and LeakyReLU, and the fade-in lerp with the residual branch:  y = alpha * lrelu(conv(x) + bias) + beta * resid  (resid shaped
 * like y; the lerp is applied to the bf16-rounded activation, exactly as sgx_axpby on the stored tensor).  bits [B][H/2][W/2][Cout/8]:
 * the sign bits of the activation -- its LeakyReLU-backward mask (sgx_lrelu_bwd_bits); the activation itself is not stored.
 * _ok: 1 if the shape has the variant (bf16, second-generation stride-2 kernel), else sgx_conv4x4s2_down + sgx_axpby. */
int sgx_conv4x4s2_down_fade_ok(int B, int H, int W, int Cin, int Cout, int dtype);
int sgx_conv4x4s2_down_fade(const void* x, const void* w, const float* bias, const void* resid, float alpha, float beta, const float* ab_dev,
                            void* y, void* bits, int B, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* Round 5: the same with the residual branch resid = from_rgb(pimg) (models/GAN.py:423-427: the 1x1 convolution of the DOWN-SAMPLED image)
 * evaluated in the store: resid[p][c] = bf16((rb[c] * bs1) * bs2 + (r * (ws * wr[c][0]) + g * (ws * wr[c][1])) + b * (ws * wr[c][2])), (r, g, b) =
 * pimg[p] -- sgx_rgb_in's arithmetic, bit for bit; pimg fp32 [B][H/2][W/2][3], wr = from_rgb.weight [Cout][3], ws = its w_mul (x the
 * residual's prescale), rb = from_rgb.bias or NULL.  sgx_fade_rgb_bwd is the backward of that tail in ONE pass over g = dL/dy:
 * gy = (alpha g) slope(bits) (bit for bit sgx_lrelu_bwd_bits), dwr / drb (written, or accumulated per bit 0 / 1 of acc; NULL = skipped) =
 * beta ws sum_p g pimg resp. beta bs sum_p g, and (gpimg != NULL) gpimg[p][j] = beta ws sum_c g[p][c] wr[c][j]; alpha / beta from the call
 * or ab_dev[0], ab_dev[1]; ws: sgx_fade_rgb_bwd_ws_bytes.  bf16, C in {32, 64, 128}. */
int sgx_conv4x4s2_down_fade_rgb(const void* x, const void* w, const float* bias, const float* pimg, const float* wr, float ws, const float* rb,
                                float bs1, float bs2, float alpha, float beta, const float* ab_dev, void* y, void* bits, int B, int H, int W,
                                int Cin, int Cout, int dtype, void* stream);
size_t sgx_fade_rgb_bwd_ws_bytes(size_t npix, int C);
int sgx_fade_rgb_bwd(const void* g, const void* bits, const float* pimg, const float* wr, float ws, float bs, float alpha, float beta,
                     const float* ab_dev, void* gy, float* dwr, float* drb, int acc, float* gpimg, void* wsbuf, size_t ws_bytes, size_t npix, int C,
                     int dtype, void* stream);    /* ab_dev (nullable): [alpha, beta] in device memory instead (graph replay) */
/* Round 6: the parameter-gradient half on its own -- after a sgx_fade_rgb_bwd call with dwr = drb = NULL (which leaves the block partials in
 * wsbuf), on any stream ordered behind it: the step accumulates EVERY .grad on one side stream (two backward branches of the D step run on
 * different streams and both reach from_rgb of the residual branch). */
int sgx_fade_rgb_bwd_finish(const void* wsbuf, size_t ws_bytes, size_t npix, int C, float ws, float bs, float beta, const float* ab_dev,
                            float* dwr, float* drb, int acc, void* stream);
/* Round 6: the adjoint of sgx_fade_rgb_bwd's data half (gy, gpimg as functions of g) -- what the R1 double backward needs of this tail -- in one
 * pass: out = bf16(t1 + t2), t1 = bf16((alpha ggy) slope(bits)) (sgx_lrelu_bwd_bits on ggy), t2 = bf16(sgx_rgb_in(ggp; wr, ws)) [ab_dev: then
 * bf16(beta t2)]: the roundings of the three passes it replaces are kept, the result is theirs bit for bit.  ggy or ggp may be NULL.  A host
 * beta rides in ws (pass 1).  bf16, C in {32, 64, 128}. */
int sgx_fade_rgb_bwd2(const void* ggy, const float* ggp, const void* bits, const float* wr, float ws, float alpha, float beta, const float* ab_dev,
                      void* out, size_t npix, int C, int dtype, void* stream);
/* out = alpha*a + beta*b (b may be NULL)    fade-in lerp: models/GAN.py:202,427,586                                 */
int sgx_axpby(const void* a, const void* b, void* out, float alpha, float beta, size_t n, int dtype, void* stream);
/* same with the coefficients read from device memory (alpha_dev[0], beta_dev[0]): the fade-in alpha changes every
 * iteration (models/GAN.py:744), so a captured hipGraph of the step must not bake it into kernel arguments */
int sgx_axpby_dev(const void* a, const void* b, void* out, const float* alpha_dev, const float* beta_dev, size_t n, int dtype,
                  void* stream);
/* depthwise [1,2,1]x[1,2,1]/16 blur, zero pad   BlurLayer: models/CustomLayers.py:251-276 (self-adjoint)           */
int sgx_blur3x3(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
/* "LeakyReLU(0.2) then blur" (discriminator block, models/Blocks.py:140-142) with the activation folded into the blur pass:
 * mode 0: y = blur(x);  1: y = blur(lrelu(x))  [forward];  2: y = blur(x) * slope(z)  [backward: z = the pre-activation];
 * 3: y = blur(x * slope(z))  [backward of mode 2 w.r.t. x: the R1 double backward].  slope(z) = z > 0 ? 1 : 0.2 */
int sgx_blur3x3_act(const void* x, const void* z, void* y, int B, int H, int W, int C, int mode, int dtype, void* stream);
/* modes 2 / 3 with z as SIGN BITS: bits[pixel][C/8] bytes, bit j of byte v = (z[8v + j] > 0), as sgx_conv3x3_signbits writes them:
 * the backward passes of the block read 1 bit per mask element instead of 16 (bf16 only). */
int sgx_blur3x3_bits(const void* x, const void* bits, void* y, int B, int H, int W, int C, int mode, int dtype, void* stream);
/* BlurLayer with any other filter (models/CustomLayers.py:251-276: kernel = outer(f,f) [/sum] [flipped], F.conv2d with
 * groups=C, padding (K-1)//2):  y[oy][ox] = sum_{i,j} taps[i*K+j] * x[oy+i-pad][ox+j-pad], zero outside, K <= 7.
 * taps_host: K*K floats in HOST memory (copied into the launch).  x: [B][IH][IW][C], y: [B][OH][OW][C] with
 * OH <= IH+2*pad-K+1.  The adjoint is the same call with the taps flipped, pad' = K-1-pad and the sizes swapped.      */
int sgx_blur_kxk(const void* x, void* y, const float* taps_host, int K, int pad, int B, int IH, int IW, int OH, int OW, int C,
                 int dtype, void* stream);
/* Input pipeline on the device: uint8 images -> NHWC activations, (v/255 - 0.5)/0.5 in fp32 -- torchvision's
 * ToTensor + Normalize((.5,.5,.5),(.5,.5,.5)) of data/transforms.py:20-33; flip[b] != 0 mirrors image b horizontally
 * (RandomHorizontalFlip, decision drawn by the caller; NULL: none).  src: [B][H][W][3] bytes, or [B][3][H][W] if src_chw.
 * W must be a multiple of 4.                                                                                          */
int sgx_images_u8_to_nhwc(const void* src, void* dst, const int* flip, int B, int H, int W, int src_chw, int dtype, void* stream);
/* y = scale * (2x2 block sum)   scale .25: AvgPool2d(2) models/GAN.py:382,423; Downscale2d CustomLayers.py:60-64   */
int sgx_pool2(const void* x, void* y, int B, int H, int W, int C, float scale, int dtype, void* stream);
/* y = scale * nearest_up2(x)    Upscale2d CustomLayers.py:27-36; F.interpolate(scale_factor=2) models/GAN.py:173     */
int sgx_up2(const void* x, void* y, int B, int H, int W, int C, float scale, int dtype, void* stream);
/* Round 6: y = a + scale * nearest-up(x), a and y shaped [B][2H][2W][C] -- the adjoint of the 2x2 pool AND the sum with the other gradient of
 * a tensor that feeds both a full-resolution branch and a pooled one (the discriminator's image: models/GAN.py:423-427), one pass.     */
int sgx_up2_add(const void* x, const void* a, void* y, int B, int H, int W, int C, float scale, int dtype, void* stream);
/* out[c] = scale * sum_p x[p][c]  (fp32 out)        bias gradient                                                    */
size_t sgx_colsum_ws_bytes(size_t npix, int C);
int sgx_colsum(const void* x, float* out, float scale, void* ws, size_t ws_bytes, size_t npix, int C, int dtype,
               void* stream);

/* 1x1 RGB convolutions; images are fp32 [p][3].  The weight is read IN PLACE from the parameter: element (j = rgb
 * channel, c = feature channel) is w[j*sj + c*sc] * wscale (wscale = w_mul); from_rgb parameters are [C][3][1][1]
 * (sj=1, sc=3), to_rgb parameters are [3][C][1][1] (sj=C, sc=1).
 *   sgx_rgb_in : y[p][c] = bias[c] + sum_j img[p][j]*W(j,c)      from_rgb: models/GAN.py:358-359,377-378,425-426
 *   sgx_rgb_out: img[p][j] = bias[j] + sum_c x[p][c]*W(j,c)      to_rgb:   models/GAN.py:157,167,199-200
 *   sgx_rgb_wgrad: dw[j*sj + c*sc] = wscale * sum_p img[p][j]*f[p][c]   (gradient in the parameter layout)          */
int sgx_rgb_in(const float* img, const float* w, int sj, int sc, float wscale, const float* bias, void* y, size_t npix,
               int C, int dtype, void* stream);
int sgx_rgb_out(const void* x, const float* w, int sj, int sc, float wscale, const float* bias, float* img, size_t npix,
                int C, int dtype, void* stream);
/* Round 6: y = add + sgx_rgb_in(img) (no bias; add and y shaped [npix][C] of `dtype`): to_rgb's data gradient joined with the gradient the
 * forked activation got from its other consumer (models/GAN.py:199-202 under fade-in), one pass, one rounding.                          */
int sgx_rgb_in_add(const float* img, const float* w, int sj, int sc, float wscale, const void* add, void* y, size_t npix, int C, int dtype,
                   void* stream);
size_t sgx_rgb_wgrad_ws_bytes(size_t npix, int C);
/* real images at the current depth with the fade-in blend (models/GAN.py:575-586): out = alpha * x + beta *
 * nearest_up2(avgpool2(x)) on fp32 [B][H][W][3] in one pass; ab_dev as in sgx_rgb_out_fade. */
int sgx_downsample_fade_rgb(const float* x, float* out, int B, int H, int W, float alpha, float beta, const float* ab_dev, void* stream);
/* the generator's output in one pass: img = alpha * to_rgb(x) + beta * nearest_up2(low) (models/GAN.py:199-202 with the 1x1
 * convolution commuted in front of the upsample): low = [B][H/2][W/2][3] fp32.  ab_dev != NULL: alpha, beta = ab_dev[0], ab_dev[1]
 * read on the device (fade-in coefficient of a captured step graph). */
int sgx_rgb_out_fade(const void* x, const float* w, int sj, int sc, float wscale, const float* bias, const float* low, float alpha,
                     float beta, const float* ab_dev, float* img, int B, int H, int W, int C, int dtype, void* stream);
int sgx_rgb_wgrad(const float* img, const void* f, float* dw, int sj, int sc, float wscale, void* ws, size_t ws_bytes,
                  size_t npix, int C, int dtype, void* stream);

/* ---------------------------------------------------------------- generator layer epilogue
 * LayerEpilogue (models/CustomLayers.py:219-248) = NoiseLayer (:191-200) -> LeakyReLU -> nn.InstanceNorm2d (:233,
 * eps 1e-5, biased variance) -> StyleMod (:210-216), with the conv's post-blur bias (:178-179) folded in:
 *   p = x + bias[c] + nw[c]*noise[b,hw];  a = lrelu(p);  xh = (a - mean[b,c]) * rstd[b,c]
 *   y = xh * (style[b,c] + 1) + style[b,C+c]
 * bias may be NULL.  mean/rstd ([B][C] fp32) are outputs of fwd and inputs of bwd.
 * bwd returns dx, dstyle[B][2C], dnw[C], dbias[C] (dbias may be NULL).
 * flags select the stages the module was built with (LayerEpilogue's use_* arguments, :224-246): SGX_EPI_ACT = the
 * LeakyReLU, SGX_EPI_NORM = the instance norm (off: mean 0 / rstd 1 are written and the backward drops the statistics
 * terms).  use_noise=False is nw = 0, use_styles=False is style = 0 (both exact).  The default is ACT|NORM.           */
enum { SGX_EPI_ACT = 1, SGX_EPI_NORM = 2 };
size_t sgx_gepi_ws_bytes(int B, int HW, int C);
/* pre_part (may be NULL): the instance-norm statistics pass was already done by the kernel that PRODUCED x -- pre_npart
 * partial (sum a, sum a^2) pairs per (image, channel), double [B][pre_npart][C][2] (sgx_blur3x3_stats, sgx_conv3x3_stats):
 * the epilogue then reads the tensor once (apply pass) instead of twice. */
int sgx_gepi_fwd(const void* x, const float* bias, const float* noise, const float* nw, const float* style, void* y,
                 float* mean, float* rstd, void* ws, size_t ws_bytes, const double* pre_part, int pre_npart, int B, int HW,
                 int C, int flags, int dtype, void* stream);
/* y = blur3x3(x) (the generator's upscale-conv -> blur, models/CustomLayers.py:176-177, Blocks.py:63-65) with the statistics
 * pass of the LayerEpilogue that follows (:232-233 after :224-229) folded into the store: part receives the per-block partial
 * sums of a = act(y + bias[c] + nw[c]*noise[b,p]) and a^2 (y as stored), sgx_blur3x3_stats_nparts(...) blocks per image.
 * bias may be NULL; act: SGX_ACT_NONE | SGX_ACT_LRELU. */
int sgx_blur3x3_stats_nparts(int B, int H, int W, int C, int dtype);
/* y = conv3x3(x, pack) (no bias, no activation: the generator's conv1, models/Blocks.py:86, whose bias the epilogue adds) with
 * the same statistics out of the convolution's store epilogue: one partial per (image, pixel tile).  sgx_conv3x3_stats_nparts:
 * tiles per image, or 0 when the shape has no fused variant (then: sgx_conv3x3 + the plain sgx_gepi_fwd).  bf16 only. */
int sgx_conv3x3_stats_nparts(int B, int H, int W, int Cin, int Cout, int dtype);
/* sgx_conv3x3 that also writes the sign bits of its output (the discriminator block's conv0, models/Blocks.py:139-140, whose
 * pre-activation is the LeakyReLU-backward mask): bits = [B][H][W][Cout/8] bytes.  _ok: 1 if the shape has the variant (bf16). */
int sgx_conv3x3_signbits_ok(int B, int H, int W, int Cin, int Cout, int dtype);
int sgx_conv3x3_signbits(const void* x, const void* w, const float* bias, void* y, void* bits, int B, int H, int W, int Cin, int Cout,
                         int act, const void* mask, int dtype, void* stream);
/* y = blur3x3(conv_transpose4x4s2(x, pack)) [* slope(mask)] in one kernel: the transposed convolution and the BlurLayer that
 * follows it -- generator conv0_up -> blur (models/CustomLayers.py:143-152,176-177) and, with mask = the pre-activation z, the
 * discriminator's backward through "LeakyReLU -> blur -> conv1_down" (models/Blocks.py:140-145: the adjoint of the stride-2
 * convolution is this transposed one, the blur is self-adjoint, slope(z) = z > 0 ? 1 : 0.2).  mask: [B][2H][2W][Cout] or NULL.
 * The kernel applies the UNNORMALISED [1,2,1]x[1,2,1] filter: the caller packs the weights with scale / 16.
 * sgx_conv4x4s2_up_blur_ok: 1 if the shape has the fused kernel (bf16), else run sgx_conv4x4s2_up + sgx_blur3x3(_act). */
int sgx_conv4x4s2_up_blur_ok(int B, int H, int W, int Cin, int Cout, int dtype);
int sgx_conv4x4s2_up_blur(const void* x, const void* w, void* y, const void* mask, int B, int H, int W, int Cin, int Cout,
                          int dtype, void* stream);
/* Round 5: the same operation as ONE 3x3 stride-1 convolution over the coarse grid to the four output parity classes, stored depth-to-space
 * (blur o transposed-convolution is a 6x6 stride-2 transposed kernel = 3x3 taps per class; the blur's zero padding of the FINE grid is
 * restored at the image border by correction taps).  Replaces conv_transpose2d + BlurLayer (models/CustomLayers.py:143-152,175-177) and, in
 * the discriminator's backward, the adjoint of LeakyReLU -> blur -> conv1_down (models/Blocks.py:140-146) with the mask as sign bits.
 *   sgx_pack_upblur : the layer's parameter w [O][I][3][3] (mode SGX_PACK_U / _UF: an up layer's own convolution, adjoint = 0; SGX_PACK_D:
 *                     the data gradient of a down layer, adjoint = 1; scale = w_mul / 16, the blur's normalisation) -> wc bf16 [9][4N][K] + [22][2N][K],
 *                     (N, K) = (O, I) resp. (I, O): 9 composite taps + 22 border-correction tiles, composed in fp32 and rounded once.
 *   sgx_conv_upblur : y[B][2H][2W][Cout] = blur3x3(conv_transpose(x)) [* slope(bits)], bits [B][2H][2W][Cout/8] or NULL.  bf16, Cin = 32,
 *                     Cout = 16, W % 32 == 0 (sgx_conv_upblur_ok); other shapes: sgx_conv4x4s2_up_blur or the separate passes. */
int sgx_pack_upblur(const float* w, void* wc, int O, int I, int mode, int adjoint, float scale, void* stream);
int sgx_conv_upblur_ok(int B, int H, int W, int Cin, int Cout, int dtype);
int sgx_conv_upblur(const void* x, const void* wc, void* y, const void* maskbits, int B, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* the same with the mask given as SIGN BITS of z, bits[B][2H][2W][Cout/8] as sgx_conv3x3_signbits / sgx_rgbconv_fwd write them */
int sgx_conv4x4s2_up_blur_bits(const void* x, const void* w, void* y, const void* bits, int B, int H, int W, int Cin, int Cout,
                               int dtype, void* stream);
int sgx_conv3x3_stats(const void* x, const void* w, void* y, const float* ebias, const float* noise, const float* nw, double* part,
                      size_t part_bytes, int B, int H, int W, int Cin, int Cout, int dtype, void* stream);

/* ---- the discriminator's first layer pair at the current resolution as ONE 3-channel convolution (round 4).
 * Discriminator.forward applies from_rgb -- a 1x1 EqualizedConv2d with NO activation (models/GAN.py:353,413-427) -- and then the
 * block's conv0 (3x3), LeakyReLU and blur (models/Blocks.py:137-142).  The two linear maps compose into a 3x3 convolution of the
 * RGB image:  W'[o][j][tap] = s0 sr sum_i W0[o][i][tap] Wr[i][j];  from_rgb's bias br reaches conv0 only through taps that fall
 * inside the image (conv0 zero-pads from_rgb's OUTPUT), T[o][tap] = s0 bscale sum_i W0[o][i][tap] br[i], carried by a 4th input
 * channel that is 1 inside the image.  img, gi: fp32 [B][H][W][3];  w0: conv0.weight [C][C][3][3], s0 = its w_mul;  wr:
 * from_rgb.weight [C][3][1][1], sr = its w_mul;  br: from_rgb.bias (bscale = its b_mul) or NULL;  b0: conv0.bias * b_mul or NULL.
 * bf16 activations, C in {16, 32}, H % 16 == 0, W % 64 == 0 (sgx_rgbconv_ok); other shapes: sgx_rgb_in + sgx_conv3x3 + sgx_blur3x3_act.
 *   sgx_rgbconv_pack : wf [3][C][16] and wd [3][16][C] (activation dtype): the operand packs of the forward / image-gradient kernels.
 *   sgx_rgbconv_fwd  : epi 1: y = blur3x3(lrelu(conv(img) + b0)) and bits[pixel][C/8] = sign bits of the pre-activation (the
 *                      LeakyReLU-backward mask, as sgx_conv3x3_signbits writes them; may be NULL);  epi 0: y = conv(img) (no b0).
 *                      ones: 1 = with from_rgb's bias channel, 0 = without (the adjoint's own backward under R1's double backward).
 *   sgx_rgbconv_dgrad: gi = the gradient w.r.t. img given gz = the gradient w.r.t. the pre-activation (R1, models/Losses.py:197-211;
 *                      the generator step, models/GAN.py:640-655).
 *   sgx_rgbconv_wgrad: gradients of all four parameters from (img, gz) by the chain rule through the composition, written -- or, per
 *                      bit of `acc` (1: dw0, 2: db0, 4: dwr, 8: dbr), accumulated -- in the parameters' own layouts; NULL outputs
 *                      are skipped; ones = 0: no bias terms (db0 and dbr must be NULL).  ws: sgx_rgbconv_wgrad_ws_bytes. */
int sgx_rgbconv_ok(int B, int H, int W, int C, int dtype);
int sgx_rgbconv_pack(const float* w0, float s0, const float* wr, float sr, const float* br, float bscale, void* wf, void* wd, int C,
                     void* stream);
int sgx_rgbconv_fwd(const float* img, const void* wf, const float* b0, void* y, void* bits, int B, int H, int W, int C, int epi, int ones,
                    int dtype, void* stream);
/* process-wide A/B and probe state of sgx_rgbconv_fwd's epi 1 kernel (tests, tools/rgbconv_probe.py; initial values from
 * SGX_RGBCONV_FWD / SGX_RGBCONV_NIT, read once): fwd_variant 0 row-streaming (default), 1 / 2 the LDS-tile kernel with persistent
 * blocks / one tile per block, -1 = the default;  nit 1..8 = six-row steps per row block, 0 = chosen by launch size;  dbg: profiling
 * ablations of the row-streaming kernel (4 no output stores, 8 no sign bits -- wrong results by design), 0 = none. */
int sgx_rgbconv_tune(int fwd_variant, int nit, int dbg);
int sgx_rgbconv_dgrad(const void* gz, const void* wd, float* gi, int B, int H, int W, int C, int dtype, void* stream);
size_t sgx_rgbconv_wgrad_ws_bytes(int B, int H, int W, int C);
int sgx_rgbconv_wgrad(const float* img, const void* gz, int ones, const float* w0, float s0, const float* wr, float sr, const float* br,
                      float bscale, float* dw0, float* db0, float* dwr, float* dbr, int acc, void* ws, size_t ws_bytes, int B, int H, int W,
                      int C, int dtype, void* stream);
int sgx_blur3x3_stats(const void* x, void* y, const float* bias, const float* noise, const float* nw, double* part,
                      size_t part_bytes, int B, int H, int W, int C, int act, int dtype, void* stream);
int sgx_gepi_bwd(const void* dy, const void* x, const float* bias, const float* noise, const float* nw,
                 const float* style, const float* mean, const float* rstd, void* dx, float* dstyle, float* dnw,
                 float* dbias, void* ws, size_t ws_bytes, int B, int HW, int C, int flags, int dtype, void* stream);

/* ---------------------------------------------------------------- small fp32 pieces
 * PixelNormLayer on [B][C] rows: y = x * rsqrt(mean_c(x^2) + 1e-8)      models/CustomLayers.py:22-23                 */
int sgx_pixelnorm_fwd(const float* x, float* y, int B, int C, void* stream);
int sgx_pixelnorm_bwd(const float* dy, const float* x, float* dx, int B, int C, void* stream);
/* StddevLayer (models/CustomLayers.py:294-305), group = min(4,B), strided groups (sample i with i+M, i+2M, i+3M).
 *   fwd: y[b][hw][0..C) = x, y[b][hw][C] = stat[b % M], y[..][C+1..Cpad) = 0;   stat[m] = mean_{c,hw} sqrt(var_g + 1e-8)
 *   bwd: dx = dy[..][0..C) + (sum_{g,hw} dy[g*M+m][hw][C]) / (C*HW) * d / (G * s)
 *   bwd2 (second order, for R1): given ggx = cotangent of dx, returns its contribution to grad wrt dy (ddy) and x (gx) */
int sgx_mbstd_fwd(const void* x, void* y, int B, int HW, int C, int Cpad, int dtype, void* stream);
int sgx_mbstd_bwd(const void* dy, const void* x, void* dx, int B, int HW, int C, int Cpad, int dtype, void* stream);
int sgx_mbstd_bwd2(const void* ggx, const void* dy, const void* x, void* ddy, void* gx, int B, int HW, int C, int Cpad,
                   int dtype, void* stream);
/* R1 penalty head (models/Losses.py:210): out[0] = sum(x^2) over n fp32 values (deterministic two-stage reduce), and
 * its backward out = alpha * s[0] * x with the upstream scalar s on the device.                                      */
size_t sgx_sumsq_ws_bytes(void);
int sgx_sumsq_f32(const float* x, size_t n, void* ws, size_t ws_bytes, float* out, void* stream);
int sgx_scale_dev_f32(const float* x, const float* s, float alpha, float* out, size_t n, void* stream);
/* Logistic loss heads (models/Losses.py:213-229: mean softplus(fake) + mean softplus(-real); generator :227-229: mean softplus(-fake)),
 * times `scale` (1 / world size under data parallelism): loss[0] and the derivative w.r.t. every logit (g_fake [n_fake], g_real
 * [n_real]) from ONE launch instead of ~9 elementwise / reduction launches forward and as many backward.  generator != 0: real /
 * g_real are ignored (n_real must be 0).  fp32 logits.                                                                           */
int sgx_logistic_loss(const float* fake, int n_fake, const float* real, int n_real, float scale, int generator, float* loss,
                      float* g_fake, float* g_real, void* stream);
/* C[M][N] = alpha * op(A) * op(B) (+ beta*C), row-major fp32, MFMA f32 16x16x4.  EqualizedLinear
 * (models/CustomLayers.py:99-103: F.linear(x, W*w_mul)) and its gradients.  ta/tb: 0 = as stored, 1 = transposed:
 *   ta=0: A is [M][K], ta=1: A is [K][M];  tb=0: B is [K][N], tb=1: B is [N][K].                                      */
size_t sgx_gemm_ws_bytes(int M, int N, int K);
int sgx_gemm_f32(const float* A, const float* Bm, float* C, int M, int N, int K, int ta, int tb, float alpha,
                 void* ws, size_t ws_bytes, void* stream);
/* EqualizedLinear (models/CustomLayers.py:64-103) of the mapping network and the style affines, three launches:
 *   sgx_linear_fwd      : y[B][N] = act(w_mul * x[B][K] W[N][K]^T + b_mul * bias)        (bias nullable)
 *   sgx_linear_bwd_data : gx[B][K] = w_mul * gz W,  gz = gy * slope(y_act) (y_act nullable: no activation)
 *   sgx_linear_bwd_param: dW[N][K] = w_mul * gz^T x,  db[N] = b_mul * sum_b gz            (db nullable)           */
int sgx_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int N, int K, float w_mul, float b_mul,
                   int act, void* stream);
int sgx_linear_bwd_data(const float* gy, const float* y_act, const float* w, float* gx, int B, int N, int K, float w_mul,
                        void* stream);
int sgx_linear_bwd_param(const float* gy, const float* y_act, const float* x, float* dw, float* db, int B, int N, int K, float w_mul,
                         float b_mul, void* stream);
/* All style affines (StyleMod.lin, models/CustomLayers.py:203-216) of one generator forward in one launch, two for the
 * backward.  lm: layer-major dlatents [L][B][D] fp32.  table: G rows of SGX_STYLE_ROW 64-bit words
 *   [W (fp32 [N][D]), bias (fp32 [N] or 0), N, y offset in elements (= B * sum of earlier N), first tile (16 outputs per
 *    tile), w_mul as fp32 bits, b_mul as fp32 bits, layer index into lm];  total_tiles = sum of ceil(N/16).
 * y / gy: flat, group g at its offset as [B][N_g].  glm: [L][B][D] (rows of layers without a group are not written).
 * dw: flat [sum N][D], db: flat [sum N] (nullable).  B <= 32, D % 64 == 0. */
#define SGX_STYLE_ROW 8
int sgx_style_fwd(const float* lm, const void* table, float* y, int G, int B, int D, int total_tiles, void* stream);
int sgx_style_bwd_data(const float* gy, const void* table, float* glm, int G, int B, int D, int max_n, void* stream);
int sgx_style_bwd_param(const float* gy, const float* lm, const void* table, float* dw, float* db, int G, int B, int D,
                        int total_tiles, void* stream);

/* ---- the generator's LAST layer epilogue inside to_rgb (round 4).  The last LayerEpilogue of the synthesis network feeds only to_rgb
 * (models/GAN.py:199-202 after models/Blocks.py:87-88 / models/CustomLayers.py:219-248): instead of writing x2 = StyleMod(InstanceNorm(
 * lrelu(y + bias + nw noise))) and reading it back in the 1x1 convolution, to_rgb reads the convolution's output y and applies the
 * epilogue per element on the fly.  Default stack only (activation + instance norm + style; flags = SGX_EPI_ACT | SGX_EPI_NORM).
 *   sgx_gepi_stats    : mean / rstd [B][C] of the epilogue (the statistics half of sgx_gepi_fwd; pre_part as there).
 *   sgx_rgb_out_epi   : img = alpha * (wscale W x2 + rbias) + beta * nearest_up2(low)   (low may be NULL: then beta is ignored)
 *   sgx_rgb_wgrad_epi : dw = scale * sum_p g[p][j] x2[p][c] in the parameter's layout, db[j] = bscale * sum_p g[p][j] (db nullable);
 *                       g = the image gradient [B][HW][3] fp32.  ws: sgx_rgb_wgrad_epi_ws_bytes.
 * The data gradient is sgx_rgb_in (g -> d x2) followed by sgx_gepi_bwd on y, as for the separate ops. */
int sgx_gepi_stats(const void* x, const float* bias, const float* noise, const float* nw, float* mean, float* rstd, void* ws, size_t ws_bytes,
                   const double* pre_part, int pre_npart, int B, int HW, int C, int flags, int dtype, void* stream);
int sgx_rgb_out_epi(const void* y, const float* ebias, const float* noise, const float* nw, const float* style, const float* mean,
                    const float* rstd, const float* w, int sj, int sc, float wscale, const float* rbias, const float* low, float alpha, float beta,
                    const float* ab_dev, float* img, int B, int H, int W, int C, int dtype, void* stream);    /* ab_dev as above */
size_t sgx_rgb_wgrad_epi_ws_bytes(int B, int HW, int C);
int sgx_rgb_wgrad_epi(const void* y, const float* g, const float* ebias, const float* noise, const float* nw, const float* style, const float* mean,
                      const float* rstd, float* dw, float* db, int sj, int sc, float scale, float bscale, void* ws, size_t ws_bytes, int B, int HW,
                      int C, int dtype, void* stream);

/* ---------------------------------------------------------------- optimizer step (multi-tensor, fp32)
 * torch.optim.Adam step (models/GAN.py:529-533, 616-618, 652) over n tensors described by DEVICE arrays of
 * pointers/sizes.  Per tensor t (torch keeps a step count per parameter; parameters of inactive resolutions have
 * no gradient and are not listed): step_sizes[t] = lr / (1 - beta1^step_t), bc2_sqrts[t] = sqrt(1 - beta2^step_t).
 *   g = grad*grad_scale[0];  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= step_size * m / (sqrt(v)/bc2_sqrt + eps)
 * grad_scale (device scalar, may be NULL) is the global-norm clip coefficient (models/GAN.py:651).
 * update_average EMA (models/__init__.py:31-36): tgt = beta*tgt + (1-beta)*src.                                      */
int sgx_adam_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const int64_t* sizes, int n, float beta1, float beta2, float eps, const float* step_sizes,
                   const float* bc2_sqrts, const float* grad_scale, void* stream);
int sgx_ema_multi(float* const* tgt, const float* const* src, const int64_t* sizes, int n, float beta, void* stream);
/* nn.utils.clip_grad_norm_ (models/GAN.py:651) without a host sync: out[0] = sum over all tensors of sum(g^2),
 * out[1] = min(1, max_norm / (sqrt(out[0]) + 1e-6)); pass out+1 as grad_scale of sgx_adam_multi.
 * partial: caller scratch of n*32 doubles.                                                                            */
int sgx_gradnorm_clip_coef(const float* const* grads, const int64_t* sizes, int n, float max_norm, double* partial,
                           float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGX_H */
