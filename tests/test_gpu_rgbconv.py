"""-m gpu: the discriminator's first layer pair as ONE convolution of the RGB image (csrc/rgbconv.hip, round 4).

Reference composition: ``from_rgb`` (1x1 EqualizedConv2d, no activation, models/GAN.py:353,425) -> ``DiscriminatorBlock.conv0`` ->
LeakyReLU -> BlurLayer (models/Blocks.py:137-142).  Every kernel against the fp64 CPU oracle of that chain (forward, image
gradient, the four parameter gradients through the chain rule, the plain convolution of the double backward), and the whole
discriminator -- scores, R1 image gradient, every parameter gradient of the logistic + R1 loss -- with the composed kernel ON
against the same network with it OFF, both measured against the fp64 oracle.  bf16 storage: tolerances are those of one bf16
rounding of the operands (2^-9 per element) plus the store, written next to each check."""
import numpy as np
import pytest
import torch

import golden_util as gu
from gpu_util import DEV, MID, MID_DEPTH, load_into, mid_params, rel_err
from oracle import stylegan_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _lib():
    from stylegan.pytorch_amd import native
    assert torch.cuda.is_available()
    native.lib()


def make_params(C, dtype=torch.float64):
    p = {"w0": gu.seeded((C, C, 3, 3), 11, dtype), "b0": 0.3 * gu.seeded((C,), 12, dtype),
         "wr": gu.seeded((C, 3, 1, 1), 13, dtype), "br": 0.5 * gu.seeded((C,), 14, dtype)}
    return {k: v.requires_grad_(True) for k, v in p.items()}


def scales(C):
    return O.he_w_mul(C * 9, 2 ** 0.5), O.he_w_mul(3, 2 ** 0.5)


def chain64(img, p, bias=True):
    """fp64: from_rgb -> conv0 (pre-activation z) -> lrelu -> blur, reference models/GAN.py:425 + models/Blocks.py:139-142."""
    f = O.eq_conv2d(img, p["wr"], p["br"] if bias else None)
    z = O.eq_conv2d(f, p["w0"], p["b0"] if bias else None)
    return z, O.blur3(O.leaky_relu(z))


def dev_params(p):
    return {k: v.detach().float().to(DEV).requires_grad_(True) for k, v in p.items()}


CASES = [(16, 2, 32, 64), (16, 1, 48, 192), (32, 2, 16, 64), (32, 1, 48, 128), (16, 3, 160, 320)]


@pytest.mark.parametrize("C,B,H,W", CASES)
def test_forward_blur_and_sign_bits_vs_oracle(C, B, H, W, monkeypatch):
    from stylegan.pytorch_amd import functional as F
    from stylegan.pytorch_amd import native as N
    p = make_params(C)
    d = dev_params(p)
    img = gu.seeded((B, 3, H, W), 21, torch.float64)
    s0, sr = scales(C)
    with torch.no_grad():
        z, xb = chain64(img, p)
        x_nhwc = img.float().permute(0, 2, 3, 1).contiguous().to(DEV)
        assert F.rgbconv_ok(B, H, W, C, torch.bfloat16)
        got, bits = F.RgbConvBlurFn.apply(x_nhwc, d["w0"], d["b0"], d["wr"], d["br"], s0, sr)
        assert got.dtype == torch.bfloat16 and got.shape == (B, H, W, C) and bits.shape == (B, H, W, C // 8)
        # every row-block size the host may pick by launch size (6 nit - 2 rows per wave, sgx_rgbconv_tune: partial
        # last blocks, images of one block) writes the same bits (tools/rgbconv_check.py, profiles/r04_rgbconv_check.txt)
        for nit in (6, 4, 1):
            N.check(N.lib().sgx_rgbconv_tune(-1, nit, 0), "sgx_rgbconv_tune")
            gn, bn = F.RgbConvBlurFn.apply(x_nhwc, d["w0"], d["b0"], d["wr"], d["br"], s0, sr)
            assert torch.equal(gn.view(torch.int16), got.view(torch.int16)) and torch.equal(bn, bits), nit
        N.check(N.lib().sgx_rgbconv_tune(-1, 0, 0), "sgx_rgbconv_tune")
        e = rel_err(got.permute(0, 3, 1, 2), xb)
        # the unfused library path in bf16 (from_rgb output rounded, conv0 output rounded, blur output rounded) for scale
        f_old = F.call(F.RgbInFn, x_nhwc, d["wr"], d["br"], sr, torch.bfloat16)
        z_old = F.conv(f_old, d["w0"], d["b0"], "S", s0, ipad=C)
        e_old = rel_err(F.call(F.ActBlurFn, z_old).permute(0, 3, 1, 2), xb)
        print(f"[rgbconv fwd C{C} {H}x{W}] rel {e:.2e} (unfused bf16 path {e_old:.2e})")
        assert e <= 6e-3 and e <= 1.5 * e_old + 1e-3, (e, e_old)
        # sign bits: bit j of byte v = (pre-activation of channel 8v + j > 0); elements within bf16 noise of zero may differ
        zz = z.permute(0, 2, 3, 1)                                                         # [B,H,W,C]
        want = (zz > 0)
        b = bits.cpu().numpy()
        gotbits = np.unpackbits(b[..., None], axis=-1, bitorder="little").reshape(B, H, W, C).astype(bool)
        sure = (zz.abs() > 2e-2 * zz.abs().mean()).numpy()
        assert (gotbits[sure] == want.numpy()[sure]).all()
        assert (gotbits != want.numpy()).mean() < 2e-2


@pytest.mark.parametrize("C,B,H,W", CASES[:1] + CASES[2:3])
def test_plain_convolution_and_image_gradient_vs_oracle(C, B, H, W):
    from stylegan.pytorch_amd import functional as F
    p = make_params(C)
    d = dev_params(p)
    s0, sr = scales(C)
    img = gu.seeded((B, 3, H, W), 31, torch.float64).requires_grad_(True)
    gz = gu.seeded((B, C, H, W), 32, torch.float64)
    z, _ = chain64(img, p, bias=False)                       # the double-backward ops carry no biases
    (gi,) = torch.autograd.grad((z * gz).sum(), img)
    with torch.no_grad():
        x_nhwc = img.detach().float().permute(0, 2, 3, 1).contiguous().to(DEV)
        got_z = F.RgbConvPlainFn.apply(x_nhwc, d["w0"], d["wr"], d["br"], s0, sr)
        e_z = rel_err(got_z.permute(0, 3, 1, 2), z)
        gz_dev = gz.float().permute(0, 2, 3, 1).contiguous().to(DEV).bfloat16()
        got_gi = F.RgbConvAdjFn.apply(gz_dev, d["w0"], d["wr"], d["br"], s0, sr)
        assert got_gi.dtype == torch.float32 and got_gi.shape == (B, H, W, 3)
        e_g = rel_err(got_gi.permute(0, 3, 1, 2), gi)
    print(f"[rgbconv C{C}] plain conv rel {e_z:.2e}, image gradient rel {e_g:.2e}")
    assert e_z <= 6e-3 and e_g <= 6e-3, (e_z, e_g)
    # The plain convolution's real input is not an image but the R1 double backward's second-order gradient w.r.t. the image,
    # gamma / B * dD/dimg per pixel: 1e-6..1e-5 at 1024^2, below fp16's normal range (6.1e-5; the forward kernels take fp16
    # operands).  The kernel prescales each tile by a power of two from its own maximum: the relative error must not depend on the
    # magnitude of the input -- at any scale, and with the scale varying from tile to tile by orders of magnitude.
    with torch.no_grad():
        for scale in (1e-6, 3e-9, 1e-30, 2e4):
            got = F.RgbConvPlainFn.apply(x_nhwc * scale, d["w0"], d["wr"], d["br"], s0, sr)
            e = rel_err(got.permute(0, 3, 1, 2), z * scale)
            print(f"[rgbconv C{C}] plain conv of an input of magnitude {scale:g}: rel {e:.2e}")
            assert e <= 1.25 * e_z + 1e-4, (scale, e, e_z)
        ramp = torch.logspace(-8, 0, W, dtype=torch.float64)[None, None, None, :]          # eight decades across the row: every 64-column tile at its own scale
        got = F.RgbConvPlainFn.apply((img.detach() * ramp).float().permute(0, 2, 3, 1).contiguous().to(DEV), d["w0"], d["wr"], d["br"], s0, sr)
        zr, _ = chain64(img.detach() * ramp, p, bias=False)
        for c0 in range(0, W, 64):
            e = rel_err(got.permute(0, 3, 1, 2)[..., c0 + 1:c0 + 63], zr[..., c0 + 1:c0 + 63])
            assert e <= 1.5 * e_z + 1e-4, (c0, e, e_z)


@pytest.mark.parametrize("C,B,H,W", [(16, 2, 32, 64), (32, 3, 32, 128), (16, 5, 64, 256)])
def test_parameter_gradients_through_the_chain_rule_vs_oracle(C, B, H, W):
    """dW0, db0, dWr, dbr from (image, gradient of the pre-activation): written, accumulated, and without the bias channel."""
    from stylegan.pytorch_amd import functional as F
    p = make_params(C)
    d = dev_params(p)
    s0, sr = scales(C)
    img = gu.seeded((B, 3, H, W), 41, torch.float64)
    gz = gu.seeded((B, C, H, W), 42, torch.float64)
    x_nhwc = img.float().permute(0, 2, 3, 1).contiguous().to(DEV)
    gz_dev = gz.float().permute(0, 2, 3, 1).contiguous().to(DEV).bfloat16()
    gz_r = gz_dev.double().cpu().permute(0, 3, 1, 2)         # the oracle sees the same (bf16-rounded) upstream gradient
    names = ["w0", "b0", "wr", "br"]
    for ones in (True, False):
        z, _ = chain64(img, p, bias=ones)
        ks = names if ones else ["w0", "wr"]
        want = dict(zip(ks, torch.autograd.grad((z * gz_r).sum(), [p[k] for k in ks])))
        with torch.no_grad():
            outs = F._rgb_wgrad(x_nhwc, gz_dev, ones, d["w0"], d["b0"] if ones else None, d["wr"], d["br"], s0, sr, (True, True, True, True))
        got = dict(zip(names, outs))
        for k in names:
            if k not in ks:
                assert got[k] is None
                continue
            e = rel_err(got[k], want[k])
            print(f"[rgbconv wgrad C{C} B{B} ones={ones}] {k}: rel {e:.2e}")
            assert e <= 8e-3, (k, e)
    # accumulation into existing .grad tensors (the training step's mode): twice the gradient, bit-deterministic
    for k in names:
        d[k].grad = None
    with torch.no_grad(), F.accumulate_param_grads():
        for _ in range(2):
            assert F._rgb_wgrad(x_nhwc, gz_dev, True, d["w0"], d["b0"], d["wr"], d["br"], s0, sr, (True, True, True, True)) == (None,) * 4
    z, _ = chain64(img, p, bias=True)
    want = dict(zip(names, torch.autograd.grad((z * gz_r).sum(), [p[k] for k in names])))
    for k in names:
        assert rel_err(d[k].grad, 2 * want[k]) <= 8e-3, k


# ---- the whole discriminator with the composed kernel on / off, against the fp64 oracle ------------------------------------
def build_dis(depth_total=MID_DEPTH):
    from stylegan.pytorch_amd.GAN import Discriminator
    dis = Discriminator(resolution=MID["resolution"], num_channels=3, use_wscale=True, blur_filter=[1, 2, 1], fmap_base=MID["fmap_base"],
                        fmap_max=MID["fmap_max"], structure="linear", act_dtype=torch.bfloat16).to(DEV)
    _, dp = mid_params(torch.float64)
    load_into(dis, dp)
    return dis.train(), dp


def d_loss_grads(dis, real, fake, depth, alpha):
    from stylegan.pytorch_amd import Losses
    for q in dis.parameters():
        q.grad = None
    loss = Losses.LogisticGAN(dis).dis_loss(real, fake, depth, alpha)
    loss.backward()
    rimg = real.detach().requires_grad_(True)
    (g_img,) = torch.autograd.grad(dis(rimg, depth, alpha).sum(), rimg)
    return float(loss), {k: q.grad.detach().clone() for k, q in dis.named_parameters() if q.grad is not None}, g_img


@pytest.mark.parametrize("depth", [5, 4])          # newest block at 128^2 with 16 channels / at 64^2 with 32
def test_discriminator_step_with_the_composed_first_layer(depth, monkeypatch):
    from stylegan.pytorch_amd import functional as F
    B, alpha = 4, 0.6
    R = 4 << depth
    real = gu.seeded((B, 3, R, R), 51); fake = gu.seeded((B, 3, R, R), 52)
    dis, dp = build_dis()
    top = dis.blocks[dis.depth - depth - 1]
    assert top.fused_from_rgb_ok((B, R, R, 3), dis.from_rgb[dis.depth - depth - 1], torch.bfloat16)
    used = []
    orig = F.rgbconv_blur
    monkeypatch.setattr(F, "rgbconv_blur", lambda *a: (used.append(1), orig(*a))[1])
    loss_on, g_on, gi_on = d_loss_grads(dis, real.to(DEV), fake.to(DEV), depth, alpha)
    assert len(used) == 3                                                  # D(real), D(fake), the extra forward of this helper
    monkeypatch.setattr(F, "RGBCONV", False)
    assert not top.fused_from_rgb_ok((B, R, R, 3), dis.from_rgb[dis.depth - depth - 1], torch.bfloat16)
    loss_off, g_off, gi_off = d_loss_grads(dis, real.to(DEV), fake.to(DEV), depth, alpha)
    assert len(used) == 3
    # fp64 oracle
    want_loss = O.logistic_d_loss(dp, real.double(), fake.double(), depth, alpha, MID_DEPTH)
    names = [k for k, v in dp.items() if v.requires_grad]
    grads = dict(zip(names, torch.autograd.grad(want_loss, [dp[k] for k in names], allow_unused=True)))
    rimg = real.double().requires_grad_(True)
    (want_gi,) = torch.autograd.grad(O.discriminator(dp, rimg, depth, alpha, MID_DEPTH).sum(), rimg)
    e_on, e_off = rel_err(gi_on, want_gi), rel_err(gi_off, want_gi)
    print(f"[rgbconv D depth {depth}] loss on {loss_on:.6f} off {loss_off:.6f} fp64 {float(want_loss):.6f}; image gradient rel on {e_on:.2e} off {e_off:.2e}")
    assert abs(loss_on - float(want_loss)) <= 2e-2 * abs(float(want_loss))
    assert e_on <= 1.25 * e_off + 2e-3
    assert sorted(g_on) == sorted(g_off) == sorted(k for k in names if grads[k] is not None)
    rows = []
    gmax = max(float(torch.linalg.vector_norm(grads[k])) for k in g_on)
    for k in sorted(g_on):
        a, b = rel_err(g_on[k], grads[k]), rel_err(g_off[k], grads[k])
        rows.append((a, b, k))
        # no tensor's gradient gets worse than the unfused bf16 path's by more than 1.6x (single realisations of a noisy quantity:
        # at depth 4 the deepest block's bias gradient is 42 % off in the UNFUSED path; the medians below are the sharp statement),
        # plus a floor: a bf16 discriminator
        # gradient is typically 5-7e-2 from fp64 (tests/golden/bf16_gates.json), a small tensor at the head moves by 1-2e-2 with
        # ANY change of the roundings upstream, and a tensor whose gradient is tiny next to the network's largest (cancellation:
        # both paths are tens of per cent off there) is bounded on the network's scale instead
        n = float(torch.linalg.vector_norm(grads[k]))
        assert a * n <= (1.6 * b + 2.5e-2) * n + 2e-3 * gmax, (k, a, b, n, gmax)
    rows.sort(reverse=True)
    print("   worst (on, off): " + ", ".join(f"{k} {a:.1e}/{b:.1e}" for a, b, k in rows[:5]))
    med_on, med_off = float(np.median([r[0] for r in rows])), float(np.median([r[1] for r in rows]))
    print(f"   median gradient rel-L2 vs fp64: on {med_on:.3e}, off {med_off:.3e}")
    # (random images + the R1 term of random weights: tens of per cent from fp64 in EITHER path -- a noisy yardstick, hence the slack)
    assert med_on <= 1.3 * med_off + 1e-3


@pytest.mark.parametrize("rgbres", [False, True])
@pytest.mark.parametrize("depth,B,composed,dev_alpha", [(5, 16, True, False), (5, 16, False, False), (5, 16, True, True)])     # (batches at which the stride-2 layer runs on the second-generation kernel)
def test_fade_in_lerp_in_the_store_of_the_stride2_convolution(depth, B, composed, dev_alpha, rgbres, monkeypatch):
    """functional.ConvDownFadeFn (round 4): alpha * lrelu(conv1_down(.)) + (1 - alpha) * from_rgb(pool(img)) with the lerp in the
    convolution's store and the activation kept only as sign bits -- the SAME roundings as the separate passes (the lerp is applied
    to the bf16-rounded activation; the backward multiplies by the same slope), so scores, the R1 image gradient and every parameter
    gradient agree with the unfused path to the order in which autograd sums contributions (reference models/GAN.py:423-427).
    ``dev_alpha``: [alpha, 1 - alpha] read from device memory by the kernel (what a replayed step graph passes).
    ``rgbres`` (round 5, functional.ConvDownFadeRgbFn): the residual branch itself evaluated in that store from the pooled image
    (sgx_rgb_in's arithmetic, rounded to bf16 like the tensor it replaces: the forward is bit-identical) and its backward -- mask pass,
    from_rgb's weight / bias gradient, image gradient -- as one pass over the incoming gradient (sgx_fade_rgb_bwd)."""
    from stylegan.pytorch_amd import functional as F
    alpha = torch.tensor([0.3, 0.7], dtype=torch.float32, device=DEV) if dev_alpha else 0.3
    R = 4 << depth
    real = gu.seeded((B, 3, R, R), 61); fake = gu.seeded((B, 3, R, R), 62)
    dis, dp = build_dis()
    monkeypatch.setattr(F, "RGBCONV", composed)
    monkeypatch.setattr(F, "FUSE_FADE_RGB", rgbres)
    calls = []
    Fn = F.ConvDownFadeRgbFn if rgbres else F.ConvDownFadeFn
    orig = Fn.forward
    monkeypatch.setattr(Fn, "forward", staticmethod(lambda ctx, *a: (calls.append(1), orig(ctx, *a))[1]))
    loss_on, g_on, gi_on = d_loss_grads(dis, real.to(DEV), fake.to(DEV), depth, alpha)
    assert len(calls) == 3, calls
    monkeypatch.setattr(F, "FUSE_FADE", False)
    loss_off, g_off, gi_off = d_loss_grads(dis, real.to(DEV), fake.to(DEV), depth, alpha)
    assert len(calls) == 3
    assert abs(loss_on - loss_off) <= 1e-6 * abs(loss_off), (loss_on, loss_off)
    # With alpha in device memory the UNFUSED residual branch rounds (1 - alpha) * g to bf16 before from_rgb's backward reads it
    # (ScaleDevFn writes a bf16 tensor); the one-pass backward multiplies in fp32: the image gradient and the residual layer's own
    # gradients then differ by that rounding (2^-9 per element: measured 1.7e-3 rel-L2), everything else stays at summation order.
    loose = 4e-3 if (rgbres and dev_alpha) else 1e-5
    assert rel_err(gi_on, gi_off) <= loose
    assert sorted(g_on) == sorted(g_off)
    res_layer = f"from_rgb.{dis.depth - depth}."
    for k in g_on:
        assert rel_err(g_on[k], g_off[k]) <= (loose if k.startswith(res_layer) else 1e-5), (k, rel_err(g_on[k], g_off[k]))


def test_residual_from_rgb_gradients_with_two_backward_branches(monkeypatch):
    """Round 6: in the default D step (LogisticGAN, auxiliary stream on) D(fake)'s backward runs on the auxiliary stream
    and D(real)'s on the main one, and BOTH reach ``ConvDownFadeRgbFn`` of the newest block, whose one-pass backward accumulates
    from_rgb's weight / bias gradient straight into ``.grad``.  Those accumulations must run on the one parameter-gradient stream like
    every other (``sgx_fade_rgb_bwd_finish`` behind the pass, on ``functional._PARAM_GRAD_STREAM``): compared, over repeated steps, with
    the single-stream run of the same kernels (summation order of the three contributions only) and with the unfused residual branch."""
    import random

    from stylegan.pytorch_amd import functional as F
    from test_gpu_graphs import make
    from gpu_util import mid_noises, pin_noise
    B, depth, alpha = 16, 5, 0.3
    calls = []
    orig = F.ConvDownFadeRgbFn.forward
    monkeypatch.setattr(F.ConvDownFadeRgbFn, "forward", staticmethod(lambda ctx, *a: (calls.append(1), orig(ctx, *a))[1]))

    def grads(aux, side, fused=True, reps=3):
        monkeypatch.setattr(F, "FUSE_FADE_RGB", fused)
        sg = make(False, torch.bfloat16, psi=-1.0)
        sg.aux_stream, sg.param_stream = aux, side
        pin_noise(sg.gen, mid_noises(B))
        res = f"from_rgb.{sg.dis.depth - depth}."
        out = []
        for i in range(reps):
            torch.manual_seed(7); random.seed(7)
            z = gu.seeded((B, 512), 300).to(DEV); real = gu.seeded((B, 3, 128, 128), 301).to(DEV)
            sg._d_grads(z, real, depth, alpha)
            torch.cuda.synchronize()
            out.append({k: p.grad.detach().clone() for k, p in sg.dis.named_parameters() if p.grad is not None and k.startswith(res)})
            assert sorted(out[-1]) == [res + "bias", res + "weight"]
        return out

    single = grads(False, False)
    n0 = len(calls)
    assert n0 == 3 * 2                      # D(real) + D(fake) per step (the R1 pass differentiates D(real)'s graph)
    for aux, side in ((True, True), (True, False)):
        multi = grads(aux, side)
        for g1 in multi:
            for k, v in single[0].items():
                assert rel_err(g1[k], v) <= 1e-5, (aux, side, k, rel_err(g1[k], v))
    unfused = grads(True, True, fused=False, reps=1)[0]
    for k, v in single[0].items():
        assert rel_err(unfused[k], v) <= 1e-5, (k, rel_err(unfused[k], v))


@pytest.mark.parametrize("B,R,C,dev_alpha,with_img", [(2, 512, 32, False, True), (2, 256, 64, True, True), (3, 128, 128, False, False), (1, 64, 32, True, True)])
def test_fade_rgb_backward_kernel_vs_oracle(B, R, C, dev_alpha, with_img):
    """``sgx_fade_rgb_bwd`` DIRECTLY against the oracle (round 5 compared it with the unfused path only): the tail of the newest
    discriminator block, y = alpha * lrelu(z) + (1 - alpha) * from_rgb(pool(img)) (reference models/GAN.py:423-427, from_rgb =
    EqualizedConv2d 1x1 with gain sqrt 2, models/GAN.py:377), differentiated by autograd on the oracle's own layers in fp64, at the
    shapes of depth index 8 / 7 / 6 (R x R x C = 512^2 x 32, 256^2 x 64, 128^2 x 128).  Inputs are bf16-exact, so what is compared is
    the kernel's arithmetic: gy is bf16 (2^-9 per element), the three parameter / image gradients are fp32 sums (fp64 across threads)."""
    import math
    from stylegan.pytorch_amd import native as N
    L = N.lib()
    alpha = 0.3
    z = gu.seeded((B, C, R, R), 71, torch.float64)                          # the pre-activation (only its sign reaches the kernel)
    g = gu.seeded((B, R, R, C), 72).bfloat16()                              # dL/dy, NHWC
    pimg = gu.seeded((B, 3, R, R), 73)                                      # the pooled image
    wr = gu.seeded((C, 3, 1, 1), 74); br = 0.1 * gu.seeded((C,), 75)
    # oracle
    p64 = [t.double().requires_grad_(True) for t in (z, pimg, wr, br)]
    y = alpha * O.leaky_relu(p64[0]) + (1.0 - alpha) * O.eq_conv2d(p64[1], p64[2], p64[3], gain=math.sqrt(2.0))
    y.backward(g.double().permute(0, 3, 1, 2))
    # kernel
    bits = torch.zeros((B, R, R, C // 8), dtype=torch.uint8)
    zb = (z.permute(0, 2, 3, 1) > 0).reshape(B, R, R, C // 8, 8).to(torch.uint8)
    for j in range(8):
        bits |= zb[..., j] << j
    ws, bs = math.sqrt(2.0) / math.sqrt(3.0), 1.0                           # from_rgb's w_mul (gain sqrt 2, fan-in 3) and b_mul
    gd, bitsd = g.to(DEV), bits.to(DEV)
    pimgd = pimg.permute(0, 2, 3, 1).contiguous().to(DEV); wrd = wr.to(DEV)
    gy = torch.empty_like(gd)
    dw = torch.empty((C, 3), dtype=torch.float32, device=DEV); db = torch.empty((C,), dtype=torch.float32, device=DEV)
    gpimg = torch.empty_like(pimgd) if with_img else None
    npix = B * R * R
    wsb = L.sgx_fade_rgb_bwd_ws_bytes(npix, C)
    wsp = N.workspace(wsb, gd.device)
    ab = torch.tensor([alpha, 1.0 - alpha], dtype=torch.float32, device=DEV) if dev_alpha else None
    N.check(L.sgx_fade_rgb_bwd(N.ptr(gd), N.ptr(bitsd), N.ptr(pimgd), N.ptr(wrd), ws, bs, 0.0 if dev_alpha else alpha, 0.0 if dev_alpha else 1.0 - alpha,
                               N.ptr(ab), N.ptr(gy), N.ptr(dw), N.ptr(db), 0, N.ptr(gpimg), N.ptr(wsp), wsb, npix, C, N.BF16, N.stream()), "sgx_fade_rgb_bwd")
    torch.cuda.synchronize()
    assert rel_err(gy.float().permute(0, 3, 1, 2), p64[0].grad) <= 3e-3                      # bf16 output
    assert rel_err(dw.view(C, 3, 1, 1), p64[2].grad) <= 1e-4, rel_err(dw.view(C, 3, 1, 1), p64[2].grad)
    assert rel_err(db, p64[3].grad) <= 1e-4
    if with_img:
        assert rel_err(gpimg.permute(0, 3, 1, 2), p64[1].grad) <= 1e-5
    # the two-call form (the pass leaves the block partials, sgx_fade_rgb_bwd_finish accumulates on any stream ordered behind it)
    dw2 = torch.full_like(dw, 2.0); db2 = torch.full_like(db, -1.0)
    gy2 = torch.empty_like(gd)
    N.check(L.sgx_fade_rgb_bwd(N.ptr(gd), N.ptr(bitsd), N.ptr(pimgd), N.ptr(wrd), ws, bs, 0.0 if dev_alpha else alpha, 0.0 if dev_alpha else 1.0 - alpha,
                               N.ptr(ab), N.ptr(gy2), None, None, 0, None, N.ptr(wsp), wsb, npix, C, N.BF16, N.stream()), "sgx_fade_rgb_bwd")
    N.check(L.sgx_fade_rgb_bwd_finish(N.ptr(wsp), wsb, npix, C, ws, bs, 0.0 if dev_alpha else 1.0 - alpha, N.ptr(ab), N.ptr(dw2), N.ptr(db2), 3,
                                      N.stream()), "sgx_fade_rgb_bwd_finish")
    torch.cuda.synchronize()
    assert torch.equal(gy2, gy)
    assert torch.equal(dw2, dw + 2.0) and torch.equal(db2, db - 1.0)                         # accumulate bits 0 and 1


@pytest.mark.parametrize("C,dev_alpha", [(32, False), (64, True), (128, False)])
def test_adjoint_of_the_fade_backward_is_bit_identical_to_the_three_passes(C, dev_alpha):
    """``sgx_fade_rgb_bwd2`` (round 6: what the R1 double backward needs of the newest block's tail, one pass) keeps the roundings of the three
    passes it replaces -- ``LReluBwdBitsFn`` on the gradient's gradient, ``RgbInFn`` (from_rgb) of the image gradient's gradient [and the
    device-beta scaling pass], autograd's bf16 add -- so its output is theirs BIT FOR BIT; each operand alone too."""
    from stylegan.pytorch_amd import functional as F
    from stylegan.pytorch_amd import native as N
    L = N.lib()
    B, R, alpha, ws = 2, 48, 0.3, 0.41
    ggy = gu.seeded((B, R, R, C), 101).bfloat16().to(DEV)
    ggp = gu.seeded((B, R, R, 3), 102).to(DEV)
    bits = (gu.seeded((B, R, R, C // 8), 103) * 1000).abs().to(torch.int64).remainder(256).to(torch.uint8).to(DEV)
    wr = gu.seeded((C, 3, 1, 1), 104).to(DEV)
    ab = torch.tensor([alpha, 1.0 - alpha], dtype=torch.float32, device=DEV) if dev_alpha else None
    with torch.no_grad():
        t1 = F.LReluBwdBitsFn.apply(ggy, bits, 0.2, ab[0:1] if dev_alpha else alpha)
        t2 = F.RgbInFn.apply(ggp, wr, None, ws, torch.bfloat16)
        if dev_alpha:
            t2 = F.ScaleDevFn.apply(t2, ab[1:2])
        want = t1 + t2
    npix = B * R * R

    def run(a, b):
        out = torch.empty_like(ggy)
        N.check(L.sgx_fade_rgb_bwd2(N.ptr(a), N.ptr(b), N.ptr(bits), N.ptr(wr), ws, 0.0 if dev_alpha else alpha, 1.0, N.ptr(ab), N.ptr(out), npix, C,
                                    N.BF16, N.stream()), "sgx_fade_rgb_bwd2")
        torch.cuda.synchronize()
        return out
    assert torch.equal(run(ggy, ggp).view(torch.int16), want.view(torch.int16))
    assert torch.equal(run(ggy, None).view(torch.int16), t1.view(torch.int16))
    assert torch.equal(run(None, ggp).view(torch.int16), t2.view(torch.int16))
