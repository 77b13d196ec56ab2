"""-m gpu: the pass fusions of round 3, each against (a) the unfused path of the library (fp32: tight, the algebra is
exact up to summation order; bf16: identical stored bits where the fused kernel performs the same roundings) and (b) the
CPU oracle in fp64.  Reference compositions: LayerEpilogue models/CustomLayers.py:219-248, BlurLayer :251-276,
GSynthesisBlock / DiscriminatorBlock models/Blocks.py:63-88,137-146, fade-in / to_rgb / from_rgb models/GAN.py:199-202,
425-427."""
import pytest
import torch
import torch.nn.functional as TF

import golden_util as gu
from gpu_util import DEV, assert_close, rel_err
from oracle import stylegan_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _lib():
    from stylegan.pytorch_amd import native
    assert torch.cuda.is_available()
    native.lib()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H", [(2, 16, 64), (3, 32, 16), (1, 64, 8), (2, 512, 4), (1, 16, 256), (2, 128, 12)])
def test_blur_with_epilogue_statistics(B, C, H, dt):
    """sgx_blur3x3_stats: y bit-identical to the plain blur; the partial sums reproduce the statistics of
    lrelu(y + bias + nw*noise) computed in fp64 from the stored y."""
    from stylegan.pytorch_amd import functional as F
    x = gu.seeded((B, H, H, C), 70).to(DEV).to(dt)
    bias = (0.1 * gu.seeded((C,), 71)).to(DEV)
    nw = (0.3 * gu.seeded((C,), 72)).to(DEV)
    noise = gu.seeded((B, 1, H, H), 73).to(DEV)
    y, part = F.BlurStatsFn.apply(x, bias, noise, nw)
    y_ref = F.BlurFn.apply(x)
    assert torch.equal(y, y_ref)
    a = TF.leaky_relu(y.double() + bias.double() + nw.double() * noise.double().reshape(B, H, H, 1), 0.2)
    s = part.sum(dim=1)                                              # [B, C, 2]
    assert_close(s[..., 0], a.sum(dim=(1, 2)), 1e-6, "sum a", floor=1e-6 * H * H)
    assert_close(s[..., 1], (a * a).sum(dim=(1, 2)), 1e-6, "sum a^2")


# (C, B, H): the six instantiations of the statistics epilogue -- 16-channel 8 / 4 waves, 32-channel blocks 8 / 4, 64-channel 8 / 4 --
# plus ragged rows and several channel blocks
CONV_STATS_CASES = [(16, 2, 512), (16, 1, 256), (32, 2, 256), (32, 1, 256), (64, 4, 256), (64, 1, 256), (128, 3, 136), (16, 3, 520)]


@pytest.mark.parametrize("C,B,H", CONV_STATS_CASES)
def test_conv3x3_with_epilogue_statistics(C, B, H):
    """sgx_conv3x3_stats: y bit-identical to the plain convolution; the per-tile partials add up to the statistics of
    lrelu(y + bias + nw*noise) computed in fp64 from the stored y."""
    from stylegan.pytorch_amd import functional as F
    W = 256 if H != 520 else 512
    x = gu.seeded((B, H, W, C), 90).to(DEV).bfloat16()
    w = gu.seeded((C, C, 3, 3), 91).to(DEV)
    bias = (0.1 * gu.seeded((C,), 92)).to(DEV)
    nw = (0.3 * gu.seeded((C,), 93)).to(DEV)
    noise = gu.seeded((B, 1, H, W), 94).to(DEV)
    scale = O.he_w_mul(C * 9, 2 ** 0.5)
    assert F.conv_stats_nparts(x, C) > 0, "shape expected to have a fused kernel"
    y, part = F.ConvFn.apply(x, w, None, "S", scale, C, False, 0, None, False, False, (bias, noise, nw))
    y_ref = F.ConvFn.apply(x, w, None, "S", scale, C, False, 0)
    assert torch.equal(y, y_ref)
    a = TF.leaky_relu(y.double() + bias.double() + nw.double() * noise.double().reshape(B, H, W, 1), 0.2)
    s = part.sum(dim=1)                                              # [B, C, 2]
    assert_close(s[..., 0], a.sum(dim=(1, 2)), 2e-6, "sum a", floor=2e-6 * H * W)
    assert_close(s[..., 1], (a * a).sum(dim=(1, 2)), 2e-6, "sum a^2")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,B,H", [(32, 16, 2, 16), (64, 64, 1, 64), (32, 32, 3, 8), (32, 16, 2, 256), (64, 32, 1, 128), (128, 64, 4, 128)])
def test_generator_block_with_fused_statistics(cin, cout, B, H, dt):
    """GSynthesisBlock with the instance-norm statistics produced by the blur / the 3x3 convolution (SGX_FUSE_EPI_STATS)
    against the same block with the separate statistics passes, forward and every gradient; fp32 also against the oracle."""
    from stylegan.pytorch_amd import Blocks
    from stylegan.pytorch_amd import functional as F
    blk = Blocks.GSynthesisBlock(cin, cout, [1, 2, 1], 512, 2 ** 0.5, True, True, False, True, True, torch.nn.LeakyReLU(0.2)).to(DEV)
    names = dict(blk.named_parameters())
    with torch.no_grad():
        for k, p in names.items():
            p.copy_(gu.fill_value("blk." + k, p.shape))
    n1 = gu.seeded((B, 1, 2 * H, 2 * H), 80).to(DEV); n2 = gu.seeded((B, 1, 2 * H, 2 * H), 81).to(DEV)
    blk.epi1.top_epi.noise.noise = n1; blk.epi2.top_epi.noise.noise = n2
    x = gu.seeded((B, H, H, cin), 82); dl = gu.seeded((B, 2, 512), 83); gy = gu.seeded((B, 2 * H, 2 * H, cout), 84)

    def run(fuse):
        keep, Blocks.FUSE_EPI_STATS = Blocks.FUSE_EPI_STATS, fuse
        keep_min, Blocks.FUSE_EPI_STATS_MIN = Blocks.FUSE_EPI_STATS_MIN, 0
        # (the statistics fusion is what is compared here: conv0_up -> blur as the same separate kernels on both sides -- the
        # round-5 composite kernel of the 32 -> 16 layer rounds differently, test_conv_upblur_composite_kernel_vs_oracle)
        keep_ub, F.CONV_UPBLUR = F.CONV_UPBLUR, False
        try:
            for p in names.values():
                p.grad = None
            xg = x.to(DEV).to(dt).requires_grad_(True); dg = dl.to(DEV).requires_grad_(True)
            y = blk.forward_nhwc(xg, dg)
            y.backward(gy.to(DEV).to(dt))
            return y.detach(), xg.grad, dg.grad, {k: p.grad.clone() for k, p in names.items()}
        finally:
            Blocks.FUSE_EPI_STATS, Blocks.FUSE_EPI_STATS_MIN = keep, keep_min
            F.CONV_UPBLUR = keep_ub
    y0, gx0, gd0, gp0 = run(0)
    y1, gx1, gd1, gp1 = run(3)
    tol = 2e-6 if dt == torch.float32 else 4e-3                      # bf16: a statistic moving by 1e-7 flips roundings of y
    assert_close(y1, y0, tol, "y fused vs unfused")
    assert_close(gx1, gx0, 10 * tol, "dx")
    assert_close(gd1, gd0, 10 * tol, "d dlatents")
    for k in gp0:
        assert_close(gp1[k], gp0[k], 10 * tol, k, floor=1e-7)
    if dt == torch.float32:                                          # and against the oracle
        p64 = {k: v.detach().double().cpu() for k, v in names.items()}
        r = O.eq_conv2d(x.permute(0, 3, 1, 2).double(), p64["conv0_up.weight"], p64["conv0_up.bias"], up=True, blur_after=True)
        r = O._epi(p64, "epi1.", r, n1.double().cpu(), dl[:, 0].double())
        r = O.eq_conv2d(r, p64["conv1.weight"], p64["conv1.bias"])
        r = O._epi(p64, "epi2.", r, n2.double().cpu(), dl[:, 1].double())
        assert_close(F.nchw_view(y1), r, 2e-5, "y vs oracle")


# transposed convolution + blur (+ mask) in one kernel: 8- and 4-wave blocks, 32- and 16-channel output blocks, several input
# chunks, ragged heights (tiles overlap by one coarse row / column: every seam position and the image borders are exercised)
UPBLUR_CASES = [
    # (cin, cout, B, H, W)
    (64, 32, 2, 256, 256), (64, 32, 1, 256, 256), (32, 16, 4, 256, 256), (32, 16, 2, 256, 256), (128, 64, 2, 128, 128),
    (64, 32, 8, 40, 256), (32, 32, 48, 17, 64), (64, 64, 3, 250, 96), (32, 16, 3, 120, 512), (96, 32, 64, 16, 64), (32, 32, 64, 15, 64),
]


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("cin,cout,B,H,W", UPBLUR_CASES)
def test_conv_up_blur_fused_vs_separate_and_oracle(cin, cout, B, H, W, masked):
    """sgx_conv4x4s2_up_blur against conv -> blur (-> mask) run as separate kernels, and both against fp64 on the bf16-rounded
    operands: both round twice to bf16 (separate: the convolution's output and the blur's; fused: the vertically blurred rows in
    the store scratch and the result), measured 2.1e-3 vs 1.9e-3 rel-L2 from fp64."""
    from stylegan.pytorch_amd import functional as F
    w = gu.seeded((cout, cin, 3, 3), 5).to(DEV)
    scale = O.he_w_mul(cin * 9, 2 ** 0.5)
    x = gu.seeded((B, H, W, cin), 7).to(DEV).bfloat16()
    z = gu.seeded((B, 2 * H, 2 * W, cout), 8).to(DEV).bfloat16() if masked else None
    keep, F.CONV_BLUR_POLICY = F.CONV_BLUR_POLICY, "all"
    try:
        assert F.conv_blur_ok(x, cout, "U", False), "shape expected to have the fused kernel"
    finally:
        F.CONV_BLUR_POLICY = keep
    with torch.no_grad():
        y1 = F.ConvBlurFn.apply(x, w, "U", scale, cin, False, z)
        t = F.ConvFn.apply(x, w, None, "U", scale, cin, False, 0)
        y0 = F.BlurMaskFn.apply(t, z) if masked else F.BlurFn.apply(t)
        wq, _ = F.packs(w, "U", scale, cin, torch.bfloat16)
    wr = wq.float().view(4, 4, cout, cin).permute(2, 3, 0, 1).double().cpu()
    ref = TF.conv_transpose2d(x.float().permute(0, 3, 1, 2).double().cpu(), wr.permute(1, 0, 2, 3), stride=2, padding=1)
    k = torch.tensor([1.0, 2.0, 1.0], dtype=torch.float64); k = (k[:, None] * k[None, :] / 16.0).expand(cout, 1, 3, 3)
    ref = TF.conv2d(ref, k, padding=1, groups=cout)
    if masked:
        ref = ref * torch.where(z.float().permute(0, 3, 1, 2).double().cpu() > 0, 1.0, 0.2)
    assert torch.isfinite(y1.float()).all()
    e1, e0 = rel_err(F.nchw_view(y1), ref), rel_err(F.nchw_view(y0), ref)
    # 32 -> 16 channels without a mask tensor run the round-5 composite kernel: ONE rounding of the composed weights (a sum of up to
    # nine taps) instead of nine independently rounded taps whose errors average: 2.5e-3 against 1.9e-3 for the separate passes,
    # measured -- under the absolute bar, 1.3x the separate path
    slack = 1.4 if (cin, cout, masked) == (32, 16, False) else 1.25
    assert e1 <= 3e-3 and e1 <= slack * e0 + 1e-4, (e1, e0)
    assert_close(y1, y0, 6e-3, "fused vs separate passes")
    # every position, borders and tile seams included: the worst element is a rounding error, as in the separate passes (a wrong
    # seam row / column or border would be off by the size of the values)
    d1 = (F.nchw_view(y1).double().cpu() - ref).abs().max().item()
    d0 = (F.nchw_view(y0).double().cpu() - ref).abs().max().item()
    assert d1 <= 2.0 * d0 + 1e-3 and d1 < 0.05 * ref.abs().max().item(), (d1, d0, ref.abs().max().item())


# Round 5: blur o transposed convolution as ONE 3x3 convolution to the four output parity classes (sgx_conv_upblur, composite weights
# from sgx_pack_upblur, depth-to-space store, border correction taps, mask from sign bits).  Every border case: one-row and two-row
# images (first AND last fine row in one wave), one tile column (first and last column in one tile), ragged rows, several tile
# columns and rows, the 4- and the 8-wave block; with and without the mask.
UPBLUR3_CASES = [(2, 1, 32), (1, 2, 32), (3, 16, 32), (2, 17, 64), (1, 40, 96), (2, 33, 128), (5, 64, 64), (2, 256, 256), (1, 130, 512)]


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("adjoint", [False, True])
@pytest.mark.parametrize("B,H,W", UPBLUR3_CASES)
def test_conv_upblur_composite_kernel_vs_oracle(B, H, W, adjoint, masked):
    """fp64 reference: conv_transpose2d of the bf16-rounded input with the layer's 16 fp32 taps, then the zero-padded [1,2,1]^2/16
    blur, then the LeakyReLU-backward slope of the mask.  The kernel rounds the COMPOSITE weights and the result to bf16 once each
    (the separate passes: the taps, the convolution's output and the blur's output).  ``adjoint``: the discriminator's use (the data
    gradient of a 16 -> 32 stride-2 layer) instead of the generator's (a 32 -> 16 up layer)."""
    import numpy as np
    from stylegan.pytorch_amd import functional as F
    cin, cout = 32, 16
    w = gu.seeded((cin, cout, 3, 3) if adjoint else (cout, cin, 3, 3), 5).to(DEV)       # [O][I][3][3] of the layer the weight belongs to
    mode = "D" if adjoint else "U"
    scale = O.he_w_mul(w.shape[1] * 9, 2 ** 0.5)
    x = gu.seeded((B, H, W, cin), 7).to(DEV).bfloat16()
    z = gu.seeded((B, 2 * H, 2 * W, cout), 8).to(DEV).bfloat16()
    bits = None
    if masked:
        zb = (z > 0).cpu().numpy().reshape(B, 2 * H, 2 * W, cout // 8, 8)
        bits = torch.from_numpy(np.packbits(zb, axis=-1, bitorder="little").reshape(B, 2 * H, 2 * W, cout // 8)).to(DEV)
    assert F.conv_upblur_ok(x, cout, mode, adjoint)
    with torch.no_grad():
        y = F.ConvBlurFn.apply(x, w, mode, scale, int(w.shape[1]), adjoint, None, bits)
        f32, a32 = F.packs(w, mode, scale, int(w.shape[1]), torch.float32)
    t4 = (a32 if adjoint else f32).double().cpu()                                         # [ky*4+kx][N = cout][K = cin]
    wr = t4.view(4, 4, cout, cin).permute(3, 2, 0, 1)                                      # conv_transpose2d weight [Cin][Cout][4][4]
    ref = TF.conv_transpose2d(x.float().permute(0, 3, 1, 2).double().cpu(), wr, stride=2, padding=1)
    k = torch.tensor([1.0, 2.0, 1.0], dtype=torch.float64); k = (k[:, None] * k[None, :] / 16.0).expand(cout, 1, 3, 3)
    ref = TF.conv2d(ref, k, padding=1, groups=cout)
    if masked:
        ref = ref * torch.where(z.float().permute(0, 3, 1, 2).double().cpu() > 0, 1.0, 0.2)
    got = F.nchw_view(y).double().cpu()
    assert y.shape == (B, 2 * H, 2 * W, cout) and torch.isfinite(got).all()
    e = rel_err(got, ref)
    # the border ring separately (first / last two fine rows and columns): a wrong correction tap is an O(1) relative error THERE and
    # invisible in the whole-tensor norm of a large image
    ring = torch.zeros_like(ref, dtype=torch.bool)
    ring[..., :2, :] = True; ring[..., -2:, :] = True; ring[..., :, :2] = True; ring[..., :, -2:] = True
    e_ring = float((got[ring] - ref[ring]).norm() / ref[ring].norm())
    d = float((got - ref).abs().max()); scale_v = float(ref.abs().max())
    print(f"[upblur3 B{B} {H}x{W} adjoint={adjoint} masked={masked}] rel {e:.2e}, border ring rel {e_ring:.2e}, max |d| {d:.2e} of {scale_v:.2e}")
    assert e <= 3e-3 and e_ring <= 4e-3 and d <= 0.02 * scale_v, (e, e_ring, d, scale_v)


@pytest.mark.parametrize("cin,cout,B,H,W", [(64, 32, 8, 40, 256), (32, 32, 48, 17, 64), (128, 64, 2, 128, 128)])
def test_conv_up_blur_with_the_mask_as_sign_bits(cin, cout, B, H, W):
    """sgx_conv4x4s2_up_blur_bits (round 4): the activation mask read as one sign bit per element -- the same bits as the mask-tensor
    variant of the kernel produces, for every position (tile seams, borders, ragged tiles).  (32 -> 16 channels take the round-5
    kernel when the mask comes as bits: test_conv_upblur_composite_kernel_vs_oracle.)"""
    from stylegan.pytorch_amd import functional as F
    w = gu.seeded((cout, cin, 3, 3), 5).to(DEV)
    scale = O.he_w_mul(cin * 9, 2 ** 0.5)
    x = gu.seeded((B, H, W, cin), 7).to(DEV).bfloat16()
    z = gu.seeded((B, 2 * H, 2 * W, cout), 8).to(DEV).bfloat16()
    zb = (z > 0).cpu().numpy().reshape(B, 2 * H, 2 * W, cout // 8, 8)
    import numpy as np
    bits = torch.from_numpy(np.packbits(zb, axis=-1, bitorder="little").reshape(B, 2 * H, 2 * W, cout // 8)).to(DEV)
    with torch.no_grad():
        y_z = F.ConvBlurFn.apply(x, w, "U", scale, cin, False, z)
        y_b = F.ConvBlurFn.apply(x, w, "U", scale, cin, False, None, bits)
    assert torch.equal(y_z, y_b)


@pytest.mark.parametrize("cin,cout,B,H", [(32, 64, 2, 256), (16, 32, 4, 512), (64, 128, 2, 128)])
def test_discriminator_block_backward_with_fused_blur(cin, cout, B, H):
    """DiscriminatorBlock in bf16: first-order gradients and the R1-style double backward with the blur + mask folded into
    conv1_down's data-gradient kernel, against the separate passes (same kernels otherwise)."""
    from stylegan.pytorch_amd import Blocks
    from stylegan.pytorch_amd import functional as F
    blk = Blocks.DiscriminatorBlock(cin, cout, 2 ** 0.5, True, torch.nn.LeakyReLU(0.2), [1, 2, 1]).to(DEV)
    names = dict(blk.named_parameters())
    with torch.no_grad():
        for k, p in names.items():
            p.copy_(gu.fill_value("blk." + k, p.shape))
    x = gu.seeded((B, H, H, cin), 60); gy = gu.seeded((B, H // 2, H // 2, cout), 61)

    def run(on):
        keep, F.CONV_BLUR_POLICY = F.CONV_BLUR_POLICY, "all" if on else "off"
        try:
            for p in names.values():
                p.grad = None
            xg = x.to(DEV).bfloat16().requires_grad_(True)
            y = blk.forward_nhwc(xg)
            (g1,) = torch.autograd.grad((y.float() * gy.to(DEV)).sum(), xg, create_graph=True)
            pen = (g1.float() ** 2).sum()
            pen.backward()                                            # second order: d pen / d params, d pen / d x
            # (a parameter the penalty does not depend on -- a bias acts through the LeakyReLU masks only -- has no gradient: zero)
            return y.detach(), g1.detach(), xg.grad, {k: (torch.zeros_like(p) if p.grad is None else p.grad.clone()) for k, p in names.items()}
        finally:
            F.CONV_BLUR_POLICY = keep
    y0, g0, gx0, gp0 = run(False)
    y1, g1, gx1, gp1 = run(True)
    assert torch.equal(y0, y1)
    assert_close(g1, g0, 6e-3, "first-order data gradient")
    assert_close(gx1, gx0, 2e-2, "second-order data gradient")
    for k in gp0:
        assert_close(gp1[k], gp0[k], 2e-2, "second-order " + k)


@pytest.mark.parametrize("cin,cout,B,H", [(64, 32, 2, 128), (32, 16, 2, 256), (128, 64, 4, 64)])
def test_generator_block_with_fused_up_blur(cin, cout, B, H):
    """GSynthesisBlock in bf16 with conv0_up -> blur as one kernel against the separate passes: output and every gradient."""
    from stylegan.pytorch_amd import Blocks
    from stylegan.pytorch_amd import functional as F
    blk = Blocks.GSynthesisBlock(cin, cout, [1, 2, 1], 512, 2 ** 0.5, True, True, False, True, True, torch.nn.LeakyReLU(0.2)).to(DEV)
    names = dict(blk.named_parameters())
    with torch.no_grad():
        for k, p in names.items():
            p.copy_(gu.fill_value("blk." + k, p.shape))
    blk.epi1.top_epi.noise.noise = gu.seeded((B, 1, 2 * H, 2 * H), 80).to(DEV)
    blk.epi2.top_epi.noise.noise = gu.seeded((B, 1, 2 * H, 2 * H), 81).to(DEV)
    x = gu.seeded((B, H, H, cin), 82); dl = gu.seeded((B, 2, 512), 83); gy = gu.seeded((B, 2 * H, 2 * H, cout), 84)

    def run(on):
        keep, F.CONV_BLUR_POLICY = F.CONV_BLUR_POLICY, "all" if on else "off"
        try:
            for p in names.values():
                p.grad = None
            xg = x.to(DEV).bfloat16().requires_grad_(True); dg = dl.to(DEV).requires_grad_(True)
            y = blk.forward_nhwc(xg, dg)
            y.backward(gy.to(DEV).bfloat16())
            return y.detach(), xg.grad, dg.grad, {k: p.grad.clone() for k, p in names.items()}
        finally:
            F.CONV_BLUR_POLICY = keep
    y0, gx0, gd0, gp0 = run(False)
    y1, gx1, gd1, gp1 = run(True)
    # the two forwards differ by bf16 rounding noise (2e-3 per element, kernel test above); the gradients see it through two
    # instance norms and every LeakyReLU whose argument it flips: measured up to 3.8e-2 on dx / 6.5e-2 on a bias with 16 channels
    assert_close(y1, y0, 1e-2, "y fused vs separate")
    assert_close(gx1, gx0, 8e-2, "dx")
    assert_close(gd1, gd0, 8e-2, "d dlatents")
    gmax = max(float(v.norm()) for v in gp0.values())
    for k in gp0:
        # (a convolution bias in front of an instance norm has a nearly cancelling gradient: noise relative to itself, judged
        # against the block's largest gradient tensor instead)
        assert_close(gp1[k], gp0[k], 8e-2, k, floor=(2e-2 * gmax if k.endswith(".bias") else 1e-6))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,B,H", [(16, 2, 64), (32, 3, 16), (128, 1, 8), (16, 4, 256)])
@pytest.mark.parametrize("dev_alpha", [False, True])
def test_generator_output_in_one_pass(C, B, H, dt, dev_alpha):
    """to_rgb + nearest upsample of the previous resolution's RGB + fade-in lerp (models/GAN.py:199-202) as one kernel against
    the three separate ops: image and every gradient; fp32 also against the formula in fp64."""
    from stylegan.pytorch_amd import functional as F
    w = (0.2 * gu.seeded((3, C, 1, 1), 40)).to(DEV).requires_grad_(True)
    bias = (0.1 * gu.seeded((3,), 41)).to(DEV).requires_grad_(True)
    x = gu.seeded((B, H, H, C), 42); low = gu.seeded((B, H // 2, H // 2, 3), 43); g = gu.seeded((B, H, H, 3), 44)
    alpha = 0.3
    al = torch.tensor([alpha, 1 - alpha], dtype=torch.float32, device=DEV) if dev_alpha else alpha

    def run(fused):
        w.grad = bias.grad = None
        xg = x.to(DEV).to(dt).requires_grad_(True); lg = low.to(DEV).requires_grad_(True)
        if fused:
            img = F.RgbOutFadeFn.apply(xg, w, bias, 0.25, lg, al)
        else:
            img = F.fade(F.RgbOutFn.apply(xg, w, bias, 0.25), F.Up2Fn.apply(lg, 1.0), al)
        img.backward(g.to(DEV))
        return img.detach(), xg.grad, lg.grad, w.grad.clone(), bias.grad.clone()
    a, b = run(False), run(True)
    tol = 2e-6 if dt == torch.float32 else 4e-3
    for u, v, what in zip(b, a, ("image", "dx", "d low", "d weight", "d bias")):
        assert_close(u, v, tol, what)
    if dt == torch.float32:
        ref = alpha * (torch.einsum("bhwc,jc->bhwj", x.double(), w.detach().double().cpu().view(3, C)) * 0.25 + bias.detach().double().cpu()) \
            + (1 - alpha) * low.double().repeat_interleave(2, 1).repeat_interleave(2, 2)
        assert_close(b[0], ref, 2e-6, "image vs fp64")


@pytest.mark.parametrize("dev_alpha", [False, True])
@pytest.mark.parametrize("B,H", [(2, 8), (3, 64), (4, 1024)])
def test_real_batch_downsample_fade_in_one_pass(B, H, dev_alpha):
    """alpha * x + (1 - alpha) * up2(avgpool2(x)) in one kernel: bit-identical to pool -> upsample -> lerp."""
    from stylegan.pytorch_amd import functional as F
    x = gu.seeded((B, H, H, 3), 50).to(DEV)
    alpha = 0.4
    al = torch.tensor([alpha, 1 - alpha], dtype=torch.float32, device=DEV) if dev_alpha else alpha
    fused = F.downsample_fade_rgb(x, al)
    ref = F.fade(x, F.Up2Fn.apply(F.Pool2Fn.apply(x, 0.25), 1.0), al)
    assert_close(fused, ref, 1e-7, "fused vs separate")
    r64 = alpha * x.double() + (1 - alpha) * torch.nn.functional.avg_pool2d(x.double().permute(0, 3, 1, 2), 2).repeat_interleave(2, 2).repeat_interleave(2, 3).permute(0, 2, 3, 1)
    assert_close(fused, r64, 1e-6, "vs fp64")


@pytest.mark.parametrize("C,B,H", [(16, 2, 512), (16, 1, 256), (32, 2, 256), (64, 4, 256), (64, 1, 256), (128, 3, 136)])
def test_sign_bits_of_the_block_preactivation(C, B, H):
    """The discriminator block's conv0 writes one sign bit per element next to z; the backward passes (blur * slope, blur of
    slope * g) read the bits instead of z: bit-identical results."""
    from stylegan.pytorch_amd import functional as F
    W = 256
    x = gu.seeded((B, H, W, C), 90).to(DEV).bfloat16()
    w = gu.seeded((C, C, 3, 3), 91).to(DEV); bias = (0.3 * gu.seeded((C,), 92)).to(DEV)
    scale = O.he_w_mul(C * 9, 2 ** 0.5)
    assert F.conv_signbits_ok(x, C)
    with torch.no_grad():
        z, bits = F.ConvFn.apply(x, w, bias, "S", scale, C, False, 0, None, False, False, None, None, True)
        z_ref = F.ConvFn.apply(x, w, bias, "S", scale, C, False, 0)
        assert torch.equal(z, z_ref)
        want = (z.float() > 0).reshape(B, H, W, C // 8, 8)
        got = ((bits.unsqueeze(-1).int() >> torch.arange(8, device=DEV)) & 1).bool()
        assert torch.equal(got, want)
        g = gu.seeded((B, H, W, C), 93).to(DEV).bfloat16()
        assert torch.equal(F.BlurMaskFn.apply(g, z, bits), F.BlurMaskFn.apply(g, z))
        assert torch.equal(F.MaskBlurFn.apply(g, z, bits), F.MaskBlurFn.apply(g, z))


@pytest.mark.parametrize("cin,cout,B,H", [(32, 64, 2, 256), (16, 32, 2, 512)])
def test_discriminator_block_with_sign_bits(cin, cout, B, H):
    """DiscriminatorBlock in bf16, first and second order, with the mask read as sign bits against the mask read from z."""
    from stylegan.pytorch_amd import Blocks
    from stylegan.pytorch_amd import functional as F
    blk = Blocks.DiscriminatorBlock(cin, cout, 2 ** 0.5, True, torch.nn.LeakyReLU(0.2), [1, 2, 1]).to(DEV)
    names = dict(blk.named_parameters())
    with torch.no_grad():
        for k, p in names.items():
            p.copy_(gu.fill_value("blk." + k, p.shape))
    x = gu.seeded((B, H, H, cin), 60); gy = gu.seeded((B, H // 2, H // 2, cout), 61)

    def run(on):
        keep, F.SIGNBITS_ON = F.SIGNBITS_ON, on
        keep_p, F.CONV_BLUR_POLICY = F.CONV_BLUR_POLICY, "off"
        try:
            for p in names.values():
                p.grad = None
            xg = x.to(DEV).bfloat16().requires_grad_(True)
            y = blk.forward_nhwc(xg)
            (g1,) = torch.autograd.grad((y.float() * gy.to(DEV)).sum(), xg, create_graph=True)
            (g1.float() ** 2).sum().backward()
            # (a parameter the penalty does not depend on -- a bias acts through the LeakyReLU masks only -- has no gradient: zero)
            return y.detach(), g1.detach(), xg.grad, {k: (torch.zeros_like(p) if p.grad is None else p.grad.clone()) for k, p in names.items()}
        finally:
            F.SIGNBITS_ON, F.CONV_BLUR_POLICY = keep, keep_p
    y0, g0, gx0, gp0 = run(False)
    y1, g1, gx1, gp1 = run(True)
    assert torch.equal(y0, y1) and torch.equal(g0, g1) and torch.equal(gx0, gx1)      # the same arithmetic, the mask from another place
    for k in gp0:
        assert_close(gp1[k], gp0[k], 1e-6, k)


@pytest.mark.parametrize("B", [1, 4, 32, 100])
@pytest.mark.parametrize("scale", [1.0, 0.125])
def test_logistic_loss_heads_in_one_launch(B, scale):
    """sgx_logistic_loss (reference models/Losses.py:213-229): loss and logit gradients of both heads against torch's own softplus /
    mean in fp64, including logits beyond softplus' linear threshold; the upstream gradient arrives as a device scalar."""
    from stylegan.pytorch_amd import functional as F
    f = (6.0 * gu.seeded((B, 1), 90)).to(DEV)
    r = (6.0 * gu.seeded((B, 1), 91)).to(DEV)
    f[0, 0] = 31.0
    r[0, 0] = -27.5
    up = torch.tensor(1.7, device=DEV)
    for gen in (False, True):
        fa, ra = f.clone().requires_grad_(True), r.clone().requires_grad_(True)
        loss = F.LogisticLossFn.apply(fa, None if gen else ra, scale, gen)
        (loss * up).backward()
        fd, rd = f.double().requires_grad_(True), r.double().requires_grad_(True)
        ref = (TF.softplus(-fd).mean() if gen else TF.softplus(fd).mean() + TF.softplus(-rd).mean()) * scale
        (ref * up.double()).backward()
        assert_close(loss, ref, 1e-6, f"loss gen={gen}")
        assert_close(fa.grad, fd.grad, 2e-6, f"d/dfake gen={gen}", floor=1e-12)
        if not gen:
            assert_close(ra.grad, rd.grad, 2e-6, "d/dreal", floor=1e-12)
        else:
            assert ra.grad is None


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("depth,B,dev_alpha", [(3, 4, False), (5, 2, False), (5, 8, False), (5, 2, True), (3, 4, True)])
def test_last_epilogue_inside_to_rgb(depth, B, dev_alpha, dt, monkeypatch):
    """functional.EpiRgbOutFn (round 4): the last LayerEpilogue of the synthesis network applied on the fly inside to_rgb (+ the
    fade-in lerp with the upsampled previous-resolution image), against the separate epilogue and to_rgb passes: image and EVERY
    generator parameter gradient (reference models/CustomLayers.py:219-248 + models/GAN.py:199-202).  fp32: the same arithmetic up to
    the order of two multiplications; bf16: the fused path skips one rounding of the epilogue's output.  ``dev_alpha``: the fade-in
    coefficients read from device memory, as the hipGraph-replayed step passes them."""
    from gpu_util import build_mid, load_into, mid_noises, mid_params, pin_noise
    from stylegan.pytorch_amd import functional as F
    gp, _ = mid_params(torch.float64)
    gen, _ = build_mid(dt)
    load_into(gen, gp)
    gen.train(); gen.style_mixing_prob = None
    pin_noise(gen, mid_noises(B))
    z = gu.seeded((B, 512), 11).to(DEV)
    R = 4 << depth
    gimg = gu.seeded((B, 3, R, R), 12).to(DEV)
    calls = []
    orig = F.EpiRgbOutFn.forward
    monkeypatch.setattr(F.EpiRgbOutFn, "forward", staticmethod(lambda ctx, *a: (calls.append(1), orig(ctx, *a))[1]))

    def run():
        for p in gen.parameters():
            p.grad = None
        avg = gen.truncation.avg_latent.clone()
        img = gen(z, depth, torch.tensor([0.4, 0.6], dtype=torch.float32, device=DEV) if dev_alpha else 0.4)
        gen.truncation.avg_latent.copy_(avg)
        (img * gimg).sum().backward()
        return img.detach().clone(), {k: p.grad.detach().clone() for k, p in gen.named_parameters() if p.grad is not None}
    img_on, g_on = run()
    assert len(calls) == 1
    monkeypatch.setattr(F, "FUSE_EPI_RGB", False)
    img_off, g_off = run()
    assert len(calls) == 1
    tol_img, tol_g = (2e-6, 5e-5) if dt == torch.float32 else (6e-3, 4e-2)
    assert rel_err(img_on, img_off) <= tol_img, rel_err(img_on, img_off)
    assert sorted(g_on) == sorted(g_off)
    gmax = max(float(torch.linalg.vector_norm(v)) for v in g_off.values())
    for k in g_on:
        err = float(torch.linalg.vector_norm(g_on[k].double() - g_off[k].double()))
        assert err <= tol_g * float(torch.linalg.vector_norm(g_off[k])) + 1e-6 * gmax, (k, rel_err(g_on[k], g_off[k]))
