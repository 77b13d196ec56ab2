"""-m gpu: every HIP kernel of libsgx_hip.so, called through the module / autograd surface, against the CPU oracle
(fp64) on the same seeded inputs.  fp32 bar: rel-L2 <= 1e-3 per tensor (BASELINE north_star); observed ~1e-6."""
import math

import pytest
import torch
import torch.nn.functional as TF

import golden_util as gu
from gpu_util import DEV, assert_close
from oracle import stylegan_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3          # north_star tolerance, fp32
TIGHT = 2e-5        # what fp32 MFMA (exact fma chain) actually achieves


@pytest.fixture(scope="module", autouse=True)
def _lib():
    from stylegan.pytorch_amd import native
    assert torch.cuda.is_available()
    native.lib()


def conv_module(cin, cout, k=3, **kw):
    from stylegan.pytorch_amd.CustomLayers import BlurLayer, EqualizedConv2d
    if kw.pop("blur", False):
        kw["intermediate"] = BlurLayer([1, 2, 1])
    m = EqualizedConv2d(cin, cout, k, use_wscale=True, **kw)
    with torch.no_grad():
        m.weight.copy_(gu.seeded(m.weight.shape, 5))
        m.bias.copy_(0.1 * gu.seeded(m.bias.shape, 6))
    return m.to(DEV)


CONV_CASES = [
    # (mode, cin, cout, B, H)
    ("plain", 16, 16, 2, 32), ("plain", 32, 64, 3, 16), ("plain", 64, 32, 5, 8), ("plain", 48, 16, 4, 4),
    ("plain", 16, 32, 1, 64), ("plain", 128, 128, 2, 8),
    ("up", 32, 16, 2, 8), ("up", 16, 32, 1, 64), ("up", 64, 64, 3, 4), ("up", 32, 32, 2, 16),
    ("down", 16, 32, 2, 16), ("down", 32, 16, 1, 128), ("down", 64, 64, 5, 8), ("down", 32, 32, 3, 32),
]


@pytest.mark.parametrize("mode,cin,cout,B,H", CONV_CASES)
def test_conv_forward_backward(mode, cin, cout, B, H):
    """EqualizedConv2d plain / up (fused + non-fused semantics, with blur) / down: y, dx, dW, db."""
    m = conv_module(cin, cout, upscale=(mode == "up"), downscale=(mode == "down"), blur=(mode == "up"))
    x = gu.seeded((B, cin, H, H), 7)
    xg = x.to(DEV).requires_grad_(True)
    y = m(xg)
    w64 = m.weight.detach().double().cpu().requires_grad_(True)
    b64 = m.bias.detach().double().cpu().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    y64 = O.eq_conv2d(x64, w64, b64, up=(mode == "up"), down=(mode == "down"), blur_after=(mode == "up"))
    assert_close(y, y64, TIGHT, "y")
    gy = gu.seeded(y64.shape, 8)
    y.backward(gy.to(DEV))
    y64.backward(gy.double())
    assert_close(xg.grad, x64.grad, TIGHT, "dx")
    assert_close(m.weight.grad, w64.grad, TIGHT, "dW")
    assert_close(m.bias.grad, b64.grad, TIGHT, "db")


@pytest.mark.parametrize("mode,cin,cout,B,H", [("plain", 16, 16, 2, 32), ("plain", 64, 32, 3, 16), ("plain", 32, 64, 5, 8),
                                                ("plain", 32, 32, 4, 4), ("up", 32, 16, 2, 16), ("up", 64, 64, 3, 4),
                                                ("down", 16, 32, 2, 32), ("down", 64, 64, 5, 8), ("down", 32, 32, 2, 64),
                                                # 128-channel K-chunks (deep-K variant): forward S/U and, as data gradients, U/D
                                                ("plain", 512, 512, 4, 8), ("plain", 256, 128, 2, 16), ("plain", 512, 512, 4, 4),
                                                ("up", 512, 512, 4, 4), ("up", 256, 256, 2, 8), ("down", 512, 512, 4, 8),
                                                ("plain", 128, 512, 1, 16),
                                                # all-parity-classes-in-one-block transposed conv (GUPA), several tile shapes
                                                ("up", 32, 16, 4, 64), ("up", 64, 32, 2, 32), ("up", 16, 16, 2, 32), ("up", 128, 64, 1, 16),
                                                ("up", 32, 32, 3, 8), ("down", 32, 64, 2, 128)])
def test_conv_bf16_storage(mode, cin, cout, B, H):
    """bf16 activations / operand packs, fp32 accumulation.  Inputs are bf16-exact, so the only error sources are the
    bf16 rounding of the packed weights (y, dx: ~4e-3) and of the stored outputs; dW sees neither (x and dy exact, fp32
    accumulate and output) and must be tight -- that pins the bf16 weight-gradient operand path."""
    from stylegan.pytorch_amd import functional as F
    m = conv_module(cin, cout, upscale=(mode == "up"), downscale=(mode == "down"))
    x = gu.seeded((B, cin, H, H), 7).bfloat16().float()
    xg = F.nhwc(x.to(DEV)).bfloat16().requires_grad_(True)
    y = m.forward_nhwc(xg, skip_bias=True)
    assert y.dtype == torch.bfloat16
    w64 = m.weight.detach().double().cpu().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    y64 = O.eq_conv2d(x64, w64, None, up=(mode == "up"), down=(mode == "down"))
    assert_close(F.nchw_view(y), y64, 1e-2, "y")
    gy = gu.seeded(y64.shape, 8).bfloat16().float()
    y.backward(F.nhwc(gy.to(DEV)).bfloat16())
    y64.backward(gy.double())
    assert_close(F.nchw_view(xg.grad), x64.grad, 1e-2, "dx")
    assert_close(m.weight.grad, w64.grad, 1e-4, "dW (bf16 operands, fp32 accumulate)")


WGRAD2_CASES = [
    # (mode, cin, cout, B, H): shapes sgx_wgrad2_plan accepts (64-multiple channels on the dy side, 32/64 on the x side, enough tiles)
    ("plain", 64, 64, 4, 64), ("plain", 128, 64, 2, 64), ("plain", 64, 128, 4, 32), ("plain", 256, 256, 1, 64),
    ("down", 32, 64, 2, 64), ("down", 64, 64, 2, 64), ("down", 64, 128, 2, 128), ("down", 128, 256, 1, 64),
    ("up", 64, 64, 2, 32), ("up", 128, 32, 2, 32), ("up", 64, 64, 4, 16), ("up", 128, 64, 1, 64),
    ("plain", 320, 512, 16, 32), ("down", 512, 512, 16, 32),        # > 32 channel tiles: taken when a block walks >= 8 pixel tiles
    # the 16 x 16-channel weights (wgrad16_s_kernel, 16x16x32 MFMA): few tiles, many tiles, ragged height, several images
    ("plain", 16, 16, 2, 64), ("plain", 16, 16, 1, 512), ("plain", 16, 16, 3, 96), ("plain", 16, 16, 5, 128),
]


@pytest.mark.parametrize("mode,cin,cout,B,H", WGRAD2_CASES)
def test_wgrad2_vs_oracle(mode, cin, cout, B, H):
    """The second-generation bf16 weight-gradient kernels (32x32x16 MFMA, LDS-DMA staged, transpose reads; wgrad2.hip)
    through the module surface: dW and db against the fp64 oracle on bf16-exact inputs (fp32 accumulation and output, so
    the comparison is tight), and the launch really is the new kernel."""
    from stylegan.pytorch_amd import functional as F, native
    m = conv_module(cin, cout, upscale=(mode == "up"), downscale=(mode == "down"))
    x = gu.seeded((B, cin, H, H), 7).bfloat16().float()
    xg = F.nhwc(x.to(DEV)).bfloat16().requires_grad_(True)
    y = m.forward_nhwc(xg)
    w64 = m.weight.detach().double().cpu().requires_grad_(True)
    b64 = m.bias.detach().double().cpu().requires_grad_(True)
    y64 = O.eq_conv2d(x.double(), w64, b64, up=(mode == "up"), down=(mode == "down"))
    gy = gu.seeded(y64.shape, 8).bfloat16().float()
    native.prof_start(1)
    y.backward(F.nhwc(gy.to(DEV)).bfloat16())
    torch.cuda.synchronize()
    native.prof_start(0)
    names = [r[0] for r in native.prof_records()]
    assert any(("wgrad16_" if cin == 16 else "wgrad2_") in n for n in names), names
    y64.backward(gy.double())
    assert_close(m.weight.grad, w64.grad, 1e-4, "dW")
    assert_close(m.bias.grad, b64.grad, 1e-4, "db")


def test_lds_transpose_read_semantics():
    """ds_read_b64_tr_b16: lane i of a 16-lane group supplies row i/4, column block i%4 and receives column i."""
    from stylegan.pytorch_amd import native as N
    out = torch.zeros(256, dtype=torch.int16, device=DEV)
    N.check(N.lib().sgx_selftest_tr16(N.ptr(out), N.stream()), "selftest")
    got = out.cpu().view(64, 4).tolist()
    expect = [[(l >> 4) * 64 + j * 16 + (l & 15) for j in range(4)] for l in range(64)]
    print("tr16 lane0..3:", got[:4], "lane16:", got[16])
    assert got == expect, got[:20]


def test_conv_edge_shapes():
    """Ragged sizes: non-power-of-two spatial extent and batch not filling the per-block image group."""
    for (B, H, W) in [(1, 4, 4), (3, 12, 20), (7, 8, 8), (2, 24, 40)]:
        m = conv_module(16, 32)
        x = gu.seeded((B, 16, H, W), 9)
        y = m(x.to(DEV))
        y64 = O.eq_conv2d(x.double(), m.weight.detach().double().cpu(), m.bias.detach().double().cpu())
        assert_close(y, y64, TIGHT, f"plain {B}x{H}x{W}")


def test_conv_final_block_513():
    """DiscriminatorTop conv: 513 input channels (512 + stddev) handled through the zero-padded channel group."""
    from stylegan.pytorch_amd.Blocks import DiscriminatorTop
    top = DiscriminatorTop(4, 1, in_channels=32, intermediate_channels=32, gain=math.sqrt(2), use_wscale=True,
                           activation_layer=torch.nn.LeakyReLU(0.2)).to(DEV)
    names = dict(top.named_parameters())
    with torch.no_grad():
        for k, p in names.items():
            p.copy_(gu.fill_value("final_block." + k, p.shape))
    x = gu.seeded((8, 32, 4, 4), 10)
    xg = x.to(DEV).requires_grad_(True)
    out = top(xg)
    p64 = {"final_block." + k: v.detach().double().cpu().requires_grad_(True) for k, v in names.items()}
    x64 = x.double().requires_grad_(True)
    h = O.minibatch_stddev(x64)
    h = O.leaky_relu(O.eq_conv2d(h, p64["final_block.conv.weight"], p64["final_block.conv.bias"]))
    h = h.reshape(8, -1)
    h = O.leaky_relu(O.eq_linear(h, p64["final_block.dense0.weight"], p64["final_block.dense0.bias"], gain=O.SQRT2))
    ref = O.eq_linear(h, p64["final_block.dense1.weight"], p64["final_block.dense1.bias"], gain=1.0)
    assert_close(out, ref, TIGHT, "scores")
    g = gu.seeded((8, 1), 11)
    out.backward(g.to(DEV)); ref.backward(g.double())
    assert_close(xg.grad, x64.grad, 1e-4, "dx")
    for k, p in names.items():
        assert_close(p.grad, p64["final_block." + k].grad, 1e-4, k)


def test_second_order_block():
    """R1-style double backward through a DiscriminatorBlock (+from_rgb): d/dtheta sum((d out / d img)^2)."""
    from stylegan.pytorch_amd import functional as F
    from stylegan.pytorch_amd.Blocks import DiscriminatorBlock
    from stylegan.pytorch_amd.CustomLayers import EqualizedConv2d
    act = torch.nn.LeakyReLU(0.2)
    rgb = EqualizedConv2d(3, 16, 1, gain=math.sqrt(2), use_wscale=True).to(DEV)
    blk = DiscriminatorBlock(16, 32, gain=math.sqrt(2), use_wscale=True, activation_layer=act, blur_kernel=[1, 2, 1]).to(DEV)
    params = dict(list({"from_rgb.0." + k: v for k, v in rgb.named_parameters()}.items())
                  + list({"blocks.0." + k: v for k, v in blk.named_parameters()}.items()))
    with torch.no_grad():
        for k, p in params.items():
            p.copy_(gu.fill_value(k, p.shape))
    img = gu.seeded((2, 3, 32, 32), 12)

    def ours(img_t):
        x = rgb.forward_nhwc(F.nhwc(img_t))
        return F.nchw_view(blk.forward_nhwc(x))

    ig = img.to(DEV).requires_grad_(True)
    out = ours(ig)
    wsum = gu.seeded(out.shape, 13)
    with F.data_grad_only():
        (g1,) = torch.autograd.grad((out * wsum.to(DEV)).sum(), ig, create_graph=True)
    pen = (g1 * g1).sum()
    pen.backward()

    p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in params.items()}
    i64 = img.double().requires_grad_(True)
    h = O.eq_conv2d(i64, p64["from_rgb.0.weight"], p64["from_rgb.0.bias"])
    h = O.leaky_relu(O.eq_conv2d(h, p64["blocks.0.conv0.weight"], p64["blocks.0.conv0.bias"]))
    h = O.blur3(h)
    o64 = O.leaky_relu(O.eq_conv2d(h, p64["blocks.0.conv1_down.weight"], p64["blocks.0.conv1_down.bias"], down=True))
    (g64,) = torch.autograd.grad((o64 * wsum.double()).sum(), i64, create_graph=True)
    pen64 = (g64 * g64).sum()
    pen64.backward()
    assert_close(g1, g64, TIGHT, "d out / d img")
    assert_close(pen, pen64, TIGHT, "penalty")
    for k, p in params.items():
        if k.endswith("weight"):
            assert_close(p.grad, p64[k].grad, 1e-4, "R1 grad " + k)


# (the last five shapes: >= 64 (image, 16-channel group) blocks at <= 64x64 -- the round-6 one-launch kernels gepi_small_fwd / _bwd)
@pytest.mark.parametrize("B,C,H", [(2, 16, 32), (3, 32, 8), (4, 64, 4), (1, 16, 128), (2, 512, 4), (4, 256, 64), (8, 128, 32), (4, 512, 16), (4, 512, 32),
                                   (16, 64, 8), (5, 256, 12)])
def test_layer_epilogue(B, C, H):
    from stylegan.pytorch_amd.CustomLayers import LayerEpilogue
    epi = LayerEpilogue(C, 512, True, True, False, True, True, torch.nn.LeakyReLU(0.2)).to(DEV)
    names = dict(epi.named_parameters())
    with torch.no_grad():
        for k, p in names.items():
            p.copy_(gu.fill_value("epi." + k, p.shape))
    x = gu.seeded((B, C, H, H), 14); noise = gu.seeded((B, 1, H, H), 15); dl = gu.seeded((B, 512), 16)
    epi.top_epi.noise.noise = noise.to(DEV)
    xg = x.to(DEV).requires_grad_(True); dg = dl.to(DEV).requires_grad_(True)
    y = epi(xg, dg)
    p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in names.items()}
    x64 = x.double().requires_grad_(True); d64 = dl.double().requires_grad_(True)
    y64 = O.layer_epilogue(x64, noise.double(), p64["top_epi.noise.weight"], p64["style_mod.lin.weight"],
                           p64["style_mod.lin.bias"], d64)
    assert_close(y, y64, TIGHT, "y")
    gy = gu.seeded(y64.shape, 17)
    y.backward(gy.to(DEV)); y64.backward(gy.double())
    assert_close(xg.grad, x64.grad, 2e-4, "dx")
    assert_close(dg.grad, d64.grad, 2e-4, "d dlatent")
    for k, p in names.items():
        assert_close(p.grad, p64[k].grad, 2e-4, k)


@pytest.mark.parametrize("B,C,H,flags", [(4, 256, 64, 3), (8, 512, 8, 3), (4, 512, 4, 3), (4, 256, 32, 1), (4, 256, 32, 2), (32, 128, 16, 3)])
def test_layer_epilogue_bf16_one_launch_kernels(B, C, H, flags):
    """The small-layer epilogue (round 6: statistics + apply in ONE launch per direction, a block per (image, 16 channels)) in bf16
    storage, straight through ``GEpilogueFn`` -- against the fp64 oracle evaluated on the SAME bf16-rounded input (so the only
    differences are the kernel's fp32 arithmetic and the bf16 rounding of its outputs: 2^-9 relative per element), every stage flag
    combination (activation / instance norm), conv bias folded in (reference models/CustomLayers.py:219-248)."""
    from stylegan.pytorch_amd import functional as F
    from stylegan.pytorch_amd import native as N
    x = gu.seeded((B, H, H, C), 41).bfloat16()
    noise = gu.seeded((B, 1, H, H), 42); nw = 0.5 * gu.seeded((C,), 43); bias = 0.3 * gu.seeded((C,), 44); style = 0.5 * gu.seeded((B, 2 * C), 45)
    xd = x.to(DEV).requires_grad_(True)
    prm = [t.to(DEV).requires_grad_(True) for t in (bias, nw, style)]
    y = F.GEpilogueFn.apply(xd, prm[0], noise.to(DEV), prm[1], prm[2], flags)
    assert y.dtype == torch.bfloat16
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    p64 = [t.double().requires_grad_(True) for t in (bias, nw, style)]
    a = x64 + p64[0].view(1, -1, 1, 1) + p64[1].view(1, -1, 1, 1) * noise.double()
    if flags & N.EPI_ACT:
        a = TF.leaky_relu(a, 0.2)
    if flags & N.EPI_NORM:
        a = O.instance_norm(a)
    s = p64[2].view(B, 2, C, 1, 1)
    y64 = a * (s[:, 0] + 1.0) + s[:, 1]
    assert_close(y.float().permute(0, 3, 1, 2), y64, 3e-3, "y")
    g = gu.seeded((B, H, H, C), 46).bfloat16()
    y.backward(g.to(DEV)); y64.backward(g.double().permute(0, 3, 1, 2))
    assert_close(xd.grad.float().permute(0, 3, 1, 2), x64.grad, 3e-3, "dx")
    for t, t64, what in zip(prm, p64, ("d bias", "d noise weight", "d style")):
        if what == "d bias" and flags == N.EPI_NORM:
            continue                                     # (no activation before the instance norm: the bias gradient is analytically zero)
        assert_close(t.grad, t64.grad, 1e-3, what, floor=1e-4)


def test_pointwise_ops():
    from stylegan.pytorch_amd import functional as F
    x = gu.seeded((2, 8, 8, 32), 20); y = gu.seeded((2, 8, 8, 32), 21)
    xd, yd = x.to(DEV), y.to(DEV)
    assert_close(F.AxpbyFn.apply(xd, yd, 0.3, 0.7), 0.3 * x + 0.7 * y, 1e-6, "axpby")
    assert_close(F.ScaleFn.apply(xd, 4.0), 4.0 * x, 1e-7, "scale")
    xn = x.permute(0, 3, 1, 2)
    assert_close(F.nchw_view(F.BlurFn.apply(xd)), O.blur3(xn.double()), 1e-6, "blur")
    assert_close(F.nchw_view(F.Pool2Fn.apply(xd, 0.25)), TF.avg_pool2d(xn, 2), 1e-6, "avgpool")
    assert_close(F.nchw_view(F.Up2Fn.apply(xd, 1.0)), O.upscale2d(xn), 0, "up2")
    # RGB images (C = 3): widths that take the 16-byte kernels (pool: W % 8 == 0, up: W % 4 == 0; round 6) and widths that fall back to the
    # scalar ones.  The up-sampling is a copy: exact; the pool sums in one fixed order in every variant: the variants agree bit for bit
    for k, shp in enumerate([(2, 16, 16, 3), (3, 24, 40, 3), (1, 6, 12, 3), (2, 10, 6, 3), (2, 64, 72, 3)]):
        img = gu.seeded(shp, 22 + 100 * k).to(DEV)
        inchw = img.cpu().permute(0, 3, 1, 2)
        pooled = F.Pool2Fn.apply(img, 0.25)
        assert_close(F.nchw_view(pooled), TF.avg_pool2d(inchw, 2), 1e-6, f"avgpool rgb {shp}")
        want = 0.25 * ((img[:, 0::2, 0::2] + img[:, 0::2, 1::2]) + (img[:, 1::2, 0::2] + img[:, 1::2, 1::2]))
        assert torch.equal(pooled, want), f"avgpool rgb {shp}: summation order"
        assert_close(F.nchw_view(F.Up2Fn.apply(img, 1.0)), O.upscale2d(inchw), 0, f"up2 rgb {shp}")
    b = 0.1 * gu.seeded((32,), 23)
    assert_close(F.BiasActFn.apply(xd, b.to(DEV), 1.0, 1), TF.leaky_relu(x + b, 0.2), 1e-6, "bias+lrelu")
    assert_close(F.ColSumFn.apply(xd, 1.0), x.double().sum(dim=(0, 1, 2)), 1e-6, "colsum")
    odd = gu.seeded((1037,), 24)                                       # ragged length: vector tail
    assert_close(F.ScaleFn.apply(odd.to(DEV), -2.0), -2.0 * odd, 1e-7, "scale tail")


@pytest.mark.parametrize("B,H,W,C,dtype", [(2, 8, 10, 3, torch.float32), (1, 4, 4, 5, torch.float32), (2, 6, 8, 16, torch.bfloat16), (3, 64, 64, 3, torch.float32)])
def test_pool_fork_joins_the_two_gradients_in_one_pass(B, H, W, C, dtype):
    """functional.PoolForkFn / UpAddFn (round 6): x -> (x, pool2(x)) for the discriminator's image, which feeds the newest block at full
    resolution and the residual from_rgb through a 2x2 average (reference models/GAN.py:423-427).  Against torch's own autograd on the CPU in
    fp64: first order (the join g_x + up(g_y) / 4 is one kernel) and second order (the R1 penalty differentiates that join)."""
    from stylegan.pytorch_amd import functional as F
    x = gu.seeded((B, H, W, C), 81)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xd = x.to(DEV).to(dtype).requires_grad_(True)
    xa, p = F.PoolForkFn.apply(xd, 0.25)
    assert xa.shape == xd.shape and p.shape == (B, H // 2, W // 2, C)
    x64 = x.double().requires_grad_(True)
    p64 = TF.avg_pool2d(x64.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    tol = 1e-6 if dtype == torch.float32 else 6e-3
    assert_close(p.float(), p64, tol, "pool")
    w1 = gu.seeded((B, H, W, C), 82); w2 = gu.seeded((B, H // 2, W // 2, C), 83)
    # first order, both / one branch only
    (g,) = torch.autograd.grad((xa.float() * w1.to(DEV)).sum() + (p.float() * w2.to(DEV)).sum(), xd, retain_graph=True)
    (g64,) = torch.autograd.grad((x64 * w1.double()).sum() + (p64 * w2.double()).sum(), x64, retain_graph=True)
    assert_close(g.float(), g64, tol, "joined gradient")
    (g,) = torch.autograd.grad((p.float() * w2.to(DEV)).sum(), xd, retain_graph=True)
    (g64,) = torch.autograd.grad((p64 * w2.double()).sum(), x64, retain_graph=True)
    assert_close(g.float(), g64, tol, "pooled branch only")
    if dtype != torch.float32:
        return
    # second order: the gradient of |d loss / d x|^2 (R1's structure)
    (gx,) = torch.autograd.grad((xa * xa * w1.to(DEV)).sum() + (p * p * w2.to(DEV)).sum(), xd, create_graph=True)
    (gx64,) = torch.autograd.grad((x64 * x64 * w1.double()).sum() + (p64 * p64 * w2.double()).sum(), x64, create_graph=True)
    assert_close(gx, gx64, 1e-5, "inner gradient")
    (gg,) = torch.autograd.grad((gx * gx).sum(), xd)
    (gg64,) = torch.autograd.grad((gx64 * gx64).sum(), x64)
    assert_close(gg, gg64, 1e-5, "gradient of the squared gradient")


@pytest.mark.parametrize("dtype,C", [(torch.float32, 16), (torch.bfloat16, 32), (torch.bfloat16, 16)])
def test_to_rgb_fork_joins_the_activation_gradients(dtype, C):
    """functional.RgbOutForkFn (round 6): x -> (x, to_rgb(x)) for the generator activation that feeds the next block AND the previous
    resolution's to_rgb under fade-in (reference models/GAN.py:199-202): to_rgb's data gradient lands on top of the other consumer's
    gradient in one pass (sgx_rgb_in_add).  Against the closed form in fp64."""
    from stylegan.pytorch_amd import functional as F
    B, H, W = 2, 16, 24
    x = gu.seeded((B, H, W, C), 91).to(dtype).float()
    w = gu.seeded((3, C, 1, 1), 92); b = 0.1 * gu.seeded((3,), 93); ws = 0.37
    q = gu.seeded((B, H, W, C), 94).to(dtype).float(); r = gu.seeded((B, H, W, 3), 95)
    xd = x.to(DEV).to(dtype).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    xa, img = F.RgbOutForkFn.apply(xd, wd, bd, ws)
    img64 = b.double() + ws * torch.einsum("bhwc,jc->bhwj", x.double(), w.double()[:, :, 0, 0])
    assert_close(img, img64, 1e-5 if dtype == torch.float32 else 1e-5, "to_rgb")
    # the other consumer's gradient arrives as a tensor of the activation's dtype (a data-gradient kernel wrote it)
    loss = (img * r.to(DEV)).sum() + (xa * q.to(DEV).to(dtype)).sum().float()
    loss.backward()
    gx64 = q.double() + ws * torch.einsum("bhwj,jc->bhwc", r.double(), w.double()[:, :, 0, 0])
    assert_close(xd.grad.float(), gx64, 1e-6 if dtype == torch.float32 else 4e-3, "joined gradient")
    assert_close(wd.grad, ws * torch.einsum("bhwj,bhwc->jc", r.double(), x.double()).view(3, C, 1, 1), 1e-5, "d weight")
    assert_close(bd.grad, r.double().sum(dim=(0, 1, 2)), 1e-5, "d bias")
    # only one of the two consumers has a gradient
    xd.grad = None
    xa, img = F.RgbOutForkFn.apply(xd, wd, bd, ws)
    (img * r.to(DEV)).sum().backward()
    assert_close(xd.grad.float(), gx64 - q.double(), 1e-6 if dtype == torch.float32 else 4e-3, "to_rgb branch only")


def test_rgb_convs():
    from stylegan.pytorch_amd.CustomLayers import EqualizedConv2d
    for C in (16, 32, 128, 512):
        fr = EqualizedConv2d(3, C, 1, gain=math.sqrt(2), use_wscale=True).to(DEV)
        to = EqualizedConv2d(C, 3, 1, gain=1, use_wscale=True).to(DEV)
        with torch.no_grad():
            fr.bias.copy_(0.1 * gu.seeded((C,), 30)); to.bias.copy_(0.1 * gu.seeded((3,), 31))
        img = gu.seeded((2, 3, 8, 8), 32)
        ig = img.to(DEV).requires_grad_(True)
        out = to(fr(ig))
        i64 = img.double().requires_grad_(True)
        p = [t.detach().double().cpu().requires_grad_(True) for t in (fr.weight, fr.bias, to.weight, to.bias)]
        ref = O.eq_conv2d(O.eq_conv2d(i64, p[0], p[1]), p[2], p[3], gain=1.0)
        assert_close(out, ref, TIGHT, "rgb roundtrip")
        g = gu.seeded(ref.shape, 33)
        out.backward(g.to(DEV)); ref.backward(g.double())
        assert_close(ig.grad, i64.grad, TIGHT, "d img")
        for ours, r in zip((fr.weight, fr.bias, to.weight, to.bias), p):
            assert_close(ours.grad, r.grad, 1e-4, "rgb param grad")


@pytest.mark.parametrize("B", [2, 4, 8, 16])
def test_minibatch_stddev(B):
    from stylegan.pytorch_amd.CustomLayers import StddevLayer
    x = gu.seeded((B, 32, 4, 4), 40)
    xg = x.to(DEV).requires_grad_(True)
    y = StddevLayer(4, 1)(xg)
    x64 = x.double().requires_grad_(True)
    y64 = O.minibatch_stddev(x64)
    assert_close(y, y64, 1e-6, "y")
    # first and second order: L = sum(w * y); g = dL/dx (create_graph); pen = sum(g^2); d pen / dx
    w = gu.seeded(y64.shape, 41)
    (g1,) = torch.autograd.grad((y * w.to(DEV)).sum(), xg, create_graph=True)
    (g64,) = torch.autograd.grad((y64 * w.double()).sum(), x64, create_graph=True)
    assert_close(g1, g64, 1e-5, "dx")
    (g1 * g1).sum().backward(); (g64 * g64).sum().backward()
    # group of 2: d0 = -d1, the second-order term cancels to O(eps) and fp32 keeps ~3 digits of it
    assert_close(xg.grad, x64.grad, 2e-3 if B == 2 else 1e-4, "second order")


def test_mapping_and_linear():
    from stylegan.pytorch_amd.GAN import GMapping
    gm = GMapping(512, 512, dlatent_broadcast=None, mapping_layers=3).to(DEV)
    names = dict(gm.named_parameters())
    with torch.no_grad():
        for k, p in names.items():
            p.copy_(gu.fill_value("g_mapping." + k, p.shape))
    z = gu.seeded((5, 512), 50)
    zg = z.to(DEV).requires_grad_(True)
    w = gm(zg)
    p64 = {"g_mapping." + k: v.detach().double().cpu().requires_grad_(True) for k, v in names.items()}
    z64 = z.double().requires_grad_(True)
    w64 = O.g_mapping(p64, z64, 3)
    assert_close(w, w64, TIGHT, "w")
    g = gu.seeded(w64.shape, 51)
    w.backward(g.to(DEV)); w64.backward(g.double())
    assert_close(zg.grad, z64.grad, 1e-4, "dz")
    for k, p in names.items():
        assert_close(p.grad, p64["g_mapping." + k].grad, 1e-4, k)


def test_fused_adam_clip_ema():
    from stylegan.pytorch_amd.optim import FusedAdam, clip_and_step, ema_update
    torch.manual_seed(0)
    shapes = [(7,), (33, 5), (4, 4, 3, 3), (1,), (257, 129)]
    ps = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    a = FusedAdam(ps, lr=0.003, betas=(0.0, 0.99), eps=1e-8)
    b = torch.optim.Adam(qs, lr=0.003, betas=(0.0, 0.99), eps=1e-8)
    for it in range(3):
        for i, (p, q) in enumerate(zip(ps, qs)):
            if it == 0 and i == 1:
                p.grad = None; q.grad = None                       # inactive parameter: skipped, own step count
                continue
            g = torch.randn_like(p) * 20
            p.grad = g.clone(); q.grad = g.clone()
        out = clip_and_step(a, 10.0)
        tot = torch.nn.utils.clip_grad_norm_(qs, 10.0)
        b.step()
        assert abs(math.sqrt(float(out[0])) - float(tot)) <= 1e-5 * float(tot)
        for p, q in zip(ps, qs):
            assert_close(p, q, 1e-6, f"adam it{it}")
    m_t = torch.nn.Linear(8, 8).to(DEV); m_s = torch.nn.Linear(8, 8).to(DEV)
    ref = {k: 0.999 * v.detach().clone() + 0.001 * dict(m_s.named_parameters())[k].detach() for k, v in m_t.named_parameters()}
    ema_update(m_t, m_s, 0.999)
    for k, v in m_t.named_parameters():
        assert_close(v, ref[k], 1e-6, "ema")


def test_library_profiler_names_and_times_launches():
    """sgx_prof_*: every launch is bracketed by HIP events inside the library; names are what rocprofv3 prints."""
    from stylegan.pytorch_amd import functional as F, native
    w = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=DEV))
    x = torch.randn(2, 64, 64, 32, device=DEV, requires_grad=True)
    F.conv(x, w, None, "S", 0.1)                                       # weight pack happens outside the profiled region
    native.prof_start(1)
    y = F.conv(x, w, None, "S", 0.1)
    y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    native.prof_start(0)
    recs = native.prof_records()
    names = [r[0] for r in recs]
    assert any(n.startswith("void conv_kernel<float, 16, 0,") for n in names), names
    assert any(n.startswith("void wgrad_kernel<float, 0,") for n in names), names
    assert any("wgrad_finish_kernel" in n for n in names), names
    conv = [r for r in recs if r[0].startswith("void conv_kernel")]
    assert len(conv) == 2                                              # forward + data gradient
    for name, ms, flops, nbytes, desc in conv:
        assert 0.0 < ms < 50.0 and flops == 2.0 * 9 * 32 * 32 * 2 * 64 * 64 and desc == "convS B2 64x64 32->32"
    # mode 2: only the kernel of a chosen record
    native.prof_start(2, names.index(conv[0][0]))
    y = F.conv(x, w, None, "S", 0.1)
    y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    native.prof_start(0)
    only = native.prof_records()
    assert len(only) == 2 and all(r[0] == conv[0][0] for r in only)
    native.prof_start(1); native.prof_start(0)
    assert native.prof_records() == []


@pytest.mark.parametrize("mode,H,cin,cout,dt", [("S", 64, 32, 64, torch.float32), ("D", 64, 32, 64, torch.float32),
                                                ("S", 128, 16, 16, torch.bfloat16), ("D", 256, 16, 32, torch.bfloat16),
                                                ("S", 8, 512, 512, torch.bfloat16),
                                                # second-generation weight-gradient kernels (wgrad2.hip): bias sums from their MFMA-against-ones
                                                ("S", 64, 64, 64, torch.bfloat16), ("S", 64, 128, 64, torch.bfloat16),
                                                ("D", 64, 32, 64, torch.bfloat16), ("D", 128, 64, 128, torch.bfloat16)])
def test_bias_gradient_fused_into_weight_gradient(mode, H, cin, cout, dt):
    """db comes out of the wgrad pass (MFMA against a tile of ones) and equals sum(gy) over batch and pixels."""
    from stylegan.pytorch_amd import functional as F, native
    torch.manual_seed(3)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV))
    b = torch.nn.Parameter(torch.randn(cout, device=DEV))
    x = torch.randn(4, H, H, cin, device=DEV).to(dt).requires_grad_(True)
    native.prof_start(1)
    y = F.conv(x, w, b, mode, 0.05, act=native.ACT_LRELU)
    g = torch.randn_like(y)
    (gw, gb) = torch.autograd.grad(y, (w, b), g)
    torch.cuda.synchronize()
    native.prof_start(0)
    assert not any("colsum" in r[0] for r in native.prof_records()), "bias gradient took a separate pass"
    gz = (g.float() * torch.where(y.float() > 0, 1.0, 0.2)).to(dt).float()          # what the wgrad kernel reads
    ref = gz.sum(dim=(0, 1, 2))
    assert_close(gb, ref, 2e-3 if dt == torch.bfloat16 else 1e-4, "fused bias gradient")


def test_prepack_rebuilds_all_stale_packs_in_one_launch():
    from stylegan.pytorch_amd import functional as F, native
    torch.manual_seed(4)
    ws = [torch.nn.Parameter(torch.randn(32, 16, 3, 3, device=DEV)), torch.nn.Parameter(torch.randn(64, 32, 3, 3, device=DEV)),
          torch.nn.Parameter(torch.randn(16, 32, 3, 3, device=DEV))]
    modes = ["S", "D", "U"]
    xs = [torch.randn(2, 16, 16, w.shape[1], device=DEV).bfloat16() for w in ws]
    ref = [F.conv(x, w, None, m, 0.07).clone() for x, w, m in zip(xs, ws, modes)]          # lazy single packs, records the usage
    with torch.no_grad():
        for w in ws:
            w.mul_(2.0)                                                                    # bumps torch's version counter
    native.prof_start(1)
    F.prepack(ws)
    out = [F.conv(x, w, None, m, 0.07) for x, w, m in zip(xs, ws, modes)]
    torch.cuda.synchronize()
    native.prof_start(0)
    names = [r[0] for r in native.prof_records()]
    assert sum("pack_weight_multi_kernel" in n for n in names) == 1 and not any("pack_weight_kernel" in n for n in names), names
    for o, r in zip(out, ref):
        assert_close(o, 2.0 * r.float(), 1e-2, "conv after batched re-pack")               # bf16 packs of 2w vs 2 * packs of w
    # and the adjoint packs: data gradients through the re-packed weights
    x = xs[0].clone().requires_grad_(True)
    y = F.conv(x, ws[0], None, "S", 0.07)
    (gx,) = torch.autograd.grad(y, x, torch.ones_like(y))
    wq = (ws[0].detach() * 0.07).bfloat16().float()
    gref = TF.conv_transpose2d(torch.ones(2, 32, 16, 16, device=DEV), wq, padding=1).permute(0, 2, 3, 1)
    assert_close(gx, gref, 1e-2, "data gradient after batched re-pack")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_act_blur_first_and_second_order(dt):
    """blur(lrelu(z)) as one op: forward, backward and the double backward the R1 term needs, vs torch fp32."""
    from stylegan.pytorch_amd import functional as F
    torch.manual_seed(8)
    B, H, C = 2, 24, 32
    z = torch.randn(B, H, H, C, device=DEV).to(dt).requires_grad_(True)
    k = torch.tensor([1., 2., 1.], device=DEV); k = (k[:, None] * k[None, :] / 16.0)[None, None].repeat(C, 1, 1, 1)

    def ref(zz):
        a = TF.leaky_relu(zz.float().permute(0, 3, 1, 2), 0.2)
        return TF.conv2d(a, k, padding=1, groups=C).permute(0, 2, 3, 1)

    y = F.ActBlurFn.apply(z)
    zr = z.detach().float().requires_grad_(True)
    yr = ref(zr)
    tol = 1e-5 if dt == torch.float32 else 8e-3
    assert_close(y, yr, tol, "act_blur forward")
    g = torch.randn_like(y)
    (gz,) = torch.autograd.grad(y, z, g, create_graph=True)
    (gzr,) = torch.autograd.grad(yr, zr, g.float(), create_graph=True)
    assert_close(gz, gzr, tol, "act_blur backward")
    # second order: d/dg of <gz, v> (the path R1 differentiates: linear in g)
    v = torch.randn_like(gz)
    gg = g.clone().requires_grad_(True)
    (gz2,) = torch.autograd.grad(F.ActBlurFn.apply(z), z, gg, create_graph=True)
    (dg,) = torch.autograd.grad(gz2, gg, v)
    ggr = g.float().clone().requires_grad_(True)
    (gz2r,) = torch.autograd.grad(ref(zr), zr, ggr, create_graph=True)
    (dgr,) = torch.autograd.grad(gz2r, ggr, v.float())
    assert_close(dg, dgr, tol, "act_blur double backward")


# (tensors large enough for the streaming blur kernels: 8-row strips, >= 131072 strip lanes; heights that are not strip multiples)
@pytest.mark.parametrize("dt,B,H,W,C", [(torch.bfloat16, 6, 200, 512, 16), (torch.bfloat16, 10, 52, 256, 64), (torch.bfloat16, 16, 36, 128, 128),
                                        (torch.bfloat16, 3, 99, 448, 32), (torch.float32, 16, 72, 256, 16)])
def test_one_load_blur_kernel_writes_the_bits_of_the_three_load_kernel(dt, B, H, W, C, monkeypatch):
    """csrc/pointwise.hip blur3x3s_kernel (round 4: one global load per input vector, neighbours by lane exchange, pre-op once per
    element) against blur3x3_kernel (SGX_BLUR_SHFL=0, read at every launch): every mode, bit for bit -- and mode 0 / 1 against torch
    (reference models/CustomLayers.py:179-217: the depthwise [1,2,1]x[1,2,1]/16 blur with zero padding)."""
    from stylegan.pytorch_amd import native as N
    L = N.lib()
    torch.manual_seed(C + H)
    x = torch.randn(B, H, W, C, device=DEV).to(dt)
    z = torch.randn(B, H, W, C, device=DEV).to(dt)
    bits = torch.randint(0, 256, (B, H, W, C // 8), device=DEV, dtype=torch.uint8)
    iview = torch.int16 if dt == torch.bfloat16 else torch.int32
    k = torch.tensor([1., 2., 1.], device=DEV); k = (k[:, None] * k[None, :] / 16.0)[None, None].repeat(C, 1, 1, 1)
    for mode in ([0, 1, 2, 4, 5] if dt == torch.bfloat16 else [0, 1, 2]):
        outs = []
        for v in ("0", None, "2"):
            if v is None:
                monkeypatch.delenv("SGX_BLUR_SHFL", raising=False)
            else:
                monkeypatch.setenv("SGX_BLUR_SHFL", v)
            y = torch.full_like(x, 7.0)
            if mode >= 4:
                N.check(L.sgx_blur3x3_bits(N.ptr(x), N.ptr(bits), N.ptr(y), B, H, W, C, mode - 2, N.dt(x), N.stream()), "blur_bits")
            else:
                N.check(L.sgx_blur3x3_act(N.ptr(x), N.ptr(z if mode == 2 else None), N.ptr(y), B, H, W, C, mode, N.dt(x), N.stream()), "blur_act")
            outs.append(y)
        assert torch.equal(outs[0].view(iview), outs[1].view(iview)), (mode, "default depth")
        assert torch.equal(outs[0].view(iview), outs[2].view(iview)), (mode, "depth 2")
        if mode in (0, 1):
            a = x.float().permute(0, 3, 1, 2)
            ref = TF.conv2d(TF.leaky_relu(a, 0.2) if mode == 1 else a, k, padding=1, groups=C).permute(0, 2, 3, 1)
            assert_close(outs[1], ref, 1e-5 if dt == torch.float32 else 8e-3, f"blur mode {mode}")


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("B,K,Nn", [(4, 512, 512), (8, 512, 1024), (4, 512, 32), (3, 100, 36)])
def test_fused_linear_matches_torch(B, K, Nn, act):
    from stylegan.pytorch_amd import functional as F
    torch.manual_seed(B + K + Nn)
    x = torch.randn(B, K, device=DEV, requires_grad=True)
    w = torch.nn.Parameter(torch.randn(Nn, K, device=DEV)); b = torch.nn.Parameter(torch.randn(Nn, device=DEV))
    w_mul, b_mul = 0.37, 0.5
    y = F.linear_fused(x, w, b, w_mul, b_mul, act)
    yr = TF.linear(x, w * w_mul, b * b_mul)
    if act:
        yr = TF.leaky_relu(yr, 0.2)
    assert_close(y, yr, 1e-5, "linear fwd")
    g = torch.randn_like(y)
    got = torch.autograd.grad(y, (x, w, b), g)
    ref = torch.autograd.grad(yr, (x, w, b), g)
    for a, r, n in zip(got, ref, ("gx", "gw", "gb")):
        assert_close(a, r, 1e-5, n)
    y2 = F.linear_fused(x, w, None, w_mul, b_mul, act)                      # no bias
    yr2 = TF.linear(x, w * w_mul)
    assert_close(y2, TF.leaky_relu(yr2, 0.2) if act else yr2, 1e-5, "linear fwd no bias")


@pytest.mark.parametrize("B", [4, 8, 3])
def test_grouped_style_affines_match_per_layer_linears(B):
    """GroupedStyleFn (one launch for all style affines, two for their backward) against torch F.linear per layer."""
    from stylegan.pytorch_amd import functional as F
    torch.manual_seed(B)
    L, D = 6, 512
    ns = [1024, 1024, 512, 64, 32, 32]
    ws = [torch.nn.Parameter(torch.randn(n, D, device=DEV)) for n in ns]
    bs = [torch.nn.Parameter(torch.randn(n, device=DEV)) for n in ns]
    lm = torch.randn(L + 2, B, D, device=DEV, requires_grad=True)           # two trailing layers have no group
    meta = tuple((i, 0.044 + 0.001 * i, 1.0 - 0.1 * i) for i in range(L))
    outs = F.GroupedStyleFn.apply(lm, meta, *ws, *bs)
    lmr = lm.detach().clone().requires_grad_(True)
    wr = [w.detach().clone().requires_grad_(True) for w in ws]; br = [b.detach().clone().requires_grad_(True) for b in bs]
    refs = [TF.linear(lmr[i], wr[i] * meta[i][1], br[i] * meta[i][2]) for i in range(L)]
    gs = [torch.randn_like(o) for o in outs]
    for o, r in zip(outs, refs):
        assert_close(o, r, 1e-5, "style fwd")
    torch.autograd.backward(outs, gs)
    torch.autograd.backward(refs, gs)
    assert_close(lm.grad, lmr.grad, 1e-5, "dlatent gradient")
    for i in range(L):
        assert_close(ws[i].grad, wr[i].grad, 1e-5, f"dW{i}")
        assert_close(bs[i].grad, br[i].grad, 1e-5, f"db{i}")


@pytest.mark.parametrize("layout", ["hwc", "chw"])
@pytest.mark.parametrize("B,H,W", [(3, 6, 8), (2, 64, 64), (4, 1024, 1024)])
def test_images_from_uint8(layout, B, H, W):
    """Device-side input pipeline against the oracle: bit-exact in fp32 (same operation order, IEEE division),
    bf16 = the fp32 result rounded to nearest even; horizontal flips by per-image decision; every byte value occurs."""
    from stylegan.pytorch_amd import functional as F
    g = torch.Generator().manual_seed(B * H + W)
    hwc = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g)
    hwc.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)
    flip = [bool(i % 2) for i in range(B)]
    src = (hwc if layout == "hwc" else hwc.permute(0, 3, 1, 2).contiguous()).to(DEV)
    if B * H * W <= 1 << 16:
        ref, ref_fl = O.images_u8_to_float(hwc).to(DEV), O.images_u8_to_float(hwc, flip).to(DEV)
    else:                                                 # full size: the oracle's 256 levels, gathered on the GPU
        lut = O.images_u8_to_float(torch.arange(256, dtype=torch.uint8).reshape(1, 1, 256, 1).expand(1, 1, 256, 3))[0, 0, 0].to(DEV)
        ref = lut[hwc.to(DEV).permute(0, 3, 1, 2).long()]
        ref_fl = torch.stack([r.flip(-1) if f else r for r, f in zip(ref, flip)])
    out = F.images_from_uint8(src, layout=layout)
    assert out.shape == (B, 3, H, W) and out.permute(0, 2, 3, 1).is_contiguous()      # NHWC storage, NCHW view
    assert torch.equal(out, ref)
    assert torch.equal(F.images_from_uint8(src, flip=flip, layout=layout), ref_fl)
    assert torch.equal(F.images_from_uint8(src, flip=flip, out_dtype=torch.bfloat16, layout=layout), ref_fl.to(torch.bfloat16))
    with pytest.raises(Exception):
        F.images_from_uint8(src.float(), layout=layout)


def test_images_from_uint8_vs_transform_chain_fixture(golden_dir):
    """The device input pipeline (sgx_images_u8_to_nhwc) against the PIL-decoded fixture of the reference's transform chain
    (tests/golden/images_u8.npz, data/transforms.py:27-32): bit-exact, both layouts, with the fixture's flips."""
    import os
    import numpy as np
    from stylegan.pytorch_amd import functional as F
    g = np.load(os.path.join(golden_dir, "images_u8.npz"))
    u8 = torch.from_numpy(g["u8"]).to(DEV)
    flips = [bool(f) for f in g["flips"]]
    want = torch.from_numpy(g["out"]).to(DEV)
    got = F.images_from_uint8(u8, flip=flips, layout="hwc")
    assert got.shape == want.shape and torch.equal(got, want)
    got = F.images_from_uint8(u8.permute(0, 3, 1, 2).contiguous(), flip=flips, layout="chw")
    assert torch.equal(got, want)


def test_images_from_uint8_feeds_the_step():
    """The uint8 batch drives the discriminator exactly like the float batch it stands for."""
    from stylegan.pytorch_amd import functional as F
    from test_gpu_graphs import make
    sg = make(False, torch.float32)
    u8 = torch.randint(0, 256, (4, 128, 128, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    as_float = O.images_u8_to_float(u8).to(DEV)
    with torch.no_grad():
        a = sg.dis(F.images_from_uint8(u8.to(DEV)), 5, 1.0)
        b = sg.dis(as_float, 5, 1.0)
    assert torch.equal(a, b)


# second-generation bf16 convolution (conv2.hip): 32x32x16 MFMA, LDS-DMA double-buffered stages.  Both block shapes
# (4 / 8 waves), all three geometries, against the fp64 oracle on bf16-exact operands and against the first-generation kernel
# on the same packs.
CONV2_CASES = [
    # (geo, cin, cout, B, H, W)   1 / 2 / 4+ K-chunks (static vs re-staged weights), 1..4 channel blocks, ragged rows, many tiles
    ("S", 32, 64, 2, 32, 32), ("S", 64, 64, 3, 16, 32), ("S", 64, 128, 2, 40, 64), ("S", 128, 64, 1, 64, 32), ("S", 256, 256, 2, 8, 32),
    ("S", 32, 128, 5, 24, 96), ("S", 96, 192, 1, 33, 32), ("S", 64, 64, 9, 64, 64),
    ("S", 32, 32, 2, 64, 64), ("S", 64, 32, 3, 24, 32), ("S", 32, 96, 1, 40, 64),        # 32-channel output blocks (MF = 1)
    ("S", 32, 128, 2, 520, 512), ("U", 32, 64, 1, 512, 1024),                            # enough tiles for the XCD-band tile order, ragged rows
    ("D", 32, 64, 2, 64, 64), ("D", 64, 32, 3, 32, 64), ("D", 128, 128, 1, 80, 128), ("D", 32, 96, 2, 36, 64), ("D", 64, 64, 5, 128, 64),
    ("U", 32, 32, 2, 32, 32), ("U", 64, 64, 3, 16, 32), ("U", 128, 32, 1, 40, 64), ("U", 32, 96, 2, 17, 32), ("U", 64, 32, 5, 64, 64),
    # the 16-channel layers (round 3: half-width planar stages / 16 real output channels in a 32-channel block): the three shapes
    # of the 1024x1024 level -- small, ragged rows, and enough tiles for the XCD-band tile order and two blocks per CU
    ("S", 16, 16, 2, 32, 32), ("S", 16, 16, 3, 40, 64), ("S", 16, 16, 1, 520, 512), ("S", 16, 16, 5, 17, 96),
    ("D", 16, 32, 2, 64, 64), ("D", 16, 32, 3, 36, 128), ("D", 16, 32, 1, 512, 1024),
    ("U", 32, 16, 2, 32, 32), ("U", 32, 16, 3, 17, 64), ("U", 32, 16, 1, 256, 512),
]


@pytest.mark.parametrize("variant", [4, 8])
@pytest.mark.parametrize("geo,cin,cout,B,H,W", CONV2_CASES)
def test_conv2_variants_vs_oracle(variant, geo, cin, cout, B, H, W):
    from stylegan.pytorch_amd import functional as F
    from stylegan.pytorch_amd import native as N
    w = gu.seeded((cout, cin, 3, 3), 5).to(DEV)
    bias = None if geo == "U" else (0.5 * gu.seeded((cout,), 6)).to(DEV)
    scale = O.he_w_mul(cin * 9, math.sqrt(2))
    x = gu.seeded((B, cin, H, W), 7).bfloat16().float()
    xn = F.nhwc(x.to(DEV)).bfloat16()
    wq, _ = F.packs(w, geo, scale, cin, torch.bfloat16)
    OH, OW = (H // 2, W // 2) if geo == "D" else ((2 * H, 2 * W) if geo == "U" else (H, W))
    act = 0 if geo == "U" else 1
    L = N.lib()
    outs = {}
    for v in (0, variant):
        y = torch.full((B, OH, OW, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
        N.check(L.sgx_conv_variant({"S": 0, "D": 1, "U": 2}[geo], N.ptr(xn), N.ptr(wq), N.ptr(bias), N.ptr(y), B, H, W, cin, cout, act,
                                   N.BF16, v, N.stream()), "variant")
        outs[v] = F.nchw_view(y).float()
    # oracle on the bf16-rounded packed weights: the only remaining error is the bf16 rounding of the stored output
    k = 3 if geo == "S" else 4
    wr = wq.float().view(k, k, cout, cin).permute(2, 3, 0, 1).double().cpu()
    if geo == "S":
        ref = TF.conv2d(x.double(), wr, bias.double().cpu(), padding=1)
    elif geo == "D":
        ref = TF.conv2d(x.double(), wr, bias.double().cpu(), stride=2, padding=1)
    else:
        ref = TF.conv_transpose2d(x.double(), wr.permute(1, 0, 2, 3), stride=2, padding=1)
    if act:
        ref = TF.leaky_relu(ref, 0.2)
    assert torch.isfinite(outs[variant]).all()
    assert_close(outs[variant], ref, 4e-3, f"conv2 {geo} v{variant} vs oracle (bf16 output rounding only)")
    assert_close(outs[variant], outs[0], 4e-3, f"conv2 {geo} v{variant} vs first-generation kernel")
    # fp32 agreement before the rounding: at most one bf16 ulp apart anywhere
    d = (outs[variant] - ref.float().to(DEV)).abs()
    assert float((d / (ref.float().to(DEV).abs() + 1e-3)).max()) < 2 ** -7


# conv3_kernel (round 5: hand-pipelined fragment reads, counted waits, optional 4 pixel rows per wave, optional interleaved DMA):
# same LDS image, same accumulation order as conv2_kernel -- every configuration must write conv2's bits, on ragged rows, partial
# last tiles, 1 / 2 / 4+ K-chunks (static vs re-staged weights), several channel blocks, and enough tiles for the XCD-band order.
CONV3_CASES = [c for c in CONV2_CASES if c[0] in "SD" and c[1] % 32 == 0 and c[2] % 64 == 0] + [
    ("S", 512, 512, 2, 32, 32), ("S", 128, 128, 4, 128, 128), ("D", 256, 512, 2, 64, 64), ("D", 64, 128, 3, 72, 64), ("S", 64, 64, 2, 20, 32)]


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("geo,cin,cout,B,H,W", CONV3_CASES)
def test_conv3_configurations_write_conv2_bits(cfg, geo, cin, cout, B, H, W):
    from stylegan.pytorch_amd import functional as F
    from stylegan.pytorch_amd import native as N
    w = gu.seeded((cout, cin, 3, 3), 5).to(DEV)
    bias = (0.5 * gu.seeded((cout,), 6)).to(DEV)
    scale = O.he_w_mul(cin * 9, math.sqrt(2))
    xn = F.nhwc(gu.seeded((B, cin, H, W), 7).to(DEV)).bfloat16()
    wq, _ = F.packs(w, geo, scale, cin, torch.bfloat16)
    OH, OW = (H // 2, W // 2) if geo == "D" else (H, W)
    L = N.lib()
    outs = []
    for v in (8, 30 + cfg):
        y = torch.full((B, OH, OW, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
        N.check(L.sgx_conv_variant({"S": 0, "D": 1}[geo], N.ptr(xn), N.ptr(wq), N.ptr(bias), N.ptr(y), B, H, W, cin, cout, 1, N.BF16, v, N.stream()), "variant")
        outs.append(y)
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), f"conv3 configuration {cfg} differs from conv2 on {geo} {cin}->{cout} B{B} {H}x{W}"


SPLITK_SHAPES = [  # (geo, B, H, Cin, Cout): the 512-channel stride-2 layers of the 1024 model at a small batch (+ ragged K-chunk counts, a narrower one), input H x H
    ("D", 4, 16, 512, 512), ("D", 4, 8, 512, 512), ("D", 8, 8, 512, 512), ("D", 4, 8, 544, 512), ("D", 2, 16, 256, 512), ("D", 1, 16, 512, 512), ("D", 4, 32, 512, 512),
]


@pytest.mark.parametrize("geo,B,H,Cin,Cout", SPLITK_SHAPES)
def test_split_k_convolution_vs_fp32_reference_and_unsplit_kernel(geo, B, H, Cin, Cout):
    """``sgx_conv_splitk`` (round 6: the reduction over input channels split over blocks for the launches that leave most of the chip idle)
    against an fp32 torch convolution of the same bf16 operands (bar 3e-3 rel-L2: one bf16 rounding of the output) and against the UNSPLIT
    kernel of the same shape (same products, another summation order: a few last-bit roundings -- rel-L2 <= 2e-3), with bias, LeakyReLU
    and -- 3x3 -- the output mask."""
    from stylegan.pytorch_amd import functional as F
    from stylegan.pytorch_amd import native as N
    L = N.lib()
    gi = "SDU".index(geo)
    k = 3 if geo == "S" else 4
    wsb = L.sgx_conv_splitk_ws_bytes(gi, B, H, H, Cin, Cout, N.BF16)
    if not wsb:
        pytest.skip("this shape does not split under the plan (enough blocks without it)")
    torch.manual_seed(H + Cin + gi)
    w = torch.randn(Cout, Cin, k, k, device=DEV)
    x = torch.randn(B, H, H, Cin, device=DEV).bfloat16()
    wq, _ = F.packs(w, geo, 0.05, Cin, torch.bfloat16)
    bias = None if geo == "U" else torch.randn(Cout, device=DEV)
    act = 0 if geo == "U" else 1
    oh = H if geo == "S" else (H // 2 if geo == "D" else 2 * H)
    mask = torch.randn(B, oh, oh, Cout, device=DEV).bfloat16() if geo == "S" else None
    y = torch.empty((B, oh, oh, Cout), dtype=torch.bfloat16, device=DEV)
    ws = N.workspace(wsb, x.device)
    N.check(L.sgx_conv_splitk(gi, N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), N.ptr(mask), B, H, H, Cin, Cout, act, N.BF16, N.ptr(ws), wsb, N.stream()),
            "sgx_conv_splitk")
    y0 = torch.empty_like(y)
    if geo == "S":
        N.check(L.sgx_conv3x3(N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y0), B, H, H, Cin, Cout, act, N.ptr(mask), N.BF16, N.stream()), "sgx_conv3x3")
    elif geo == "D":
        N.check(L.sgx_conv4x4s2_down(N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y0), B, H, H, Cin, Cout, act, N.BF16, N.stream()), "sgx_conv4x4s2_down")
    else:
        N.check(L.sgx_conv4x4s2_up(N.ptr(x), N.ptr(wq), N.ptr(y0), B, H, H, Cin, Cout, N.BF16, N.stream()), "sgx_conv4x4s2_up")
    torch.cuda.synchronize()
    wr = wq.float().view(k, k, Cout, Cin).permute(2, 3, 0, 1).contiguous()               # the bf16-rounded operands
    xi = x.float().permute(0, 3, 1, 2)
    if geo == "S":
        ref = TF.conv2d(xi, wr, bias, padding=1)
    elif geo == "D":
        ref = TF.conv2d(xi, wr, bias, stride=2, padding=1)
    else:
        ref = TF.conv_transpose2d(xi, wr.permute(1, 0, 2, 3), stride=2, padding=1)
    if act:
        ref = TF.leaky_relu(ref, 0.2)
    ref = ref.permute(0, 2, 3, 1)
    if mask is not None:
        ref = ref * torch.where(mask.float() > 0, 1.0, 0.2)
    assert_close(y, ref, 3e-3, f"split-K conv{geo} vs fp32")
    assert_close(y, y0, 2e-3, f"split-K conv{geo} vs the unsplit kernel")
