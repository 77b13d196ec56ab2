"""-m gpu: size-independent properties at BASELINE.json's FULL sizes (1024x1024, depth index 8, batch 4), where the
CPU oracle is too slow to be the checker:

* adjoint identities  <conv(x), y> == <x, conv^T(y)>  and  <dW, V> == <conv_V(x), y>  for the three MFMA convolution
  geometries (the dgrad / wgrad kernels are exact transposes of the forward kernel, fp32);
* linearity of the convolution in x;
* bit-reproducibility: the kernels use no floating-point atomics, so two runs of a full G+D iteration from the same
  state give bit-identical losses and parameters;
* the 1024x1024 networks agree between bf16 storage and fp32 within the bf16 budget.
"""
import random

import pytest
import torch

from gpu_util import DEV

pytestmark = pytest.mark.gpu


def dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("mode,H,cin,cout", [("S", 1024, 16, 16), ("D", 1024, 16, 32), ("U", 512, 32, 16), ("S", 128, 128, 128)])
def test_adjoint_identities_full_resolution(mode, H, cin, cout):
    from stylegan.pytorch_amd import functional as F
    torch.manual_seed(1)
    B = 2
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV))
    x = torch.randn(B, H, H, cin, device=DEV, requires_grad=True)
    y = F.conv(x, w, None, mode, 0.05)
    g = torch.randn_like(y)
    (gx, gw) = torch.autograd.grad(y, (x, w), g)
    lhs = dot(y, g)
    assert abs(dot(x, gx) - lhs) <= 2e-4 * abs(lhs) + 1e-3, "dgrad is not the transpose of the forward kernel"
    # <dW, W> == <conv_W(x), g>  (the op is linear in W)
    assert abs(dot(w, gw) - lhs) <= 2e-4 * abs(lhs) + 1e-3, "wgrad is not the transpose of the forward kernel"
    # linearity in x
    x2 = torch.randn_like(x)
    with torch.no_grad():
        y2 = F.conv(x2, w, None, mode, 0.05)
        y12 = F.conv(x + 2.0 * x2, w, None, mode, 0.05)
    err = (y12 - (y + 2.0 * y2)).abs().max().item()
    assert err <= 1e-4 * y12.abs().max().item()


def build(act_dtype, seed=0):
    from stylegan.pytorch_amd.GAN import StyleGAN
    torch.manual_seed(seed)
    opt = dict(learning_rate=0.003, beta_1=0, beta_2=0.99, eps=1e-8)
    sg = StyleGAN("linear", 1024, 3, 512,
                  g_args=dict(latent_size=512, mapping_layers=8, blur_filter=[1, 2, 1], truncation_psi=-1.0, truncation_cutoff=8),
                  d_args=dict(use_wscale=True, blur_filter=[1, 2, 1]), g_opt_args=opt, d_opt_args=opt, loss="logistic",
                  use_ema=True, device=torch.device(DEV), act_dtype=act_dtype)
    return sg


def one_step(sg, seed):
    torch.manual_seed(seed); random.seed(seed)
    torch.cuda.manual_seed(seed)
    gen = torch.Generator(device=DEV); gen.manual_seed(seed)
    z = torch.randn(4, 512, device=DEV, generator=gen)
    real = torch.randn(4, 1024, 1024, 3, device=DEV, generator=gen).permute(0, 3, 1, 2)
    d = sg.optimize_discriminator(z, real, 8, 0.5)
    g = sg.optimize_generator(z, real, 8, 0.5)
    return d, g


def test_full_step_is_reproducible_and_finite():
    """FFHQ-1024 model, depth index 8, batch 4, bf16: two runs from identical state agree.  The kernels use no float
    atomics (test_kernels_are_bit_deterministic); what is left is autograd's summation order of the three gradient
    contributions of a discriminator parameter, which may differ by an fp32 ulp between runs in one process."""
    outs = []
    for _ in range(2):
        sg = build(torch.bfloat16, seed=3)
        losses = one_step(sg, 11)
        sig = [float(p.detach().double().sum()) for p in list(sg.dis.parameters())[:6] + list(sg.gen.parameters())[:6]]
        outs.append(([float(x) for x in losses], sig))
        for p in list(sg.gen.parameters()) + list(sg.dis.parameters()):
            assert torch.isfinite(p).all()
        del sg
        torch.cuda.empty_cache()
    for a, b in zip(outs[0][0] + outs[0][1], outs[1][0] + outs[1][1]):
        assert abs(a - b) <= 1e-5 * abs(a) + 1e-6, outs


def test_kernels_are_bit_deterministic():
    """Same inputs, same bits: forward, data-gradient, weight- and bias-gradient of the MFMA convolutions at 1024x1024."""
    from stylegan.pytorch_amd import functional as F, native
    torch.manual_seed(2)
    w = torch.nn.Parameter(torch.randn(16, 16, 3, 3, device=DEV)); b = torch.nn.Parameter(torch.randn(16, device=DEV))
    x = torch.randn(4, 1024, 1024, 16, device=DEV).bfloat16().requires_grad_(True)
    res = []
    for _ in range(2):
        y = F.conv(x, w, b, "S", 0.05, act=native.ACT_LRELU)
        g = torch.ones_like(y)
        res.append((y.detach().clone(),) + tuple(t.clone() for t in torch.autograd.grad(y, (x, w, b), g)))
    for a, c in zip(*res):
        assert torch.equal(a, c)


def test_bf16_tracks_fp32_at_1024():
    torch.manual_seed(5)
    sg32, sg16 = build(torch.float32, seed=7), build(torch.bfloat16, seed=7)
    sg16.gen.load_state_dict(sg32.gen.state_dict()); sg16.dis.load_state_dict(sg32.dis.state_dict())
    z = torch.randn(2, 512, device=DEV)
    sg32.gen.style_mixing_prob = None; sg16.gen.style_mixing_prob = None
    with torch.no_grad():
        torch.cuda.manual_seed(9); a = sg32.gen(z, 8, 0.5)
        torch.cuda.manual_seed(9); b = sg16.gen(z, 8, 0.5)
        assert a.shape == (2, 3, 1024, 1024)
        rel = float((a - b).double().norm() / a.double().norm())
        assert rel <= 5e-2, rel
        img = torch.randn(4, 3, 1024, 1024, device=DEV)
        sa, sb = sg32.dis(img, 8, 0.5), sg16.dis(img, 8, 0.5)
        assert float((sa - sb).abs().max()) <= 5e-2 * float(sa.abs().max()) + 5e-2


@pytest.mark.parametrize("mode,B,H,ci,co", [("U", 4, 128, 128, 64), ("U", 4, 64, 256, 128), ("U", 4, 256, 64, 32), ("U", 4, 32, 512, 256),
                                            ("U", 4, 16, 512, 512), ("U", 4, 512, 32, 16), ("U", 2, 32, 64, 64), ("U", 4, 8, 64, 64),
                                            ("S", 4, 256, 32, 64), ("S", 4, 256, 32, 128), ("S", 4, 128, 16, 64), ("D", 4, 256, 32, 64),
                                            ("S", 4, 16, 512, 512), ("S", 4, 1024, 16, 16), ("D", 4, 1024, 16, 32)])
def test_bf16_conv_instantiations_against_fp32(mode, B, H, ci, co):
    """Every bf16 tile configuration the 1024x1024 step dispatches to (persistent multi-tile blocks, weights staged once,
    LDS-staged stores, all-parity-class transposed conv, deep-K stages) against the fp32 kernels on the same data."""
    from stylegan.pytorch_amd import functional as F
    torch.manual_seed(1)
    w = torch.nn.Parameter(torch.randn(co, ci, 3, 3, device=DEV))
    x = torch.randn(B, H, H, ci, device=DEV).bfloat16()
    with torch.no_grad():
        y16 = F.conv(x, w, None, mode, 0.05).float()
        y32 = F.conv(x.float(), w, None, mode, 0.05)
    rel = float((y16 - y32).norm() / y32.norm())
    assert rel <= 5e-3, rel                                           # bf16 rounding of the packed weights and the output
