"""Pin the CPU oracle (oracle/stylegan_oracle.py) against fixtures produced by executing the
reference itself (tests/golden/make_golden.py).  CPU only."""
import os
import random

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import stylegan_oracle as O

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name), allow_pickle=False).items()}


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def close(a, b, rtol=1e-5, atol=1e-5):
    a = a.detach().double(); b = T(b, torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"max err {err:.3e} vs ref max {ref:.3e}"


def conv_params(prefix, cout, cin, k):
    return gu.fill_value(prefix + "weight", (cout, cin, k, k)), gu.fill_value(prefix + "bias", (cout,))


# ------------------------------------------------------------------ layers
def test_layers(golden_dir):
    g = load(golden_dir, "layers.npz")
    x8 = T(g["plain_x"])
    w, b = conv_params("lay.plain.", 5, 3, 3)
    close(O.eq_conv2d(x8, w, b), g["plain_y"])
    w, b = conv_params("lay.rgb.", 4, 3, 1)
    close(O.eq_conv2d(x8, w, b, gain=1.0), g["rgb_y"])
    w, b = conv_params("lay.up.", 4, 3, 3)
    close(O.eq_conv2d(x8, w, b, up=True, blur_after=True), g["up_nf_y"])
    close(O.eq_conv2d(T(g["up_f_x"]), w, b, up=True, blur_after=True), g["up_f_y"])
    w, b = conv_params("lay.down.", 4, 3, 3)
    close(O.eq_conv2d(T(g["down_nf_x"]), w, b, down=True), g["down_nf_y"])
    close(O.eq_conv2d(T(g["down_f_x"]), w, b, down=True), g["down_f_y"])
    xl = T(g["lin_x"])
    close(O.eq_linear(xl, gu.fill_value("g_mapping.lay.lin.weight", (16, 24)),
                      gu.fill_value("g_mapping.lay.lin.bias", (16,)), gain=O.SQRT2, lrmul=0.01), g["lin_map_y"])
    close(O.eq_linear(xl, gu.fill_value("lay.lin1.weight", (16, 24)), gu.fill_value("lay.lin1.bias", (16,)),
                      gain=1.0), g["lin_g1_y"])
    close(O.layer_epilogue(x8, T(g["epi_noise"]), gu.fill_value("lay.epi.top_epi.noise.weight", (3,)),
                           gu.fill_value("lay.epi.style_mod.lin.weight", (6, 512)),
                           gu.fill_value("lay.epi.style_mod.lin.bias", (6,)), T(g["epi_dlat"])), g["epi_y"])
    close(O.blur3(x8), g["blur_y"])
    close(O.pixel_norm(xl), g["pn_y"])
    close(O.upscale2d(x8), g["up2_y"], 0, 0)
    close(O.downscale2d(x8), g["down2_y"])
    close(O.minibatch_stddev(T(g["std_x"])), g["std_y"])
    close(O.minibatch_stddev(T(g["std_x"])[:2]), g["std2_y"])
    xt = T(g["trunc_x"])
    avg = O.truncation_update(gu.fill_value("truncation.avg_latent", (512,)), xt[0, 0])
    close(avg, g["trunc_avg"])
    close(O.truncation_apply(avg, xt), g["trunc_y"])


def test_fused_equivalences():
    """The algebra the HIP path relies on: fused-down == conv->avgpool, and
    fused-up == nearest-up -> conv with the spatially FLIPPED kernel."""
    torch.manual_seed(0)
    x = torch.randn(2, 3, 16, 16, dtype=torch.float64)
    w = torch.randn(4, 3, 3, 3, dtype=torch.float64)
    import torch.nn.functional as F
    a = F.conv2d(x, O.fused_down_weight(w), stride=2, padding=1)
    b = F.avg_pool2d(F.conv2d(x, w, padding=1), 2)
    assert (a - b).abs().max() < 1e-12
    a = F.conv_transpose2d(x, O.fused_up_weight(w), stride=2, padding=1)
    b = F.conv2d(O.upscale2d(x), w.flip(2, 3), padding=1)
    assert (a - b).abs().max() < 1e-12
    c = F.conv2d(O.upscale2d(x), w, padding=1)
    assert (a - c).abs().max() > 1e-2          # NOT equal un-flipped


# ------------------------------------------------------------------ networks
def tiny_params(dtype=torch.float32):
    gp = O.make_generator_params(gu.TINY["resolution"], gu.TINY["mapping_layers"], 512,
                                 gu.TINY["fmap_base"], gu.TINY["fmap_max"], dtype=dtype)
    dp = O.make_discriminator_params(gu.TINY["resolution"], gu.TINY["fmap_base"], gu.TINY["fmap_max"], dtype=dtype)
    for p in (gp, dp):
        for k in list(p):
            rg = p[k].requires_grad
            p[k] = gu.fill_value(k, p[k].shape, dtype).requires_grad_(rg)
    return gp, dp


def tiny_noises(batch, dtype=torch.float32, seed0=100):
    return [gu.seeded((batch, 1, 4 * 2 ** (i // 2), 4 * 2 ** (i // 2)), seed0 + i, dtype) for i in range(2 * gu.TINY_DEPTH)]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float64, 2e-4)])
def test_networks(golden_dir, dtype, tol):
    g = load(golden_dir, "networks.npz")
    gp, dp = tiny_params(dtype)
    z = T(g["z"], dtype)
    noises = tiny_noises(4, dtype)
    with torch.no_grad():
        close(O.g_mapping(gp, z, gu.TINY["mapping_layers"]), g["map_w"], tol, tol)
        for depth, alpha in [(0, 1), (2, 0.3), (5, 0.7)]:
            gp["truncation.avg_latent"] = gu.fill_value("truncation.avg_latent", (512,), dtype)
            img, avg = O.generator(gp, z, depth, alpha, noises, mapping_layers=gu.TINY["mapping_layers"],
                                   num_layers=2 * gu.TINY_DEPTH)
            close(img, g[f"g_d{depth}_img"], tol, tol)
            close(avg, g[f"g_d{depth}_avg"], tol, tol)
            score = O.discriminator(dp, T(g[f"g_d{depth}_img"], dtype), depth, alpha, gu.TINY_DEPTH)
            close(score, g[f"d_d{depth}_score"], 10 * tol, 10 * tol)
        gp["truncation.avg_latent"] = gu.fill_value("truncation.avg_latent", (512,), dtype)
        torch.manual_seed(1234); random.seed(1234)
        l2, cut = O.draw_mixing(z.shape, 3)
        img, _ = O.generator(gp, z, 3, 0.5, noises, mapping_layers=gu.TINY["mapping_layers"],
                             num_layers=2 * gu.TINY_DEPTH, latents2=l2.to(dtype), mixing_cutoff=cut)
        close(img, g["g_mix_d3_img"], tol, tol)


# ------------------------------------------------------------------ full training iteration
def run_oracle_step(dtype):
    gp, dp = tiny_params(dtype)
    shadow = {k: v.detach().clone() for k, v in gp.items()}
    B, depth, alpha = 4, 5, 0.5
    noises = tiny_noises(B, dtype)
    z = gu.seeded((B, 512), 21, dtype); real = gu.seeded((B, 3, 128, 128), 22, dtype)
    kw = dict(total_depth=gu.TINY_DEPTH, mapping_layers=gu.TINY["mapping_layers"], noises=noises)
    d_opt, g_opt = O.AdamState(), O.AdamState()
    torch.manual_seed(77); random.seed(77)
    l2, cut = O.draw_mixing(z.shape, depth)
    d_loss, d_grads = O.d_step(gp, dp, d_opt, z, real, depth, alpha, latents2=l2.to(dtype), mixing_cutoff=cut, **kw)
    torch.manual_seed(78); random.seed(78)
    l2, cut = O.draw_mixing(z.shape, depth)
    g_loss, g_grads = O.g_step(gp, dp, g_opt, z, depth, alpha, latents2=l2.to(dtype), mixing_cutoff=cut,
                               shadow=shadow, **kw)
    return d_loss, g_loss, d_grads, g_grads, gp, dp, shadow


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f64", torch.float64)])
def test_step(golden_dir, tag, dtype):
    g = load(golden_dir, "step.npz")
    d_loss, g_loss, d_grads, g_grads, gp, dp, shadow = run_oracle_step(dtype)
    rel = 1e-4 if dtype == torch.float32 else 1e-9
    assert abs(d_loss - float(g[f"{tag}_d_loss"])) <= rel * abs(float(g[f"{tag}_d_loss"]))
    assert abs(g_loss - float(g[f"{tag}_g_loss"])) <= rel * abs(float(g[f"{tag}_g_loss"]))
    close(gp["truncation.avg_latent"], g[f"{tag}_avg_latent"], 1e-5, 1e-6)
    for net, grads in (("d", d_grads), ("g", g_grads)):
        names = [str(n) for n in g[f"{tag}_{net}_grad_names"]]
        assert sorted(k for k, v in grads.items() if v is not None) == names      # same active set
        stats = g[f"{tag}_{net}_grad_stats"]
        # analytically-zero gradients (e.g. init_block.bias: a per-channel constant removed by the
        # InstanceNorm) are pure round-off -> absolute floor relative to the network's gradient scale
        net_scale = max(abs(float(v)) for v in stats[:, 2])
        floor = (1e-12 if dtype == torch.float64 else 1e-7) * net_scale
        for k, st in zip(names, stats):
            ref64 = g.get(f"f64_{net}_grad::{k}")
            full = g.get(f"{tag}_{net}_grad::{k}")
            if full is not None:
                a = grads[k].detach().double(); b = T(full, torch.float64)
                # fp32: the reference's own fp32 error vs its fp64 run bounds what can be asked
                scale = T(ref64, torch.float64).abs().max().item() + 1e-30
                ref_err = (T(g[f"f32_{net}_grad::{k}"], torch.float64) - T(ref64, torch.float64)).abs().max().item()
                tol = max(floor, 1e-9 * scale if dtype == torch.float64 else max(1e-3 * scale, 4 * ref_err))
                assert (a - b).abs().max().item() <= tol, (k, (a - b).abs().max().item(), tol)
            else:
                n = gu.tensor_stats(grads[k])[2]
                assert abs(n - st[2]) <= (1e-8 if dtype == torch.float64 else 2e-2) * st[2] + floor, k
    if dtype == torch.float64:
        for net, params in (("d", dp), ("g", gp), ("s", shadow)):
            names = [str(n) for n in g[f"{tag}_{net}_param_names"]]
            for k, st in zip(names, g[f"{tag}_{net}_param_stats"]):
                s = gu.tensor_stats(params[k])
                assert abs(s[2] - st[2]) <= 1e-9 * st[2] + 1e-12, (net, k)
                assert abs(s[0] - st[0]) <= 1e-7 * st[1] + 1e-12, (net, k)


# ------------------------------------------------------------------ schedule (bit exact)
def test_schedule_bit_exact(golden_dir):
    g = load(golden_dir, "schedule.npz")
    for ci in range(2):
        n, start, fb, ck = [int(v) for v in g[f"c{ci}_cfg"]]
        epochs = [int(v) for v in g[f"c{ci}_epochs"]]; bs = [int(v) for v in g[f"c{ci}_bs"]]
        fade = [int(v) for v in g[f"c{ci}_fade"]]
        rows = list(O.schedule(n, epochs, bs, fade, len(epochs), start, fb, ck))
        rec = g[f"c{ci}_rec"]
        assert len(rows) == len(rec)
        marks = []
        for r, (depth, alpha, is_int) in zip(rows, rec):
            assert r[0] == int(depth)
            assert float(r[5]) == float(alpha)                  # bit-exact float
            assert isinstance(r[5], int) == bool(is_int)        # int 1 after the fade point (GAN.py:753)
        for idx, r in enumerate(rows, 1):
            if r[6]:
                marks.append(idx)
            if r[7]:
                marks.append(-idx)
        assert marks == [int(m) for m in g[f"c{ci}_marks"]]
        assert [r[4] for r in rows] == list(range(1, len(rows) + 1))


def test_images_u8_known_values():
    """Input pipeline: ToTensor + Normalize(0.5, 0.5) + horizontal flip on uint8 batches.  torchvision is not
    installed here, so this pins the oracle to the published semantics by known values and by an independent numpy
    evaluation in the same fp32 operation order."""
    u8 = torch.arange(256, dtype=torch.uint8).repeat(3)[: 2 * 4 * 8 * 3].reshape(2, 4, 8, 3)
    out = O.images_u8_to_float(u8)
    assert out.shape == (2, 3, 4, 8) and out.dtype == torch.float32
    lut = O.images_u8_to_float(torch.arange(256, dtype=torch.uint8).reshape(1, 1, 256, 1).expand(1, 1, 256, 3))[0, 0, 0]
    ref = (u8.numpy().astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)
    assert np.array_equal(out.numpy(), ref.transpose(0, 3, 1, 2))
    assert float(lut[0]) == -1.0 and float(lut[255]) == 1.0 and abs(float(lut[128]) - 1.0 / 255.0) < 1e-7
    assert torch.all(lut[1:] > lut[:-1])                                  # strictly monotone: 256 distinct levels
    fl = O.images_u8_to_float(u8, flip=[True, False])
    assert torch.equal(fl[0], out[0].flip(-1)) and torch.equal(fl[1], out[1])


def test_images_u8_against_the_transform_chain_fixture(golden_dir):
    """oracle.images_u8_to_float against tests/golden/images_u8.npz: PNG bytes decoded by PIL and pushed through the
    reference's transform chain (data/transforms.py:27-32) restated operation by operation from torchvision's published
    functional source (make_golden_images.py -- torchvision itself cannot be installed here).  Bit-exact."""
    g = np.load(os.path.join(golden_dir, "images_u8.npz"))
    out = O.images_u8_to_float(torch.from_numpy(g["u8"]), flip=[bool(f) for f in g["flips"]])
    assert out.dtype == torch.float32 and np.array_equal(out.numpy(), g["out"])


def test_other_losses_against_the_reference(golden_dir):
    """StandardGAN / HingeGAN / RelativisticAverageHingeGAN of stylegan.pytorch_amd.Losses (plain torch on the [B,1]
    logits) against values recorded from the reference's own classes (tests/golden/make_golden_losses.py: identity
    discriminator, so the inputs are the prediction vectors).  The reference's StandardGAN.gen_loss cannot run at any
    batch size (models/Losses.py:131 unpacks the output into three values; the fixture records its error): ours is
    BCE(f, 1), checked against the closed form softplus(-f)."""
    from stylegan.pytorch_amd import Losses
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    ident = lambda x, height, alpha: x
    for B in (3, 4, 8):
        r, f = torch.from_numpy(g[f"r_{B}"]), torch.from_numpy(g[f"f_{B}"])
        for name, cls in (("standard", Losses.StandardGAN), ("hinge", Losses.HingeGAN), ("relhinge", Losses.RelativisticAverageHingeGAN)):
            loss = cls(ident)
            want = float(g[f"{name}_dis_{B}"])
            assert abs(float(loss.dis_loss(r, f, 0, 1.0)) - want) <= 1e-6 * max(1.0, abs(want)), (name, B)
            if f"{name}_gen_{B}" in g:
                want = float(g[f"{name}_gen_{B}"])
                assert abs(float(loss.gen_loss(r, f, 0, 1.0)) - want) <= 1e-6 * max(1.0, abs(want)), (name, B)
            else:
                assert "Error" in str(g[f"{name}_gen_{B}_error"]) and name == "standard"
                want = float(torch.nn.functional.softplus(-f.double()).mean())
                assert abs(float(loss.gen_loss(r, f, 0, 1.0)) - want) <= 1e-6


def test_oracle_loss_heads_against_the_reference(golden_dir):
    """oracle.gan_dis_loss / gan_gen_loss (what the GPU parity test of the non-default losses checks against) reproduce the
    values recorded from the reference's own loss classes."""
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    for B in (3, 4, 8):
        r, f = torch.from_numpy(g[f"r_{B}"]).double(), torch.from_numpy(g[f"f_{B}"]).double()
        for name, kind in (("standard", "standard-gan"), ("hinge", "hinge"), ("relhinge", "relativistic-hinge")):
            want = float(g[f"{name}_dis_{B}"])
            assert abs(float(O.gan_dis_loss(kind, r, f)) - want) <= 1e-6 * max(1.0, abs(want)), (name, B)
            if f"{name}_gen_{B}" in g:
                want = float(g[f"{name}_gen_{B}"])
                assert abs(float(O.gan_gen_loss(kind, r, f)) - want) <= 1e-6 * max(1.0, abs(want)), (name, B)


def test_fixed_structure_against_the_reference(golden_dir):
    """structure='fixed' (models/GAN.py:186-190, 408-411): the reference's fixed networks, recorded by
    tests/golden/make_golden_fixed.py, equal the linear networks at the last depth index with alpha = 1 -- in the reference
    itself (recorded difference: exactly 0) and in the oracle.  This is the equivalence the GPU test
    test_fixed_structure_equals_linear_at_full_depth relies on."""
    g = load(golden_dir, "networks_fixed.npz")
    assert float(np.max(g["linear_alpha1_max_abs_diff"])) == 0.0
    gp, dp = tiny_params(torch.float32)
    z = gu.seeded((4, 512), 11); real = gu.seeded((4, 3, 128, 128), 65)
    with torch.no_grad():
        gp["truncation.avg_latent"] = gu.fill_value("truncation.avg_latent", (512,), torch.float32)
        img, _ = O.generator(gp, z, 5, 1.0, tiny_noises(4), mapping_layers=gu.TINY["mapping_layers"], num_layers=2 * gu.TINY_DEPTH)
        close(img[:, :, ::4, ::4], g["g_fixed_img_sub"], 2e-5, 2e-5)
        want = g["g_fixed_img_stats"]
        got = np.array(gu.tensor_stats(img))
        assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-4), (got, want)
        close(O.discriminator(dp, real, 5, 1.0, gu.TINY_DEPTH), g["d_fixed_score"], 2e-4, 2e-4)
