"""Pin the oracle's NON-DEFAULT options (layer flags, ReLU, other blur filters) and its label-conditioned path against
fixtures produced by executing the reference (tests/golden/make_golden_flags.py).  CPU only."""
import os
import random

import numpy as np
import torch

import golden_util as gu
from oracle import stylegan_oracle as O

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))

EPI_CASES = {"pn_in": (True, True, True, True, "lrelu"), "bare": (False, False, False, False, "lrelu"),
             "relu": (True, False, True, True, "relu"), "relu_pn": (True, True, False, True, "relu"),
             "noin": (True, False, False, True, "lrelu"), "nostyle": (True, False, True, False, "lrelu")}
BLUR_CASES = {"b5": ([1, 4, 6, 4, 1], True), "b3asym": ([1, 2, 3], True), "b7raw": ([1, 1, 2, 3, 2, 1, 1], False)}
NET = dict(resolution=32, fmap_base=512, fmap_max=32, mapping_layers=2)
NET_DEPTH = 4


def T(a):
    return torch.from_numpy(np.asarray(a)).double()


def close(a, b, tol=1e-9, what=""):
    a = a.detach().double(); b = T(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = torch.linalg.vector_norm(a - b).item(); den = torch.linalg.vector_norm(b).item()
    assert err <= tol * den + 1e-12, f"{what}: rel {err / (den + 1e-30):.3e}"


def module_params(module, prefix="", dtype=torch.float64):
    """Oracle parameter dict of one of OUR modules (same state_dict keys as the reference's), deterministic fill."""
    out = {}
    grads = {k for k, _ in module.named_parameters()}
    for k, v in module.state_dict().items():
        if k.endswith(".kernel"):
            continue
        out[k] = gu.fill_value(prefix + k, v.shape, dtype).requires_grad_(k in grads)
    return out


def test_epilogue_stage_combinations(golden_dir):
    g = np.load(os.path.join(golden_dir, "flags.npz"))
    x, dl, noise, probe = T(g["x"]), T(g["dlat"]), T(g["noise"]), T(g["probe"])
    for name, (un, upn, uin, us, act) in EPI_CASES.items():
        pre = f"fl.{name}."
        nw = gu.fill_value(pre + "top_epi.noise.weight", (16,), torch.float64).requires_grad_(True) if un else None
        sw = gu.fill_value(pre + "style_mod.lin.weight", (32, 512), torch.float64).requires_grad_(True) if us else None
        sb = gu.fill_value(pre + "style_mod.lin.bias", (32,), torch.float64).requires_grad_(True) if us else None
        xi = x.clone().requires_grad_(True)
        y = O.layer_epilogue(xi, noise, nw, sw, sb, dl, O.Flags(un, upn, uin, us, act))
        (y * probe).sum().backward()
        close(y, g[f"epi_{name}_y"], what=name + " y"); close(xi.grad, g[f"epi_{name}_dx"], what=name + " dx")
        if un:
            close(nw.grad, g[f"epi_{name}_g::top_epi.noise.weight"], what=name + " dnw")
        if us:
            close(sw.grad, g[f"epi_{name}_g::style_mod.lin.weight"], what=name + " dsw")
            close(sb.grad, g[f"epi_{name}_g::style_mod.lin.bias"], what=name + " dsb")


def test_blur_filters_and_relu_block(golden_dir):
    g = np.load(os.path.join(golden_dir, "flags.npz"))
    assert int(g["relu_constructs"]) == 0 and int(g["blur_flip_constructs"]) == 0     # dead options of the reference (see the generator)
    for name, (taps, normalize) in BLUR_CASES.items():
        xi = T(g["x"]).requires_grad_(True)
        y = O.blur3(xi, taps, normalize)
        (y * T(g[f"blur_{name}_probe"])).sum().backward()
        close(y, g[f"blur_{name}_y"], what=name); close(xi.grad, g[f"blur_{name}_dx"], what=name + " dx")
    from stylegan.pytorch_amd.Blocks import DiscriminatorBlock
    blk = DiscriminatorBlock(16, 32, gain=np.sqrt(2), use_wscale=True, activation_layer=torch.nn.ReLU(), blur_kernel=[1, 4, 6, 4, 1])
    p = module_params(blk, "fl.dblk.")
    xb = T(g["dblk_x"]).requires_grad_(True)
    yb = O.discriminator_block(p, "", xb, O.Flags(act="relu", blur_taps=(1, 4, 6, 4, 1)))
    (yb * T(g["dblk_probe"])).sum().backward()
    close(yb, g["dblk_y"], what="dblk y"); close(xb.grad, g["dblk_dx"], what="dblk dx")
    for k, v in p.items():
        close(v.grad, g[f"dblk_g::{k}"], what="dblk " + k)


def flag_nets():
    from stylegan.pytorch_amd.GAN import Discriminator, Generator
    gen = Generator(resolution=NET["resolution"], latent_size=512, mapping_layers=NET["mapping_layers"], blur_filter=[1, 4, 6, 4, 1],
                    truncation_psi=0.7, truncation_cutoff=8, fmap_base=NET["fmap_base"], fmap_max=NET["fmap_max"], structure="linear",
                    use_pixel_norm=True, use_noise=False)
    dis = Discriminator(resolution=NET["resolution"], num_channels=3, use_wscale=True, blur_filter=[1, 4, 6, 4, 1],
                        fmap_base=NET["fmap_base"], fmap_max=NET["fmap_max"], structure="linear")
    return gen, dis


FLAGS_NET = O.Flags(use_noise=False, use_pixel_norm=True, blur_taps=(1, 4, 6, 4, 1))


def test_networks_with_flags(golden_dir):
    g = np.load(os.path.join(golden_dir, "flags.npz"))
    gen, dis = flag_nets()
    gp, dp = module_params(gen), module_params(dis)
    B, depth, alpha = 4, 3, 0.4
    z = gu.seeded((B, 512), 11).double()
    img, _ = O.generator(gp, z, depth, alpha, [None] * (2 * NET_DEPTH), mapping_layers=NET["mapping_layers"], num_layers=2 * NET_DEPTH,
                         flags=FLAGS_NET)
    score = O.discriminator(dp, img, depth, alpha, NET_DEPTH, flags=FLAGS_NET)
    close(img, g["net_f64_img"], what="img"); close(score, g["net_f64_score"], what="score")
    score.sum().backward()
    for net, p in (("g", gp), ("d", dp)):
        names = [str(n) for n in g[f"net_{net}_grad_names"]]
        assert sorted(k for k, v in p.items() if v.grad is not None) == names
        for k, n64 in zip(names, g[f"net_{net}_grad_norm64"]):
            assert abs(torch.linalg.vector_norm(p[k].grad).item() - n64) <= 1e-9 * n64 + 1e-12, k
            if f"net_{net}_grad64::{k}" in g:
                close(p[k].grad, g[f"net_{net}_grad64::{k}"], what=k)


def cond_nets(n_classes=5):
    from stylegan.pytorch_amd.GAN import Discriminator, Generator
    gen = Generator(resolution=NET["resolution"], latent_size=512, mapping_layers=NET["mapping_layers"], blur_filter=[1, 2, 1],
                    truncation_psi=0.7, truncation_cutoff=8, fmap_base=NET["fmap_base"], fmap_max=NET["fmap_max"], structure="linear",
                    conditional=True, n_classes=n_classes)
    dis = Discriminator(resolution=NET["resolution"], num_channels=3, use_wscale=True, blur_filter=[1, 2, 1], fmap_base=NET["fmap_base"],
                        fmap_max=NET["fmap_max"], structure="linear", conditional=True, n_classes=n_classes)
    return gen, dis


def test_conditional_step(golden_dir):
    g = np.load(os.path.join(golden_dir, "conditional.npz"))
    gen, dis = cond_nets()
    gp, dp = module_params(gen), module_params(dis)
    B, depth, alpha = 4, 3, 0.5
    labels = torch.from_numpy(g["labels"])
    noises = [gu.seeded((B, 1, 4 * 2 ** (i // 2), 4 * 2 ** (i // 2)), 100 + i, torch.float64) for i in range(2 * NET_DEPTH)]
    z = gu.seeded((B, 512), 21, torch.float64); real = gu.seeded((B, 3, 32, 32), 22, torch.float64)
    kw = dict(total_depth=NET_DEPTH, mapping_layers=NET["mapping_layers"], noises=noises, loss="conditional-loss", labels=labels)
    torch.manual_seed(77); random.seed(77)
    l2, cut = O.draw_mixing((B, 1024), depth)                    # the mixing latents have the CONCATENATED width (models/GAN.py:282)
    d_loss, d_grads = O.d_step(gp, dp, O.AdamState(), z, real, depth, alpha, latents2=l2.double(), mixing_cutoff=cut, **kw)
    torch.manual_seed(78); random.seed(78)
    l2, cut = O.draw_mixing((B, 1024), depth)
    g_loss, g_grads = O.g_step(gp, dp, O.AdamState(), z, depth, alpha, latents2=l2.double(), mixing_cutoff=cut, **kw)
    assert abs(d_loss - float(g["f64_d_loss"])) <= 1e-9 * abs(float(g["f64_d_loss"]))
    assert abs(g_loss - float(g["f64_g_loss"])) <= 1e-9 * abs(float(g["f64_g_loss"]))
    for net, grads in (("d", d_grads), ("g", g_grads)):
        names = [str(n) for n in g[f"{net}_grad_names"]]
        assert sorted(k for k, v in grads.items() if v is not None) == names, net
        for k in names:
            if f"{net}_grad64::{k}" in g and net == "d":           # (G's recorded gradients are pre-clip; g_step returns clipped ones)
                close(grads[k], g[f"{net}_grad64::{k}"], 1e-8, k)
