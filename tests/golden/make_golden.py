#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by EXECUTING THE REFERENCE on CPU.

Runs only in the build container (needs /root/reference; never on the GPU box, never in tests).
Nothing from the reference is copied: its modules are imported in-process, driven with
deterministic inputs, and only inputs/outputs are written (as .npz) next to this script.

Shims: a stub ``data`` module (models/GAN.py:25 imports torchvision through it),
``sys.dont_write_bytecode`` (do not litter the read-only mount), float Adam betas (config.py:81
gives an int that torch 2.10 rejects).

    python tests/golden/make_golden.py
"""
import os
import random
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/
sys.path.insert(0, "/root/reference")
_d = types.ModuleType("data"); _d.get_data_loader = None; sys.modules["data"] = _d

import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as gu  # noqa: E402
from models import CustomLayers as CL  # noqa: E402
from models.GAN import Discriminator, Generator, StyleGAN  # noqa: E402

torch.set_num_threads(8)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, len(out), "arrays")


def fill_module(m, prefix="", dtype=torch.float32):
    sd = m.state_dict()
    new = {}
    for k, v in sd.items():
        new[k] = v if k.endswith(".kernel") else gu.fill_value(prefix + k, v.shape, dtype)
    m.load_state_dict(new)
    return m


# ------------------------------------------------------------------ layer fixtures
def layers():
    out = {}
    x8 = gu.seeded((2, 3, 8, 8), 1)
    # plain 3x3 and 1x1
    c = fill_module(CL.EqualizedConv2d(3, 5, 3, use_wscale=True), "lay.plain.")
    out["plain_x"], out["plain_y"] = x8, c(x8)
    c = fill_module(CL.EqualizedConv2d(3, 4, 1, gain=1, use_wscale=True), "lay.rgb.")
    out["rgb_y"] = c(x8)
    # up, non-fused (8 -> 16) with blur
    c = fill_module(CL.EqualizedConv2d(3, 4, 3, use_wscale=True, intermediate=CL.BlurLayer([1, 2, 1]), upscale=True), "lay.up.")
    out["up_nf_y"] = c(x8)
    # up, fused (64 -> 128) with blur, same weights
    x64 = gu.seeded((1, 3, 64, 64), 2)
    out["up_f_x"], out["up_f_y"] = x64, c(x64)
    # down, non-fused (16 -> 8) and fused (128 -> 64)
    c = fill_module(CL.EqualizedConv2d(3, 4, 3, use_wscale=True, downscale=True), "lay.down.")
    x16 = gu.seeded((2, 3, 16, 16), 3)
    out["down_nf_x"], out["down_nf_y"] = x16, c(x16)
    x128 = gu.seeded((1, 3, 128, 128), 4)
    out["down_f_x"], out["down_f_y"] = x128, c(x128)
    # linear (mapping-style lrmul and plain)
    xl = gu.seeded((4, 24), 5)
    l = fill_module(CL.EqualizedLinear(24, 16, use_wscale=True, lrmul=0.01), "g_mapping.lay.lin.")
    out["lin_x"], out["lin_map_y"] = xl, l(xl)
    l = fill_module(CL.EqualizedLinear(24, 16, gain=1.0, use_wscale=True), "lay.lin1.")
    out["lin_g1_y"] = l(xl)
    # epilogue
    epi = CL.LayerEpilogue(3, 512, True, True, False, True, True, torch.nn.LeakyReLU(0.2))
    fill_module(epi, "lay.epi.")
    noise = gu.seeded((2, 1, 8, 8), 6)
    epi.top_epi.noise.noise = noise
    dl = gu.seeded((2, 512), 7)
    out["epi_noise"], out["epi_dlat"], out["epi_y"] = noise, dl, epi(x8, dl)
    # blur, pixel norm, up/down scale, stddev, truncation
    out["blur_y"] = CL.BlurLayer([1, 2, 1])(x8)
    out["pn_y"] = CL.PixelNormLayer()(xl)
    out["up2_y"] = CL.Upscale2d()(x8)
    out["down2_y"] = CL.Downscale2d()(x8)
    xs = gu.seeded((8, 5, 4, 4), 8)
    out["std_x"], out["std_y"] = xs, CL.StddevLayer(4, 1)(xs)
    out["std2_y"] = CL.StddevLayer(4, 1)(xs[:2])
    tr = CL.Truncation(gu.fill_value("truncation.avg_latent", (512,)), max_layer=8, threshold=0.7, beta=0.995)
    xt = gu.seeded((2, 12, 512), 9)
    tr.update(xt[0, 0])
    out["trunc_x"], out["trunc_avg"], out["trunc_y"] = xt, tr.avg_latent.clone(), tr(xt)
    npz("layers.npz", **out)


# ------------------------------------------------------------------ network fixtures
G_KW = dict(resolution=gu.TINY["resolution"], latent_size=512, mapping_layers=gu.TINY["mapping_layers"],
            blur_filter=[1, 2, 1], truncation_psi=0.7, truncation_cutoff=8,
            fmap_base=gu.TINY["fmap_base"], fmap_max=gu.TINY["fmap_max"], structure="linear")
D_KW = dict(resolution=gu.TINY["resolution"], num_channels=3, use_wscale=True, blur_filter=[1, 2, 1],
            fmap_base=gu.TINY["fmap_base"], fmap_max=gu.TINY["fmap_max"], structure="linear")


def pin_noise(gen, batch, seed0=100, dtype=torch.float32):
    mods = [m for m in gen.modules() if isinstance(m, CL.NoiseLayer)]
    for i, m in enumerate(mods):                      # module order == execution order (2 per resolution)
        r = 4 * 2 ** (i // 2)
        m.noise = gu.seeded((batch, 1, r, r), seed0 + i, dtype)
    return mods


def networks():
    out = {}
    B = 4
    gen = fill_module(Generator(**G_KW))
    dis = fill_module(Discriminator(**D_KW))
    gen.train(); dis.train()
    pin_noise(gen, B)
    z = gu.seeded((B, 512), 11)
    out["z"] = z
    for depth, alpha in [(0, 1), (2, 0.3), (5, 0.7)]:
        # mapping+synthesis without mixing (deterministic), truncation active, avg buffer reset each time
        gen.style_mixing_prob = None
        gen.truncation.avg_latent.copy_(gu.fill_value("truncation.avg_latent", (512,)))
        img = gen(z, depth, alpha)
        out[f"g_d{depth}_img"] = img
        out[f"g_d{depth}_avg"] = gen.truncation.avg_latent.clone()
        out[f"d_d{depth}_score"] = dis(img.detach(), depth, alpha)
    # style mixing with pinned RNG (reference order: CPU randn, random.random, random.randint)
    gen.style_mixing_prob = 0.9
    gen.truncation.avg_latent.copy_(gu.fill_value("truncation.avg_latent", (512,)))
    torch.manual_seed(1234); random.seed(1234)
    out["g_mix_d3_img"] = gen(z, 3, 0.5)
    # mapping only
    out["map_w"] = gen.g_mapping(z)[:, 0]
    npz("networks.npz", **out)


# ------------------------------------------------------------------ one full training iteration
def step():
    out = {}
    B, depth, alpha = 4, 5, 0.5
    for tag, dtype in [("f32", torch.float32), ("f64", torch.float64)]:
        torch.manual_seed(0)
        sg = StyleGAN(structure="linear", resolution=128, num_channels=3, latent_size=512,
                      g_args={k: v for k, v in G_KW.items() if k not in ("resolution", "structure")},
                      d_args={k: v for k, v in D_KW.items() if k not in ("resolution", "structure", "num_channels")},
                      g_opt_args=dict(learning_rate=0.003, beta_1=0.0, beta_2=0.99, eps=1e-8),
                      d_opt_args=dict(learning_rate=0.003, beta_1=0.0, beta_2=0.99, eps=1e-8),
                      loss="logistic", d_repeats=1, use_ema=True, ema_decay=0.999, device=torch.device("cpu"))
        if dtype == torch.float64:
            sg.gen.double(); sg.dis.double(); sg.gen_shadow.double()
        fill_module(sg.gen, dtype=dtype); fill_module(sg.dis, dtype=dtype)
        sg.gen_shadow.load_state_dict(sg.gen.state_dict())
        sg.gen.train(); sg.dis.train(); sg.gen_shadow.train()
        pin_noise(sg.gen, B, dtype=dtype)
        z = gu.seeded((B, 512), 21, dtype)
        real = gu.seeded((B, 3, 128, 128), 22, dtype)
        torch.manual_seed(77); random.seed(77)
        _randn = torch.randn
        if dtype == torch.float64:
            # latents2 = torch.randn(shape) (models/GAN.py:282) must carry the SAME values as in the
            # fp32 run: draw in fp32, then widen.
            torch.randn = lambda *a, **k: _randn(*a, **k).double()
        d_loss = sg.optimize_discriminator(z, real, depth, alpha)
        d_grads = {k: p.grad.clone() for k, p in sg.dis.named_parameters() if p.grad is not None}
        torch.manual_seed(78); random.seed(78)
        g_loss = sg.optimize_generator(z, real, depth, alpha)
        torch.randn = _randn
        g_grads = {k: p.grad.clone() for k, p in sg.gen.named_parameters() if p.grad is not None}
        out[f"{tag}_d_loss"], out[f"{tag}_g_loss"] = d_loss, g_loss
        out[f"{tag}_avg_latent"] = sg.gen.truncation.avg_latent
        for net, grads in (("d", d_grads), ("g", g_grads)):
            names = sorted(grads)
            out[f"{tag}_{net}_grad_names"] = np.array(names)
            out[f"{tag}_{net}_grad_stats"] = np.array([gu.tensor_stats(grads[k]) for k in names])
            for k in names:                            # full tensors for everything small
                if grads[k].numel() <= 4096:
                    out[f"{tag}_{net}_grad::{k}"] = grads[k]
        # post-step parameters: checksums only (Adam at beta1=0 gives +-lr steps)
        for net, mod in (("d", sg.dis), ("g", sg.gen), ("s", sg.gen_shadow)):
            names = sorted(dict(mod.named_parameters()))
            sd = dict(mod.named_parameters())
            out[f"{tag}_{net}_param_names"] = np.array(names)
            out[f"{tag}_{net}_param_stats"] = np.array([gu.tensor_stats(sd[k]) for k in names])
    out["z"] = gu.seeded((B, 512), 21); out["depth"] = depth; out["alpha"] = alpha
    npz("step.npz", **out)


# ------------------------------------------------------------------ "mid" fixtures: channel counts the HIP kernels take
# (multiples of 16), so the GPU path can be checked against reference OUTPUTS directly, not only via the oracle.
MID = dict(resolution=128, fmap_base=1024, fmap_max=32, mapping_layers=2)
GM_KW = dict(resolution=128, latent_size=512, mapping_layers=2, blur_filter=[1, 2, 1], truncation_psi=0.7,
             truncation_cutoff=8, fmap_base=1024, fmap_max=32, structure="linear")
DM_KW = dict(resolution=128, num_channels=3, use_wscale=True, blur_filter=[1, 2, 1], fmap_base=1024, fmap_max=32,
             structure="linear")


def networks_mid():
    out = {}
    B = 4
    gen = fill_module(Generator(**GM_KW)); dis = fill_module(Discriminator(**DM_KW))
    gen.train(); dis.train()
    pin_noise(gen, B)
    z = gu.seeded((B, 512), 11)
    gen.style_mixing_prob = None
    for depth, alpha in [(0, 1), (3, 0.25), (5, 0.6)]:
        gen.truncation.avg_latent.copy_(gu.fill_value("truncation.avg_latent", (512,)))
        img = gen(z, depth, alpha)
        out[f"g_d{depth}_img"] = img.to(torch.float16) if depth == 5 else img       # 128x128: store compactly
        out[f"g_d{depth}_img_stats"] = np.array(gu.tensor_stats(img))
        real = gu.seeded((B, 3, 4 * 2 ** depth, 4 * 2 ** depth), 60 + depth)
        out[f"d_d{depth}_score"] = dis(real, depth, alpha)
    npz("networks_mid.npz", **out)


def step_mid():
    """Losses and per-tensor gradient norms of one reference iteration (fp32 and fp64) on the mid networks:
    ||g64||, ||g32 - g64|| per tensor -- the yardstick for the HIP path's gradient error."""
    out = {}
    B, depth, alpha = 4, 5, 0.5
    grads = {}
    for tag, dtype in [("f32", torch.float32), ("f64", torch.float64)]:
        torch.manual_seed(0)
        sg = StyleGAN(structure="linear", resolution=128, num_channels=3, latent_size=512,
                      g_args={k: v for k, v in GM_KW.items() if k not in ("resolution", "structure")},
                      d_args={k: v for k, v in DM_KW.items() if k not in ("resolution", "structure", "num_channels")},
                      g_opt_args=dict(learning_rate=0.003, beta_1=0.0, beta_2=0.99, eps=1e-8),
                      d_opt_args=dict(learning_rate=0.003, beta_1=0.0, beta_2=0.99, eps=1e-8),
                      loss="logistic", d_repeats=1, use_ema=True, ema_decay=0.999, device=torch.device("cpu"))
        if dtype == torch.float64:
            sg.gen.double(); sg.dis.double(); sg.gen_shadow.double()
        fill_module(sg.gen, dtype=dtype); fill_module(sg.dis, dtype=dtype)
        sg.gen_shadow.load_state_dict(sg.gen.state_dict())
        sg.gen.train(); sg.dis.train(); sg.gen_shadow.train()
        pin_noise(sg.gen, B, dtype=dtype)
        z = gu.seeded((B, 512), 21, dtype); real = gu.seeded((B, 3, 128, 128), 22, dtype)
        _randn = torch.randn
        if dtype == torch.float64:
            torch.randn = lambda *a, **k: _randn(*a, **k).double()
        torch.manual_seed(77); random.seed(77)
        out[f"{tag}_d_loss"] = sg.optimize_discriminator(z, real, depth, alpha)
        grads[tag, "d"] = {k: p.grad.clone().double() for k, p in sg.dis.named_parameters() if p.grad is not None}
        torch.manual_seed(78); random.seed(78)
        out[f"{tag}_g_loss"] = sg.optimize_generator(z, real, depth, alpha)
        torch.randn = _randn
        grads[tag, "g"] = {k: p.grad.clone().double() for k, p in sg.gen.named_parameters() if p.grad is not None}
    for net in ("d", "g"):
        names = sorted(grads["f64", net])
        out[f"{net}_grad_names"] = np.array(names)
        out[f"{net}_grad_norm64"] = np.array([float(torch.linalg.vector_norm(grads["f64", net][k])) for k in names])
        out[f"{net}_grad_err32"] = np.array([float(torch.linalg.vector_norm(grads["f32", net][k] - grads["f64", net][k]))
                                             for k in names])
        for k in names:
            if grads["f64", net][k].numel() <= 1024:
                out[f"{net}_grad64::{k}"] = grads["f64", net][k]
    npz("step_mid.npz", **out)


# ------------------------------------------------------------------ schedule (bit-exact bookkeeping)
def schedule():
    """Replays the loop bookkeeping of StyleGAN.train (models/GAN.py:730-803) by running the
    reference's own ``train`` with the compute methods stubbed out."""
    import logging
    rec = []

    class FakeData(list):
        pass

    def run(num_images, epochs, batch_sizes, fade, start_depth, feedback_factor, checkpoint_factor):
        sg = StyleGAN.__new__(StyleGAN)
        sg.depth = len(epochs); sg.structure = "linear"; sg.use_ema = False; sg.conditional = False
        sg.latent_size = 8; sg.device = torch.device("cpu"); sg.n_classes = 0
        sg.gen = torch.nn.Linear(1, 1); sg.dis = torch.nn.Linear(1, 1)
        sg.gen_optim = torch.optim.SGD(sg.gen.parameters(), lr=0.1)
        sg.dis_optim = torch.optim.SGD(sg.dis.parameters(), lr=0.1)
        cur = {}

        def fake_loader(dataset, batch_size, num_workers):
            return [torch.zeros(batch_size, 1)] * (num_images // batch_size)

        import models.GAN as G
        G.get_data_loader = fake_loader

        def od(noise, images, depth, alpha, labels=None):
            cur["alpha"] = alpha; cur["depth"] = depth
            return 0.0

        def og(noise, images, depth, alpha, labels=None):
            rec.append((depth, float(alpha), int(isinstance(alpha, int))))
            return 0.0
        sg.optimize_discriminator = od; sg.optimize_generator = og
        sg.create_grid = lambda **kw: None
        fb = []
        log = logging.getLogger("golden"); log.handlers = []; log.propagate = False

        class H(logging.Handler):
            def emit(self, r):
                m = r.getMessage()
                if m.startswith("Elapsed"):
                    fb.append(len(rec))
                if m.startswith("Saving the model to") and "GAN_GEN_" in m and "SHADOW" not in m:
                    fb.append(-len(rec))
        log.addHandler(H()); log.setLevel(logging.INFO)
        import tempfile
        gen_orig = sg.gen
        sg.gen = lambda *a, **k: torch.zeros(1)      # sample-grid forward stub
        sg.gen.train = lambda: None
        sg.gen.state_dict = gen_orig.state_dict
        with tempfile.TemporaryDirectory() as td:
            sg.train(None, 0, epochs, batch_sizes, fade, log, td, num_samples=1, start_depth=start_depth,
                     feedback_factor=feedback_factor, checkpoint_factor=checkpoint_factor)
        return fb

    cases = [
        dict(num_images=97, epochs=[2, 3, 2], batch_sizes=[16, 8, 4], fade=[50, 50, 50], start_depth=0,
             feedback_factor=10, checkpoint_factor=2),
        dict(num_images=1000, epochs=[1, 2, 4], batch_sizes=[128, 64, 32], fade=[50, 30, 75], start_depth=1,
             feedback_factor=4, checkpoint_factor=3),
    ]
    out = {}
    for ci, c in enumerate(cases):
        rec.clear()
        marks = run(**c)
        out[f"c{ci}_rec"] = np.array(rec, dtype=np.float64)
        out[f"c{ci}_marks"] = np.array(marks, dtype=np.int64)
        out[f"c{ci}_cfg"] = np.array([c["num_images"], c["start_depth"], c["feedback_factor"], c["checkpoint_factor"]])
        out[f"c{ci}_epochs"] = np.array(c["epochs"]); out[f"c{ci}_bs"] = np.array(c["batch_sizes"])
        out[f"c{ci}_fade"] = np.array(c["fade"])
    npz("schedule.npz", **out)


if __name__ == "__main__":
    with torch.no_grad():
        layers()
        networks()
        networks_mid()
    step()
    step_mid()
    schedule()
