#!/usr/bin/env python3
"""Golden fixtures at BASELINE's REAL configurations, made by EXECUTING THE REFERENCE on CPU (build container only):

* ``real128.npz``  -- configs/sample_ffhq_128.yaml: 128-model at its true widths (fmap_max 512, 4 mapping layers,
  truncation psi 0.7), depth index 5, batch 4, alpha 0.5: G image / D score and one full G+D iteration (fp32 and fp64).
* ``real1024.npz`` -- configs/sample_ffhq_1024.yaml: 1024-model (8 mapping layers, truncation off), depth index 8,
  batch 2, alpha 0.5: G image / D score and one full G+D iteration (fp32 and fp64).

Weights / noise / inputs are regenerated from seeds (tests/golden_util.py); the fixtures hold outputs only: losses,
per-tensor gradient norms (fp64) and the reference's own fp32 error, small gradient tensors, image statistics, an
8x8-mean-pooled image and a crop.  Same shims as make_golden.py.

    python tests/golden/make_golden_real.py [128] [1024]
"""
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets up the shims and imports the reference)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as gu  # noqa: E402
from models.GAN import StyleGAN  # noqa: E402

CONFIGS = {
    "128": dict(resolution=128, mapping_layers=4, truncation_psi=0.7, depth=5, batch=4),
    "1024": dict(resolution=1024, mapping_layers=8, truncation_psi=-1.0, depth=8, batch=2),
}
ALPHA = 0.5


def image_summary(out, key, img):
    img = img.detach().double()
    out[key + "_stats"] = np.array(gu.tensor_stats(img))
    pool = 8 if img.shape[-1] >= 64 else 1
    out[key + "_pool"] = torch.nn.functional.avg_pool2d(img, pool).float()
    out[key + "_crop"] = img[:, :, :32, :32].float()
    c = img.shape[-1] // 2
    out[key + "_crop_mid"] = img[:, :, c - 16:c + 16, c - 16:c + 16].float()


def build(cfg, dtype):
    torch.manual_seed(0)
    g_args = dict(latent_size=512, mapping_layers=cfg["mapping_layers"], blur_filter=[1, 2, 1],
                  truncation_psi=cfg["truncation_psi"], truncation_cutoff=8)
    d_args = dict(use_wscale=True, blur_filter=[1, 2, 1])
    opt = dict(learning_rate=0.003, beta_1=0.0, beta_2=0.99, eps=1e-8)
    sg = StyleGAN(structure="linear", resolution=cfg["resolution"], num_channels=3, latent_size=512, g_args=g_args,
                  d_args=d_args, g_opt_args=opt, d_opt_args=opt, loss="logistic", d_repeats=1, use_ema=True,
                  ema_decay=0.999, device=torch.device("cpu"))
    if dtype == torch.float64:
        sg.gen.double(); sg.dis.double(); sg.gen_shadow.double()
    MG.fill_module(sg.gen, dtype=dtype); MG.fill_module(sg.dis, dtype=dtype)
    sg.gen_shadow.load_state_dict(sg.gen.state_dict())
    sg.gen.train(); sg.dis.train(); sg.gen_shadow.train()
    MG.pin_noise(sg.gen, cfg["batch"], dtype=dtype)
    return sg


def make(name):
    cfg = CONFIGS[name]
    B, depth, R = cfg["batch"], cfg["depth"], cfg["resolution"]
    out = {"depth": depth, "alpha": ALPHA, "batch": B}
    grads = {}
    for tag, dtype in [("f32", torch.float32), ("f64", torch.float64)]:
        t0 = time.time()
        sg = build(cfg, dtype)
        z = gu.seeded((B, 512), 21, dtype); real = gu.seeded((B, 3, R, R), 22, dtype)
        # ---- forward only (no mixing, the W average restored afterwards)
        with torch.no_grad():
            smp = sg.gen.style_mixing_prob
            sg.gen.style_mixing_prob = None
            avg = sg.gen.truncation.avg_latent.clone() if sg.gen.truncation is not None else None
            img = sg.gen(z, depth, ALPHA)
            if avg is not None:
                sg.gen.truncation.avg_latent.copy_(avg)
            sg.gen.style_mixing_prob = smp
            image_summary(out, f"{tag}_g_img", img)
            out[f"{tag}_d_score"] = sg.dis(real, depth, ALPHA)
            out[f"{tag}_d_score_fake"] = sg.dis(img, depth, ALPHA)
        # ---- one full iteration
        _randn = torch.randn
        if dtype == torch.float64:
            torch.randn = lambda *a, **k: _randn(*a, **k).double()      # latents2 carries the fp32 run's values
        torch.manual_seed(77); random.seed(77)
        out[f"{tag}_d_loss"] = sg.optimize_discriminator(z, real, depth, ALPHA)
        grads[tag, "d"] = {k: p.grad.clone().double() for k, p in sg.dis.named_parameters() if p.grad is not None}
        torch.manual_seed(78); random.seed(78)
        out[f"{tag}_g_loss"] = sg.optimize_generator(z, real, depth, ALPHA)
        torch.randn = _randn
        grads[tag, "g"] = {k: p.grad.clone().double() for k, p in sg.gen.named_parameters() if p.grad is not None}
        if sg.gen.truncation is not None:
            out[f"{tag}_avg_latent"] = sg.gen.truncation.avg_latent
        print(name, tag, "d_loss", out[f"{tag}_d_loss"], "g_loss", out[f"{tag}_g_loss"], f"{time.time() - t0:.1f} s", flush=True)
        del sg
    for net in ("d", "g"):
        names = sorted(grads["f64", net])
        out[f"{net}_grad_names"] = np.array(names)
        out[f"{net}_grad_norm64"] = np.array([float(torch.linalg.vector_norm(grads["f64", net][k])) for k in names])
        out[f"{net}_grad_err32"] = np.array([float(torch.linalg.vector_norm(grads["f32", net][k] - grads["f64", net][k])) for k in names])
        out[f"{net}_grad_stats64"] = np.array([gu.tensor_stats(grads["f64", net][k]) for k in names])
        for k in names:
            if grads["f64", net][k].numel() <= 1024:
                out[f"{net}_grad64::{k}"] = grads["f64", net][k]
    MG.npz(f"real{name}.npz", **out)


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["128", "1024"]):
        make(n)
