#!/usr/bin/env python3
"""Fixtures for the inference path by EXECUTING THE REFERENCE on CPU: the Generator used exactly the way
the reference's generate scripts use it, on the MID network (128x128, channel counts the HIP kernels take):
  * generate_samples.py:99-110        ``gen(point, depth=out_depth, alpha=1)`` (the script never calls eval(): train-mode
                                      forward under no_grad, i.e. with the style-mixing draw and the W-average update)
  * generate_mixing_figure.py:17-25,38-43  ``g_mapping`` -> row dlatents with a style range replaced -> ``g_synthesis``
  * generate_truncation_figure.py:22-34    ``g_mapping``, ``truncation.avg_latent``, psi sweep -> ``g_synthesis``
Noise is pinned per layer; weights come from tests/golden_util.py.  Writes inference_mid.npz (images 2x2-mean-pooled, fp16).

    python tests/golden/make_golden_inference.py
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as gu  # noqa: E402
from models.GAN import Generator  # noqa: E402

pool = lambda img: torch.nn.functional.avg_pool2d(img, 2).to(torch.float16)
out = {}
with torch.no_grad():
    gen = MG.fill_module(Generator(**MG.GM_KW))
    out_depth = int(np.log2(128)) - 2
    latent_size = 512
    # ---- generate_samples.py
    MG.pin_noise(gen, 1)
    torch.manual_seed(11); random.seed(11)
    point = torch.randn(1, latent_size)
    point = (point / point.norm()) * (latent_size ** 0.5)
    out["sample_point"] = point
    out["sample_img"] = pool(gen(point, depth=out_depth, alpha=1))
    out["sample_avg_after"] = gen.truncation.avg_latent.clone()
    # ---- generate_mixing_figure.py
    src_seeds, dst_seeds, style_ranges = [639, 701], [888, 829], [range(0, 4), range(4, 8)]
    src = torch.from_numpy(np.stack([np.random.RandomState(s).randn(latent_size) for s in src_seeds]).astype(np.float32))
    dst = torch.from_numpy(np.stack([np.random.RandomState(s).randn(latent_size) for s in dst_seeds]).astype(np.float32))
    MG.pin_noise(gen, 2)
    src_dl, dst_dl = gen.g_mapping(src), gen.g_mapping(dst)
    out["mix_src_dlat0"] = src_dl[:, 0]
    out["mix_src_img"] = pool(gen.g_synthesis(src_dl, depth=out_depth, alpha=1))
    out["mix_dst_img"] = pool(gen.g_synthesis(dst_dl, depth=out_depth, alpha=1))
    for row in range(2):
        row_dl = np.stack([dst_dl.numpy()[row]] * 2)
        row_dl[:, style_ranges[row]] = src_dl.numpy()[:, style_ranges[row]]
        out[f"mix_row{row}_img"] = pool(gen.g_synthesis(torch.from_numpy(row_dl), depth=out_depth, alpha=1))
    # ---- generate_truncation_figure.py
    seeds, psis = [91, 388], [1, 0.5, -0.5]
    lat = torch.from_numpy(np.stack([np.random.RandomState(s).randn(latent_size) for s in seeds]).astype(np.float32))
    dl = gen.g_mapping(lat).detach().numpy()
    avg = gen.truncation.avg_latent.numpy()
    MG.pin_noise(gen, 3)
    for row, d in enumerate(list(dl)):
        row_dl = (d[np.newaxis] - avg) * np.reshape(psis, [-1, 1, 1]) + avg
        out[f"trunc_row{row}_img"] = pool(gen.g_synthesis(torch.from_numpy(row_dl.astype(np.float32)), depth=out_depth, alpha=1))
MG.npz("inference_mid.npz", **out)
