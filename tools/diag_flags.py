#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient error of the discriminator alone (fixed input image) against the fp64 oracle, for the
default blur and a 5-tap blur, at a fade-in depth.  usage (GPU box): python tools/diag_flags.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402
from oracle import stylegan_oracle as O  # noqa: E402
from stylegan.pytorch_amd.GAN import Discriminator  # noqa: E402

DEV = "cuda:0"
B, depth, alpha, total = 4, 3, 0.4, 4
for taps in ([1, 2, 1], [1, 4, 6, 4, 1]):
    dis = Discriminator(resolution=32, num_channels=3, use_wscale=True, blur_filter=taps, fmap_base=512, fmap_max=32, structure="linear")
    sd = {k: (v if k.endswith(".kernel") else gu.fill_value(k, v.shape)) for k, v in dis.state_dict().items()}
    dis.load_state_dict(sd); dis.to(DEV).train()
    dp = {k: gu.fill_value(k, v.shape, torch.float64).requires_grad_(True) for k, v in dis.state_dict().items() if not k.endswith(".kernel")}
    img = gu.seeded((B, 3, 32, 32), 5)
    for dtype in (torch.float64, torch.float32):
        p = {k: v.detach().to(dtype).requires_grad_(True) for k, v in dp.items()}
        s = O.discriminator(p, img.to(dtype), depth, alpha, total, flags=O.Flags(blur_taps=taps))
        s.sum().backward()
        if dtype == torch.float64:
            ref, sref = {k: v.grad for k, v in p.items() if v.grad is not None}, s.detach()
        else:
            r32 = {k: v.grad.double() for k, v in p.items() if v.grad is not None}
    score = dis(img.to(DEV), depth, alpha)
    score.sum().backward()
    print(f"== blur {taps}: score rel err {float((score.detach().double().cpu() - sref).norm() / sref.norm()):.2e}")
    for k, q in dis.named_parameters():
        if q.grad is None:
            continue
        n = ref[k].norm().item()
        print(f"  {k:32s} ours {float((q.grad.double().cpu() - ref[k]).norm()) / n:.2e}   oracle-fp32 {float((r32[k] - ref[k]).norm()) / n:.2e}")
