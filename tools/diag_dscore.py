#!/usr/bin/env python3
"""D(real) scores of the bf16 MID discriminator against the fp64 oracle at a batch large enough for the statistic to mean something
(64 scores), with the composed first layer (functional.RGBCONV) on and off.   python tools/diag_dscore.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import golden_util as gu  # noqa: E402
from gpu_util import DEV, MID_DEPTH, build_mid, load_into, mid_params, rel_err  # noqa: E402
from oracle import stylegan_oracle as O  # noqa: E402
from stylegan.pytorch_amd import functional as F  # noqa: E402

torch.set_num_threads(16)
gp, dp = mid_params(torch.float64)
for depth, alpha in ((5, 0.6), (5, 1.0), (4, 0.5)):
    R = 4 << depth
    B = 64
    real = gu.seeded((B, 3, R, R), 65)
    with torch.no_grad():
        ref = O.discriminator(dp, real.double(), depth, alpha, MID_DEPTH)
        out = {}
        for dt in (torch.float32, torch.bfloat16):
            _, dis = build_mid(dt)
            load_into(dis, dp); dis.train()
            for on in (True, False):
                F.RGBCONV = on
                out[(str(dt)[6:], on)] = dis(real.to(DEV), depth, alpha)
    print(f"depth {depth} alpha {alpha} B {B}: |ref| {float(ref.norm()):.3f}  " + "  ".join(f"{k[0]} rgbconv={int(k[1])}: {rel_err(v, ref):.2e}" for k, v in out.items()))
