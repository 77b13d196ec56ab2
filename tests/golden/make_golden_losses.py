#!/usr/bin/env python3
"""Golden values of the reference's non-default GAN losses (models/Losses.py:96-189), by EXECUTING THE REFERENCE on CPU.

Build container only (needs /root/reference).  The reference loss classes are driven with an identity "discriminator"
(dis(x, height, alpha) = x), so the inputs ARE the prediction vectors and only the loss arithmetic is recorded.
Writes losses.npz next to this script.

    python tests/golden/make_golden_losses.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
_d = types.ModuleType("data"); _d.get_data_loader = None; sys.modules["data"] = _d

import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as gu  # noqa: E402
from models import Losses as RL  # noqa: E402


def identity_dis(x, height, alpha):
    return x


out = {}
for B in (3, 4, 8):
    r = 1.5 * gu.seeded((B, 1), 300 + B); f = 1.5 * gu.seeded((B, 1), 400 + B)
    out[f"r_{B}"], out[f"f_{B}"] = r.numpy(), f.numpy()
    for name, cls in (("standard", RL.StandardGAN), ("hinge", RL.HingeGAN), ("relhinge", RL.RelativisticAverageHingeGAN)):
        loss = cls(identity_dis)
        out[f"{name}_dis_{B}"] = float(loss.dis_loss(r, f, 0, 1.0))
        try:
            out[f"{name}_gen_{B}"] = float(loss.gen_loss(r, f, 0, 1.0))
        except Exception as e:                               # StandardGAN.gen_loss unpacks the [B,1] output into 3 values (:131)
            out[f"{name}_gen_{B}_error"] = f"{type(e).__name__}: {e}"[:200]
np.savez_compressed(os.path.join(HERE, "losses.npz"), **{k: np.asarray(v) for k, v in out.items()})
for k, v in out.items():
    if not k.startswith(("r_", "f_")):
        print(k, v)
