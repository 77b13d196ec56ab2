#!/usr/bin/env python3
"""Diagnostic: discriminator-step gradients (logistic + R1) of the fp32 HIP path vs the fp64 / fp32 CPU oracle, same fakes."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
torch.set_num_threads(16)
import golden_util as gu
from oracle import stylegan_oracle as O
import test_gpu_realconfigs as T

name = sys.argv[1] if len(sys.argv) > 1 else "1024"
GAMMA = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
cfg = T.CFG[name]
sg, gp, dp = T.make_stylegan(cfg)
B, depth, Rr = cfg["batch"], cfg["depth"], cfg["resolution"]
real = gu.seeded((B, 3, Rr, Rr), 22); fake = gu.seeded((B, 3, Rr, Rr), 23)
from stylegan.pytorch_amd import functional as F
loss = sg.loss.dis_loss(sg.progressive_down_sampling(real.to(T.DEV), depth, T.ALPHA), sg.progressive_down_sampling(fake.to(T.DEV), depth, T.ALPHA), depth, T.ALPHA, r1_gamma=GAMMA)
sg.dis_optim.zero_grad()
with F.accumulate_param_grads():
    loss.backward()
torch.cuda.synchronize()
h = {k: p.grad.detach().double().cpu() for k, p in sg.dis.named_parameters() if p.grad is not None}


def oracle(dtype):
    d2 = {k: v.detach().to(dtype).requires_grad_(True) for k, v in dp.items()}
    r = O.progressive_down_sampling(real.to(dtype), depth, T.ALPHA, cfg["total_depth"]); f = O.progressive_down_sampling(fake.to(dtype), depth, T.ALPHA, cfg["total_depth"])
    l = O.logistic_d_loss(d2, r, f, depth, T.ALPHA, cfg["total_depth"], r1_gamma=GAMMA)
    names = list(d2)
    gl = torch.autograd.grad(l, [d2[k] for k in names], allow_unused=True)
    return float(l), {k: g.double() for k, g in zip(names, gl) if g is not None}


l64, g64 = oracle(torch.float64)
l32, g32 = oracle(torch.float32)
print(f"loss hip {float(loss):.8f} o64 {l64:.8f} o32 {l32:.8f}")
rel = lambda a, b: (torch.linalg.vector_norm(a - b) / (torch.linalg.vector_norm(b) + 1e-30)).item()
print(f"{'tensor':48s} {'hip/o64':>9s} {'o32/o64':>9s} {'signflip hip':>12s} {'signflip o32':>12s}")
for k in g64:
    a, b = h[k], g64[k]
    sf = lambda x: float((torch.sign(x) != torch.sign(b)).double().mean())
    print(f"{k:48s} {rel(a, b):9.2e} {rel(g32[k], b):9.2e} {sf(a):12.2e} {sf(g32[k]):12.2e}")
