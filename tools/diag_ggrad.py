#!/usr/bin/env python3
"""Diagnostic: where does the generator-gradient error of the fp32 HIP path at 1024x1024 come from?
HIP vs fp64 oracle vs fp32 oracle (CPU): d loss / d image, then per-tensor parameter gradients (scale + residual)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import torch.nn.functional as TF
torch.set_num_threads(16)
import golden_util as gu
from oracle import stylegan_oracle as O
import test_gpu_realconfigs as T

name = sys.argv[1] if len(sys.argv) > 1 else "1024"
cfg = T.CFG[name]
sg, gp, dp = T.make_stylegan(cfg)
B, depth, Rr = cfg["batch"], cfg["depth"], cfg["resolution"]
z = gu.seeded((B, 512), 21)
sg.gen.style_mixing_prob = None
for p in sg.dis.parameters():
    p.requires_grad_(False)
fake = sg.gen(z.to(T.DEV), depth, T.ALPHA)
fake.retain_grad()
loss = TF.softplus(-sg.dis(fake, depth, T.ALPHA)).mean()
loss.backward()
torch.cuda.synchronize()
h_img = fake.grad.detach().double().cpu()
h = {k: p.grad.detach().double().cpu() for k, p in sg.gen.named_parameters() if p.grad is not None}


def oracle(dtype):
    g2 = {k: v.detach().to(dtype).requires_grad_(v.requires_grad) for k, v in gp.items()}
    d2 = {k: v.detach().to(dtype) for k, v in dp.items()}
    ns = [n.to(dtype) for n in T.noises(cfg)]
    f, _ = O.generator(g2, z.to(dtype), depth, T.ALPHA, ns, mapping_layers=cfg["mapping_layers"], num_layers=2 * cfg["total_depth"],
                       truncation_psi=cfg["psi"])
    f.retain_grad()
    l = TF.softplus(-O.discriminator(d2, f, depth, T.ALPHA, cfg["total_depth"])).mean()
    names = [k for k, v in g2.items() if v.requires_grad]
    gl = torch.autograd.grad(l, [g2[k] for k in names] + [f], allow_unused=True)
    return float(l), gl[-1].double(), {k: g.double() for k, g in zip(names, gl[:-1]) if g is not None}


l64, img64, g64 = oracle(torch.float64)
l32, img32, g32 = oracle(torch.float32)
print(f"loss hip {float(loss):.8f} o64 {l64:.8f} o32 {l32:.8f}")
rel = lambda a, b: (torch.linalg.vector_norm(a - b) / (torch.linalg.vector_norm(b) + 1e-30)).item()
print(f"d loss/d image: hip vs o64 {rel(h_img, img64):.2e}   o32 vs o64 {rel(img32, img64):.2e}")
print(f"{'tensor':60s} {'hip/o64':>9s} {'o32/o64':>9s} {'scale-1':>9s} {'resid':>9s}")
for k in g64:
    a, b = h[k], g64[k]
    s = (a * b).sum() / (b * b).sum()
    print(f"{k:60s} {rel(a, b):9.2e} {rel(g32[k], b):9.2e} {float(s) - 1:9.2e} {rel(a / s, b):9.2e}")
