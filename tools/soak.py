#!/usr/bin/env python3
"""Soak test: N iterations of the 1024x1024 step with a fade-in alpha that changes every iteration, alternating launch
modes (eager / hipGraph replay) every 25 iterations; every loss and, at the end, every parameter must be finite.
    python tools/soak.py [iterations=120]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylegan.pytorch_amd.GAN import StyleGAN

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda:0")
torch.manual_seed(0); random.seed(0)
opt = dict(learning_rate=0.003, beta_1=0, beta_2=0.99, eps=1e-8)
sg = StyleGAN("linear", 1024, 3, 512, g_args=dict(latent_size=512, mapping_layers=8, blur_filter=[1, 2, 1], truncation_psi=0.7, truncation_cutoff=8),
              d_args=dict(use_wscale=True, blur_filter=[1, 2, 1]), g_opt_args=opt, d_opt_args=opt, loss="logistic", use_ema=True,
              device=dev, act_dtype=torch.bfloat16, use_graphs=True)
gen = torch.Generator(device=dev); gen.manual_seed(1)
losses = []
t0 = time.perf_counter()
for i in range(n):
    sg.use_graphs = (i // 25) % 2 == 0
    z = torch.randn(4, 512, device=dev, generator=gen)
    x = torch.randn(4, 1024, 1024, 3, device=dev, generator=gen).permute(0, 3, 1, 2)
    alpha = min(1.0, (i + 1) / (0.75 * n))
    losses.append((sg.optimize_discriminator(z, x, 8, alpha), sg.optimize_generator(z, x, 8, alpha)))
vals = [(float(d), float(g)) for d, g in losses]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
bad = [i for i, (d, g) in enumerate(vals) if not (abs(d) < 1e6 and abs(g) < 1e6)]
nonfinite = [k for k, p in list(sg.gen.named_parameters()) + list(sg.dis.named_parameters()) + list(sg.gen_shadow.named_parameters()) if not torch.isfinite(p).all()]
print(f"{n} iterations in {dt:.2f} s ({dt / n * 1e3:.1f} ms/it incl. synthetic data); first {vals[0]}, last {vals[-1]}; bad losses {bad[:5]}; non-finite params {nonfinite[:5]}")
assert not bad and not nonfinite
print("soak ok; graphs captured:", {k[0]: (g.graph is not None, g.calls) for k, g in sg._step_graphs.items()})
