#!/usr/bin/env python3
"""The ANCHOR of the bf16 gradient bars (tests/test_gpu_bf16.py): what a NAIVE bf16 cast of the REFERENCE itself gives.

The reference 128-model (configs/sample_ffhq_128.yaml: fmap_max 512, 4 mapping layers, psi 0.7; depth index 5, batch 4, alpha 0.5 --
the configuration of real128.npz) is executed on CPU three ways with identical weights, noise, latents and seeds:

  f64  -- the truth (``.double()``);
  bf16 -- ``.bfloat16()`` on generator and discriminator: every parameter, activation, accumulator-visible tensor and gradient in bf16
          (the "whole-model cast" of SURVEY.md 8c, now for gradients too);
  ac   -- ``torch.autocast("cpu", dtype=torch.bfloat16)`` around both half-iterations: fp32 parameters, bf16 convolutions / linears with
          fp32 accumulation -- the mixed-precision mode a user of the reference would actually train in.

As in tests/test_gpu_realconfigs.decoupled_step the generator half runs on the f64 run's UPDATED discriminator (Adam at beta1 = 0 turns
every rounding-flipped sign of a near-zero D gradient into a +-lr weight difference; coupled, the G gradients would measure that chaos).

Written: per-tensor rel-L2 and 1 - cosine of every parameter gradient against the f64 run, the loss errors, and the per-network medians.
Only numbers derived from the reference's outputs are stored -- no reference code.

    python tests/golden/make_golden_bf16_anchor.py        (build container only: needs /root/reference; ~10 min of CPU)
"""
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (shims + reference import)
import make_golden_real as MR  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as gu  # noqa: E402

ALPHA = 0.5


def run(cfg, mode, d_after=None):
    """One decoupled D+G iteration of the reference in ``mode``; -> (d_loss, g_loss, d_grads, g_grads, updated D state)."""
    B, depth, R = cfg["batch"], cfg["depth"], cfg["resolution"]
    dtype = {"f64": torch.float64, "bf16": torch.float32, "ac": torch.float32}[mode]
    sg = MR.build(cfg, dtype)
    z = gu.seeded((B, 512), 21, dtype); real = gu.seeded((B, 3, R, R), 22, dtype)
    _randn = torch.randn
    if mode == "bf16":
        sg.gen.bfloat16(); sg.dis.bfloat16(); sg.gen_shadow.bfloat16()
        for m in sg.gen.modules():
            if isinstance(m, MG.CL.NoiseLayer) and m.noise is not None:
                m.noise = m.noise.bfloat16()
        z, real = z.bfloat16(), real.bfloat16()
        torch.randn = lambda *a, **k: _randn(*a, **k).bfloat16()
    elif mode == "f64":
        torch.randn = lambda *a, **k: _randn(*a, **k).double()
    import contextlib
    ctx = (lambda: torch.autocast("cpu", dtype=torch.bfloat16)) if mode == "ac" else contextlib.nullcontext
    try:
        torch.manual_seed(77); random.seed(77)
        with ctx():
            d_loss = float(sg.optimize_discriminator(z, real, depth, ALPHA))
        d_grads = {k: p.grad.detach().double().clone() for k, p in sg.dis.named_parameters() if p.grad is not None}
        state = {k: v.detach().double().clone() for k, v in sg.dis.state_dict().items()}
        if d_after is not None:
            own = sg.dis.state_dict()
            sg.dis.load_state_dict({k: d_after[k].to(own[k].dtype) for k in own})
        torch.manual_seed(78); random.seed(78)
        with ctx():
            g_loss = float(sg.optimize_generator(z, real, depth, ALPHA))
        g_grads = {k: p.grad.detach().double().clone() for k, p in sg.gen.named_parameters() if p.grad is not None}
    finally:
        torch.randn = _randn
    return d_loss, g_loss, d_grads, g_grads, state


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "128"
    cfg = MR.CONFIGS[name]
    t0 = time.time()
    ref = run(cfg, "f64")
    print(f"f64: d_loss {ref[0]:.6f} g_loss {ref[1]:.6f}  {time.time() - t0:.0f} s", flush=True)
    out = {"config": np.array(name), "f64_d_loss": ref[0], "f64_g_loss": ref[1]}
    for mode in ("ac", "bf16"):
        t0 = time.time()
        try:
            got = run(cfg, mode, d_after=ref[4])
        except Exception as e:                                   # an op without a CPU bf16 kernel: say so in the fixture
            print(f"{mode}: the reference does not run in this mode on CPU: {type(e).__name__}: {e}", flush=True)
            out[f"{mode}_runs"] = 0
            continue
        out[f"{mode}_runs"] = 1
        out[f"{mode}_d_loss_rel"] = abs(got[0] - ref[0]) / abs(ref[0]); out[f"{mode}_g_loss_rel"] = abs(got[1] - ref[1]) / abs(ref[1])
        # the f64 run clips G's gradients (max-norm 10) before we read them; so does this run, each by its own norm: compare directions
        # and norms after undoing neither -- both sides are post-clip, like the fixtures of make_golden_real.py
        for net, a, b in (("d", got[2], ref[2]), ("g", got[3], ref[3])):
            names = sorted(b)
            rel, cos = [], []
            for k in names:
                x, y = a[k].reshape(-1), b[k].reshape(-1)
                rel.append(float(torch.linalg.vector_norm(x - y) / (torch.linalg.vector_norm(y) + 1e-30)))
                cos.append(max(0.0, 1.0 - float(torch.dot(x, y) / (torch.linalg.vector_norm(x) * torch.linalg.vector_norm(y) + 1e-30))))
            out[f"{net}_grad_names"] = np.array(names)
            out[f"{mode}_{net}_grad_rel"] = np.array(rel); out[f"{mode}_{net}_grad_1mcos"] = np.array(cos)
            keep = [i for i, k in enumerate(names) if not k.endswith("init_block.bias")]
            print(f"{mode} {net}: gradient rel-L2 median {np.median(np.array(rel)[keep]):.3f} max {np.max(np.array(rel)[keep]):.3f}; "
                  f"1-cos max {np.max(np.array(cos)[keep]):.3f}", flush=True)
        print(f"{mode}: d_loss rel {out[f'{mode}_d_loss_rel']:.2e} g_loss rel {out[f'{mode}_g_loss_rel']:.2e}  {time.time() - t0:.0f} s", flush=True)
    MG.npz(f"bf16_anchor_{name}.npz", **out)


if __name__ == "__main__":
    main()
