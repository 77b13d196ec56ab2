"""-m gpu: the bf16 activation-storage mode -- the dtype BASELINE.json's metric is quoted in -- at the network level:
per-tensor parameter gradients of a full D+G iteration against the fp64 oracle (MID and the real ffhq128 widths), the
headline configuration (ffhq1024, depth index 8) at the BENCHMARKED batch 4 (four minibatch-stddev groups) forward + losses
against the fp64 oracle, and the exact timed mode (bf16, hipGraph replay, auxiliary + side stream) against the eager
single-stream step.

Gates: 2 x the error MEASURED on the MI355X for this code, recorded in tests/golden/bf16_gates.json by running this file
with SGX_RECORD_BF16_GATES=<path> (tools/gpu_s.sh; the kernels are bit-deterministic, so a gate is a statement about the
arithmetic, not about noise).  Reference functions: models/Losses.py:192-229 (logistic + R1), models/GAN.py:591-659,
models/CustomLayers.py:288-305 (minibatch stddev groups)."""
import json
import os
import random

import numpy as np
import pytest
import torch

import golden_util as gu
from gpu_util import DEV, MID, MID_DEPTH, load_into, mid_params, pin_noise, rel_err
from oracle import stylegan_oracle as O
import test_gpu_realconfigs as RC

pytestmark = pytest.mark.gpu
GATES_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_gates.json")
RECORD = os.environ.get("SGX_RECORD_BF16_GATES")


@pytest.fixture(scope="module", autouse=True)
def _oracle_threads():
    n = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    yield
    torch.set_num_threads(n)


def gate(section, measured):
    """measured: {name: value} (errors: smaller is better).  Recording run: merge into the JSON at RECORD.  Otherwise every value
    must stay within 2 x the committed one (+ a floor for quantities that are zero up to round-off)."""
    if RECORD:
        data = json.load(open(RECORD)) if os.path.exists(RECORD) else {}
        data[section] = {k: float(v) for k, v in measured.items()}
        json.dump(data, open(RECORD, "w"), indent=1, sort_keys=True)
        return
    want = json.load(open(GATES_PATH))[section]
    assert sorted(want) == sorted(measured), (section, sorted(set(want) ^ set(measured)))

    def floor(k):
        # quantities that are small differences of large ones move by more than 2x under ANY change of the summation / rounding
        # order (a fused kernel, another tile size): gradients of a tensor to 2e-2 of its norm, directions to 1e-3, scalars 1e-4
        return 2e-2 if k.endswith(":rel") else (1e-3 if k.endswith(":1-cos") else 1e-4)
    bad = {k: (measured[k], want[k]) for k in want if measured[k] > 2.0 * want[k] + floor(k)}
    assert not bad, f"{section}: beyond 2x the measured error: {bad}"


MID_CFG = dict(resolution=MID["resolution"], mapping_layers=MID["mapping_layers"], psi=0.7, depth=5, batch=4, total_depth=MID_DEPTH)


def mid_stylegan(act_dtype):
    from stylegan.pytorch_amd.GAN import StyleGAN
    opt = dict(learning_rate=0.003, beta_1=0, beta_2=0.99, eps=1e-8)
    sg = StyleGAN(structure="linear", resolution=128, num_channels=3, latent_size=512,
                  g_args=dict(latent_size=512, mapping_layers=MID["mapping_layers"], blur_filter=[1, 2, 1], truncation_psi=0.7,
                              truncation_cutoff=8, fmap_base=MID["fmap_base"], fmap_max=MID["fmap_max"]),
                  d_args=dict(use_wscale=True, blur_filter=[1, 2, 1], fmap_base=MID["fmap_base"], fmap_max=MID["fmap_max"]),
                  g_opt_args=opt, d_opt_args=opt, loss="logistic", d_repeats=1, use_ema=True, ema_decay=0.999,
                  device=torch.device(DEV), act_dtype=act_dtype)
    gp, dp = mid_params(torch.float64)
    load_into(sg.gen, gp); load_into(sg.dis, dp); load_into(sg.gen_shadow, gp)
    sg.gen.train(); sg.dis.train()
    pin_noise(sg.gen, RC.noises(MID_CFG))
    return sg, gp, dp


@pytest.mark.parametrize("name", ["mid", "128"])
def test_bf16_step_gradients_vs_fp64(name):
    """Every parameter gradient of one full bf16 D+G iteration against the fp64 oracle: rel-L2 and 1 - cosine per tensor."""
    if name == "mid":
        cfg = MID_CFG
        sg, gp, dp = mid_stylegan(torch.bfloat16)
    else:
        cfg = RC.CFG[name]
        sg, gp, dp = RC.make_stylegan(cfg, torch.bfloat16)
    z, real, d_loss, g_loss, d_grads, g_grads = RC.run_step(sg, cfg)
    od, og, odg, ogg, _ = RC.oracle_step(cfg, gp, dp, z, real)
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(v.double()) for v in g_grads.values()])).item()
    coef = min(1.0, 10.0 / (total + 1e-6))                                   # oracle G gradients are post-clip, ours pre-clip
    measured = {"d_loss": abs(d_loss - od) / abs(od), "g_loss": abs(g_loss - og) / abs(og)}
    rels = []
    for net, ours, ref, scale in (("d", d_grads, odg, 1.0), ("g", g_grads, ogg, coef)):
        assert sorted(ours) == sorted(k for k, v in ref.items() if v is not None)
        for k, v in ours.items():
            if k.endswith("init_block.bias"):
                continue                     # analytically zero gradient (the instance norm removes it): pure round-off
            a = v.double().cpu().reshape(-1) * scale; r = ref[k].double().reshape(-1)
            rel = (torch.linalg.vector_norm(a - r) / (torch.linalg.vector_norm(r) + 1e-30)).item()
            cos = (torch.dot(a, r) / (torch.linalg.vector_norm(a) * torch.linalg.vector_norm(r) + 1e-30)).item()
            measured[f"{net}:{k}:rel"] = rel
            measured[f"{net}:{k}:1-cos"] = max(0.0, 1.0 - cos)
            rels.append((rel, f"{net}:{k}"))
    rels.sort(reverse=True)
    measured["median_rel"] = float(np.median([r for r, _ in rels]))
    print(f"[bf16 grads {name}] d_loss rel {measured['d_loss']:.2e} g_loss rel {measured['g_loss']:.2e}; gradient rel-L2 median "
          f"{measured['median_rel']:.2e}; worst: " + ", ".join(f"{k} {r:.1e}" for r, k in rels[:6]))
    gate(f"grads_{name}", measured)
    # absolute sanity on top of the relative gates (measured: median rel-L2 0.10 on both models, worst tensor 0.45 / 1 - cos 0.11
    # -- the generator's first layers, at the far end of two networks of bf16-stored activations and of every LeakyReLU kink
    # that a bf16 rounding flips; the discriminator's head is at 4e-3): no tensor points elsewhere
    assert measured["median_rel"] < 0.2 and all(measured[k] < 0.25 for k in measured if k.endswith(":1-cos"))


def test_bf16_headline_config_at_the_benchmarked_batch():
    """ffhq1024, depth index 8, batch 4 (= the bench.py workload: four minibatch-stddev groups of one... G = 4 groups): G image,
    D scores and both losses of a bf16 iteration against the fp64 oracle run here on the same weights, noise and seeds."""
    cfg = dict(RC.CFG["1024"], batch=4)
    sg, gp, dp = RC.make_stylegan(cfg, torch.bfloat16)
    z, real, img, score, score_fake = RC.forward_pair(sg, cfg)
    with torch.no_grad():
        ref, _ = O.generator(gp, z.double(), cfg["depth"], RC.ALPHA, RC.noises(cfg), mapping_layers=cfg["mapping_layers"],
                             num_layers=2 * cfg["total_depth"], truncation_psi=cfg["psi"])
        ref_s = O.discriminator(dp, real.double(), cfg["depth"], RC.ALPHA, cfg["total_depth"])
    measured = {"image": rel_err(img, ref), "d_score_real": rel_err(score, ref_s)}
    del ref
    z, real, d_loss, g_loss, _, _ = RC.run_step(sg, cfg)
    od, og, _, _, _ = RC.oracle_step(cfg, gp, dp, z, real)
    measured["d_loss"] = abs(d_loss - od) / abs(od); measured["g_loss"] = abs(g_loss - og) / abs(og)
    print("[bf16 ffhq1024 B=4] " + ", ".join(f"{k} rel {v:.2e}" for k, v in measured.items()))
    gate("real1024_b4", measured)
    for p in list(sg.gen.parameters()) + list(sg.dis.parameters()):
        assert torch.isfinite(p).all()


def test_bf16_timed_mode_matches_the_eager_single_stream_step(monkeypatch):
    """The mode bench.py times -- bf16, hipGraph replay, auxiliary + side stream -- against the eager single-stream step with the
    same arithmetic (``alpha_on_device``: the fade-in coefficient read from device memory, as a replayed graph must).  Same
    kernels on the same data: sharp (the fp32 tolerances of test_gpu_graphs, x10 for the bf16 re-rounding of ulp-level
    differences in the gradient sums)."""
    import test_gpu_graphs as TG
    iters = 4
    monkeypatch.setenv("SGX_AUX_STREAM", "0"); monkeypatch.setenv("SGX_PARAM_STREAM", "0")
    real_make = TG.make

    def make_dev_alpha(use_graphs, act_dtype, **kw):
        sg = real_make(use_graphs, act_dtype, **kw)
        sg.alpha_on_device = True
        return sg
    monkeypatch.setattr(TG, "make", make_dev_alpha)
    le, se, sge = TG.run(False, torch.bfloat16, iters)
    assert "_aux_compute_stream" not in sge.__dict__ and "_param_side_stream" not in sge.__dict__
    monkeypatch.setenv("SGX_AUX_STREAM", "1"); monkeypatch.setenv("SGX_PARAM_STREAM", "1")
    lg, sgr, sg = TG.run(True, torch.bfloat16, iters)
    assert all(g.graph is not None and g.calls == iters for g in sg._step_graphs.values()) and len(sg._step_graphs) == 2
    assert "_aux_compute_stream" in sg.__dict__ and "_param_side_stream" in sg.__dict__
    TG.losses_agree(le, lg, 10.0)
    for part in ("gen", "dis", "shadow"):
        for k, v in se[part].items():
            assert k in TG.SKIP or TG.close(sgr[part][k], v, 5e-2), (part, k)
