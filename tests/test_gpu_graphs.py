"""-m gpu: the hipGraph-replayed training iteration (StyleGAN(use_graphs=True)) against the eager one, over iterations
that change the fade-in alpha, the latents, the images and the style-mixing draw (all of which reach the graph through
static device tensors, not through kernel arguments), with the per-layer noise pinned.

Both run the same kernels on the same data.  The comparison is to a few fp32 ulps rather than bit for bit because
torch's autograd engine sums the three gradient contributions of a discriminator parameter (fake pass, real pass,
R1 double-backward) in an order that depends on thread-local sequence numbers, i.e. on the history of the process: two
EAGER runs in one process already differ by one ulp in such a gradient (measured), and Adam with beta1 = 0 then carries
that forward (a wrong alpha, a stale Adam scalar or a stale weight pack would show up at the 1e-2 level)."""
import random

import pytest
import torch

import golden_util as gu
from gpu_util import DEV, MID, load_into, mid_params, pin_noise
from test_gpu_networks import mid_noises

pytestmark = pytest.mark.gpu


def make(use_graphs, act_dtype, psi=0.7, dp=None):
    from stylegan.pytorch_amd.GAN import StyleGAN
    kw = dict(learning_rate=0.003, beta_1=0, beta_2=0.99, eps=1e-8)
    sg = StyleGAN(structure="linear", resolution=128, num_channels=3, latent_size=512,
                  g_args=dict(latent_size=512, mapping_layers=MID["mapping_layers"], blur_filter=[1, 2, 1], truncation_psi=psi,
                              truncation_cutoff=8, fmap_base=MID["fmap_base"], fmap_max=MID["fmap_max"]),
                  d_args=dict(use_wscale=True, blur_filter=[1, 2, 1], fmap_base=MID["fmap_base"], fmap_max=MID["fmap_max"]),
                  g_opt_args=kw, d_opt_args=kw, loss="logistic", d_repeats=1, use_ema=True, ema_decay=0.999,
                  device=torch.device(DEV), act_dtype=act_dtype, use_graphs=use_graphs, data_parallel=dp)
    gp, dpar = mid_params(torch.float64)
    if psi <= 0:
        gp = {k: v for k, v in gp.items() if not k.startswith("truncation.")}
    load_into(sg.gen, gp); load_into(sg.dis, dpar); load_into(sg.gen_shadow, gp)
    sg.gen.train(); sg.dis.train()
    pin_noise(sg.gen, mid_noises(4))
    return sg


def run(use_graphs, act_dtype, iters, depth=5, dev_alpha=False, **kw):
    sg = make(use_graphs, act_dtype, **kw)
    if dev_alpha:
        sg.alpha_on_device = True
    torch.manual_seed(5); random.seed(5)
    losses = []
    for i in range(iters):
        alpha = min(1.0, 0.25 + 0.15 * i)
        z = gu.seeded((4, 512), 100 + i).to(DEV); real = gu.seeded((4, 3, 128, 128), 200 + i).to(DEV)
        d = sg.optimize_discriminator(z, real, depth, alpha)
        g = sg.optimize_generator(z, real, depth, alpha)
        losses.append((float(d), float(g)))
    torch.cuda.synchronize()
    state = {"gen": {k: v.detach().clone() for k, v in sg.gen.state_dict().items()},
             "dis": {k: v.detach().clone() for k, v in sg.dis.state_dict().items()},
             "shadow": {k: v.detach().clone() for k, v in sg.gen_shadow.state_dict().items()},
             "dgrad": {k: p.grad.detach().clone() for k, p in sg.dis.named_parameters() if p.grad is not None},
             "dstep": [float(st["step"]) for st in sg.dis_optim.state.values() if "step" in st]}
    return losses, state, sg


# g_synthesis.init_block.bias has an analytically ZERO gradient (it feeds an instance norm), so its computed gradient is
# pure round-off and Adam with beta1 = 0 turns that into +-lr steps: not comparable between two runs of anything.
SKIP = ("g_synthesis.init_block.bias",)


# loss agreement per iteration: ulp-level at first, then the amplification described in the module docstring.  Measured
# relative differences between two runs of the same kernels on the fp32 mid-size model, over several library versions
# (the growth depends on which elements the round-off lands on): 1e-7, 2e-7..5e-7, 5e-6..1.5e-5, 7e-5..1.1e-3, ...
# The first three entries are the sharp ones: a wrong alpha, a stale Adam scalar, a stale weight pack or a cross-stream
# race shows at the 1e-2 level in iteration 0..2.
LOSS_TOL = [2e-6, 5e-6, 1e-4, 1e-2, 5e-2, 1e-1, 2e-1]
LR = 0.003


def losses_agree(a, b, scale=1.0):
    for i, ((d0, g0), (d1, g1)) in enumerate(zip(a, b)):
        tol = LOSS_TOL[min(i, len(LOSS_TOL) - 1)] * scale
        assert abs(d0 - d1) <= tol * abs(d0) and abs(g0 - g1) <= tol * abs(g0), (i, a, b)


def close(a, b, tol):
    """rel-L2.  Loose on purpose: with beta1 = 0 an element whose gradient is ~0 moves by +-lr on round-off alone, so a few
    elements of a bias can differ by 2*lr after some iterations; the sharp check is the per-iteration loss schedule."""
    a = a.double(); b = b.double()
    if float((a - b).norm()) <= tol * float(b.norm()) + 1e-12:
        return True
    # a zero-initialised bias after a few iterations is a handful of +-lr steps: one element whose first gradient was
    # round-off makes the relative norm meaningless; bound the absolute drift by the steps Adam can have taken instead
    return float((a - b).abs().max()) <= 2.0 * LR * 8


# 4 iterations = 2 eager warm-up calls + the capture + one pure replay: tight.  6 iterations: the sign-like Adam update
# (beta1 = 0) amplifies the ulp-level summation-order differences by ~10x per iteration, so the bound is loose there.
@pytest.mark.parametrize("act_dtype,iters,lscale,ptol", [(torch.float32, 4, 1.0, 3e-2), (torch.float32, 6, 1.0, 5e-2),
                                                         # bf16, eager with alpha_on_device (the arithmetic of the replay: the
                                                         # fade-in coefficient read from device memory, nothing folded into
                                                         # from_rgb): sharp.  x10: bf16 re-rounds ulp-level gradient-sum differences
                                                         (torch.bfloat16, 4, 10.0, 5e-2),
                                                         # bf16, eager with the HOST alpha: it folds the residual branch's (1-alpha)
                                                         # into from_rgb's weights where the replay scales the stored bf16
                                                         # activation -- two roundings of the same quantity, 5e-3 on the first G
                                                         # loss (measured): the documented difference between the two eager forms
                                                         (torch.bfloat16, 4, 5e3, 5e-2)])
def test_graph_replay_matches_eager(act_dtype, iters, lscale, ptol):
    le, se, _ = run(False, act_dtype, iters, dev_alpha=(act_dtype == torch.bfloat16 and lscale <= 10.0))   # 2 eager warm-up calls, the capture, pure replays
    lg, sgr, sg = run(True, act_dtype, iters)
    assert all(g.graph is not None and g.calls == iters for g in sg._step_graphs.values()) and len(sg._step_graphs) == 2
    losses_agree(le, lg, lscale)
    for part in ("gen", "dis", "shadow"):
        for k, v in se[part].items():
            assert k in SKIP or close(sgr[part][k], v, ptol), (part, k)
    assert max(se["dstep"]) == iters and sorted(set(sgr["dstep"])) in ([float(iters)], [0.0, float(iters)])   # graph_advance kept Adam's t
    assert set(se["dgrad"]) == set(sgr["dgrad"])                     # .grad of the replayed step is visible


def test_multi_stream_step_matches_single_stream(monkeypatch):
    """The step's independent branches run on extra streams (fake branch of the D step on an auxiliary stream, weight
    gradients on a side stream).  Same kernels, same data: against the single-stream run after 4 iterations."""
    monkeypatch.setenv("SGX_AUX_STREAM", "0"); monkeypatch.setenv("SGX_PARAM_STREAM", "0")
    l1, s1, _ = run(False, torch.float32, 4)
    monkeypatch.setenv("SGX_AUX_STREAM", "1"); monkeypatch.setenv("SGX_PARAM_STREAM", "1")
    l2, s2, sg = run(False, torch.float32, 4)
    assert "_aux_compute_stream" in sg.__dict__ and "_param_side_stream" in sg.__dict__
    losses_agree(l1, l2)
    for part in ("gen", "dis", "shadow"):
        for k, v in s1[part].items():
            assert k in SKIP or close(s2[part][k], v, 3e-2), (part, k)
    # auxiliary stream WITHOUT the side stream: both backward branches of the D step accumulate into the same .grad, so
    # the fake branch's weight-gradient launches must be ordered on the main stream (a race shows at the 1e-1 level)
    monkeypatch.setenv("SGX_AUX_STREAM", "1"); monkeypatch.setenv("SGX_PARAM_STREAM", "0")
    l3, s3, _ = run(False, torch.float32, 4)
    losses_agree(l1, l3)


def test_graph_and_eager_calls_interleave():
    """An eager call between replays (bench.py's surveyed step, or a user generating samples) sees current weights and
    leaves the graphs consistent."""
    le, se, _ = run(False, torch.float32, 7)
    sg = make(True, torch.float32)
    torch.manual_seed(5); random.seed(5)
    losses = []
    for i in range(7):
        sg.use_graphs = i != 4                                       # iteration 4 runs eagerly, after the capture
        alpha = min(1.0, 0.25 + 0.15 * i)
        z = gu.seeded((4, 512), 100 + i).to(DEV); real = gu.seeded((4, 3, 128, 128), 200 + i).to(DEV)
        d = sg.optimize_discriminator(z, real, 5, alpha); g = sg.optimize_generator(z, real, 5, alpha)
        losses.append((float(d), float(g)))
    losses_agree(le, losses)
    for k, v in se["gen"].items():
        assert k in SKIP or close(sg.gen.state_dict()[k], v, 1e-1), k


@pytest.mark.parametrize("eager_update", [True, False], ids=["eager-update", "update-graph"])
def test_data_parallel_graphs_split_around_the_all_reduce(monkeypatch, eager_update):
    """With a process group the half-iteration is [graph: losses + backward] -> eager all-reduce -> update (eager launches, the default
    since round 5, or a second graph: SGX_DP_EAGER_UPDATE=0).  One rank (RCCL group of size 1) on this box: same results as the
    single-process eager run."""
    import torch.distributed as dist
    from stylegan.pytorch_amd import GAN as G
    from stylegan.pytorch_amd.dist import DataParallelGroup
    monkeypatch.setattr(G, "DP_EAGER_UPDATE", eager_update)
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        # (eager runs with the fade-in coefficient in device memory, launch for launch the replay's arithmetic: what this test is about
        # is the split around the all-reduce; host vs device alpha is test_graph_replay_matches_eager's subject -- with the host form
        # the third generator loss sat at 1.2e-4 against this schedule's 1e-4 once the last epilogue moved into to_rgb, round 4)
        le, se, _ = run(False, torch.float32, 5, psi=-1.0, dev_alpha=True)
        # force_collectives: the bucketed RCCL all-reduce (side stream, flat buckets, multi-tensor copy-back) really runs
        lg, sgr, sg = run(True, torch.float32, 5, psi=-1.0, dp=DataParallelGroup(force_collectives=True, bucket_mb=1.0))
        graphs = list(sg._step_graphs.values())
        assert len(graphs) == 2 and all(g.graph is not None and g.split and (g.graph_update is None) == eager_update for g in graphs)
        # eager with a process group: all-reduce + update run on their own stream, overlapped with the next half-iteration
        la, sa, sga = run(False, torch.float32, 5, psi=-1.0, dev_alpha=True, dp=DataParallelGroup(force_collectives=True, bucket_mb=1.0))
        assert "_update_stream" in sga.__dict__
        for other_l, other_s in ((lg, sgr), (la, sa)):
            losses_agree(le, other_l)
            for part in ("gen", "dis", "shadow"):
                for k, v in se[part].items():
                    assert k in SKIP or close(other_s[part][k], v, 3e-2), (part, k)
    finally:
        dist.destroy_process_group()


def test_deferred_loss_behaves_like_a_float():
    from stylegan.pytorch_amd.GAN import DeferredLoss
    x = DeferredLoss(torch.tensor(2.5, device=DEV), scale=0.5)
    assert float(x) == 1.25 and "%.2f" % x == "1.25" and f"{x:.1f}" == "1.2" and x + 1 == 2.25 and abs(x - 1.25) == 0 and x < 2


def test_failed_capture_falls_back_to_a_correct_eager_step(monkeypatch):
    """A capture that aborts half way (here: the optimizer raises while its launches are being RECORDED) must leave
    nothing behind: not the never-executed weight packs, not Adam step counts advanced on the host, not the capture's
    gradient buffers.  Every iteration of the run that tried to capture equals the plain eager run."""
    from stylegan.pytorch_amd import GAN as G
    iters = 5
    le, se, _ = run(False, torch.float32, iters)
    real_step = G.FusedAdam.step
    armed = {"n": 0}

    def flaky_step(self, closure=None, grad_scale=None):
        if torch.cuda.is_current_stream_capturing():
            armed["n"] += 1
            real_step(self, closure, grad_scale)                     # advances the host-side step counts, records launches
            raise RuntimeError("injected failure inside the capture")
        return real_step(self, closure, grad_scale)
    monkeypatch.setattr(G.FusedAdam, "step", flaky_step)
    lg, sgr, sg = run(True, torch.float32, iters)
    assert armed["n"] >= 1 and sg.use_graphs is False                # the capture was attempted, failed, and was abandoned
    assert all(g.graph is None for g in sg._step_graphs.values())
    losses_agree(le, lg)
    for part in ("gen", "dis", "shadow"):
        for k, v in se[part].items():
            assert k in SKIP or close(sgr[part][k], v, 5e-2), (part, k)
    assert max(sgr["dstep"]) == iters == max(se["dstep"])            # no double-counted Adam step
