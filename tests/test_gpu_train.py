"""-m gpu: the reference's training DRIVER flow on the HIP path -- reference train.py:51-54 (defaults merged with the
YAML), :84-99 (``StyleGAN(...)`` from the config nodes) and :129-139 (``style_gan.train(...)``) -- for configs/sample.yaml,
first two depths (4x4 -> 8x8, BASELINE configs[0]) on a synthetic dataset; then resuming from the checkpoints it wrote
the way train.py:24-29,106-126 does."""
import logging
import os

import pytest
import torch

from gpu_util import DEV
from test_train_schedule import SAMPLE_YAML

pytestmark = pytest.mark.gpu


def test_sample_yaml_trains_first_two_depths(tmp_path):
    from stylegan.pytorch_amd.GAN import StyleGAN
    from stylegan.pytorch_amd.config import default_cfg
    from stylegan.pytorch_amd.data import SyntheticImages
    y = tmp_path / "sample.yaml"; y.write_text(SAMPLE_YAML)
    opt = default_cfg()
    opt.merge_from_file(str(y))
    out = str(tmp_path / "run")
    opt.merge_from_list(["output_dir", out, "dataset.img_dir", "synthetic"])        # the YAML points at /home/hzh/...
    opt.freeze()
    os.makedirs(opt.output_dir)
    log = logging.getLogger("train-test"); log.handlers = []; log.propagate = False; log.setLevel(logging.INFO)
    lines = []

    class H(logging.Handler):
        def emit(self, r): lines.append(r.getMessage())
    log.addHandler(H())
    dataset = SyntheticImages(256, opt.dataset.resolution, opt.dataset.channels, seed=3)
    torch.manual_seed(0)            # the networks' initial weights: not whatever state the tests before this one left the generator in
    style_gan = StyleGAN(structure=opt.structure, conditional=opt.conditional, n_classes=opt.n_classes,
                         resolution=opt.dataset.resolution, num_channels=opt.dataset.channels,
                         latent_size=opt.model.gen.latent_size, g_args=opt.model.gen, d_args=opt.model.dis,
                         g_opt_args=opt.model.g_optim, d_opt_args=opt.model.d_optim, loss=opt.loss, drift=opt.drift,
                         d_repeats=opt.d_repeats, use_ema=opt.use_ema, ema_decay=opt.ema_decay, device=torch.device(DEV))
    epochs = list(opt.sched.epochs[:2]) + [0] * (len(opt.sched.epochs) - 2)        # sample.yaml's [2, 4, ...]: depths 0 and 1 only
    torch.manual_seed(0)
    # num_workers 0 (train.py passes opt.num_works = 4): the loader's worker processes are outside the path under test, and forking four
    # of them per epoch from a pytest process that has run 500 GPU tests took 280 of the suite's 646 s (25 s when this file runs alone)
    assert opt.num_works == 4
    style_gan.train(dataset=dataset, num_workers=0, epochs=epochs, batch_sizes=opt.sched.batch_sizes,
                    fade_in_percentage=opt.sched.fade_in_percentage, logger=log, output=opt.output_dir,
                    num_samples=opt.num_samples, start_depth=0, feedback_factor=opt.feedback_factor,
                    checkpoint_factor=opt.checkpoint_factor)
    # 256 images / batch 128 = 2 batches per epoch; feedback on every batch (int(2/10 + 1) = 1): 2*2 + 4*2 lines
    fb = [l for l in lines if l.startswith("Elapsed")]
    assert len(fb) == 12, fb
    steps = [int(l.split("Step: ")[1].split()[0]) for l in fb]
    assert steps == list(range(1, 13))
    for l in fb:
        d, g = float(l.split("D_Loss: ")[1].split()[0]), float(l.split("G_Loss: ")[1])
        # finite and not exploding.  (the first D loss of freshly initialised networks at 4x4 is dominated by the R1 term, 5 |grad|^2:
        # hundreds to ~1.2e3 depending on the initial weights -- measured 1170.9 with the generator state another test order left)
        assert d == d and g == g and abs(d) < 1e4 and abs(g) < 1e4, l
    assert lines[-1].startswith("Training completed")
    # round 5: the loop decided per depth, from iterations 2..4, between eager launches and hipGraph replay (StyleGAN.train_graphs "auto");
    # depth 1 (index) ran its last four iterations in the mode it chose -- replay included where this box's host is the limit
    mode_lines = [l for l in lines if "launches (host enqueue" in l]
    assert len(mode_lines) == 2 and mode_lines[0].startswith("Depth 1:") and mode_lines[1].startswith("Depth 2:"), mode_lines
    assert style_gan.use_graphs is False                                            # the caller's setting is restored
    assert sorted(os.listdir(os.path.join(out, "samples")))[0].startswith("gen_0_1_1")
    models = os.listdir(os.path.join(out, "models"))
    # checkpoint epochs: depth 0 -> 1, 2 (first and last); depth 1 -> 1, 4: five files each
    assert len(models) == 5 * 4, sorted(models)
    # ---- resume as train.py does: fresh object, load G (subset loader of :24-29), D, shadow and both optimizers
    sg2 = StyleGAN(structure=opt.structure, resolution=opt.dataset.resolution, num_channels=opt.dataset.channels,
                   latent_size=opt.model.gen.latent_size, g_args=opt.model.gen, d_args=opt.model.dis,
                   g_opt_args=opt.model.g_optim, d_opt_args=opt.model.d_optim, loss=opt.loss, use_ema=opt.use_ema,
                   ema_decay=opt.ema_decay, device=torch.device(DEV))
    m = os.path.join(out, "models")
    pre = torch.load(os.path.join(m, "GAN_GEN_1_4.pth"))
    md = sg2.gen.state_dict(); md.update({k: v for k, v in pre.items() if k in md}); sg2.gen.load_state_dict(md)
    sg2.dis.load_state_dict(torch.load(os.path.join(m, "GAN_DIS_1_4.pth")))
    sg2.gen_shadow.load_state_dict(torch.load(os.path.join(m, "GAN_GEN_SHADOW_1_4.pth")))
    sg2.gen_optim.load_state_dict(torch.load(os.path.join(m, "GAN_GEN_OPTIM_1_4.pth")))
    sg2.dis_optim.load_state_dict(torch.load(os.path.join(m, "GAN_DIS_OPTIM_1_4.pth")))
    for (k, a), (_, b) in zip(sg2.gen.state_dict().items(), style_gan.gen.state_dict().items()):
        assert torch.equal(a, b), k
    z = torch.randn(8, 512, device=DEV); real = torch.randn(8, 3, 128, 128, device=DEV)
    d = float(sg2.optimize_discriminator(z, real, 1, 0.5)); g = float(sg2.optimize_generator(z, real, 1, 0.5))
    assert d == d and g == g
