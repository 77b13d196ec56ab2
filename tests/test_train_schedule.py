"""CPU: the product's ``StyleGAN.train`` (reference models/GAN.py:682-826) replays the reference's progressive schedule
BIT-exactly -- (depth, alpha, int-ness of alpha) per iteration, feedback ticks and checkpoint epochs -- against
tests/golden/schedule.npz, which tests/golden/make_golden.py recorded from the reference's own loop with the compute
methods stubbed.  The compute is stubbed here in exactly the same way.  Also: the yacs-free config tree merges the
reference's YAMLs and feeds ``StyleGAN(...)`` / ``train(...)`` the way reference train.py:51-54,84-139 does."""
import logging
import os
import tempfile

import numpy as np
import pytest
import torch

from stylegan.pytorch_amd import GAN as G
from stylegan.pytorch_amd.config import CfgNode, default_cfg


def run_product_loop(num_images, epochs, batch_sizes, fade, start_depth, feedback_factor, checkpoint_factor, monkeypatch):
    rec, marks = [], []
    sg = G.StyleGAN.__new__(G.StyleGAN)                       # no device, no networks: only the loop is under test
    sg.depth = len(epochs); sg.structure = "linear"; sg.use_ema = False; sg.conditional = False
    sg.latent_size = 8; sg.device = torch.device("cpu"); sg.n_classes = 0
    lin_g, lin_d = torch.nn.Linear(1, 1), torch.nn.Linear(1, 1)
    sg.gen_optim = torch.optim.SGD(lin_g.parameters(), lr=0.1)
    sg.dis_optim = torch.optim.SGD(lin_d.parameters(), lr=0.1)
    sg.dis = lin_d

    class GenStub:
        def __call__(self, *a, **k): return torch.zeros(1, 3, 4, 4)
        def train(self): return None
        def state_dict(self): return lin_g.state_dict()
    sg.gen = GenStub()
    monkeypatch.setattr(G, "get_data_loader", lambda dataset, batch_size, num_workers: [torch.zeros(batch_size, 1)] * (num_images // batch_size))
    sg.optimize_discriminator = lambda noise, images, depth, alpha, labels=None: 0.0

    def og(noise, images, depth, alpha, labels=None):
        rec.append((depth, float(alpha), int(isinstance(alpha, int))))
        return 0.0
    sg.optimize_generator = og
    sg.create_grid = lambda **kw: None
    log = logging.getLogger("schedule-test"); log.handlers = []; log.propagate = False

    class H(logging.Handler):
        def emit(self, r):
            m = r.getMessage()
            if m.startswith("Elapsed"):
                marks.append(len(rec))
            if m.startswith("Saving the model to") and "GAN_GEN_" in m and "SHADOW" not in m:
                marks.append(-len(rec))
    log.addHandler(H()); log.setLevel(logging.INFO)
    with tempfile.TemporaryDirectory() as td:
        sg.train(None, 0, epochs, batch_sizes, fade, log, td, num_samples=1, start_depth=start_depth,
                 feedback_factor=feedback_factor, checkpoint_factor=checkpoint_factor)
        saved = sorted(os.listdir(os.path.join(td, "models")))
    return rec, marks, saved


def test_train_replays_the_reference_schedule_bit_exactly(golden_dir, monkeypatch):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    for ci in range(2):
        num_images, start_depth, ff, cf = (int(v) for v in g[f"c{ci}_cfg"])
        epochs, bs, fade = list(map(int, g[f"c{ci}_epochs"])), list(map(int, g[f"c{ci}_bs"])), list(map(int, g[f"c{ci}_fade"]))
        rec, marks, saved = run_product_loop(num_images, epochs, bs, fade, start_depth, ff, cf, monkeypatch)
        want = g[f"c{ci}_rec"]
        got = np.array(rec, dtype=np.float64)
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.int64), want.view(np.int64)), "schedule differs bitwise"      # alpha: same float bits
        assert marks == [int(v) for v in g[f"c{ci}_marks"]]
        # checkpoint files: G, D, both optimizers per checkpoint epoch, named as the reference names them
        n_ckpt = sum(1 for m in marks if m < 0)
        assert len(saved) == 4 * n_ckpt and all(s.startswith(("GAN_GEN_", "GAN_DIS_")) for s in saved)


def test_schedule_helpers_match_the_oracle_generator():
    """The four predicates/formulas against the oracle's restatement over odd sizes (feedback divisor truncation, alpha
    staying an int 1 after the fade point)."""
    from oracle import stylegan_oracle as O
    for num_images, bs, ep, fd, ff, cf in [(97, 16, 3, 50, 10, 2), (1000, 64, 2, 30, 4, 3), (13, 4, 5, 75, 100, 1), (64, 64, 1, 50, 1, 1)]:
        tb = num_images // bs
        for (d, epoch, i, ticker, step, alpha, feedback, ckpt) in O.schedule(num_images, [ep], [bs], [fd], 1, 0, ff, cf):
            fp = G.StyleGAN.fade_point_of(fd, ep, tb)
            a = G.StyleGAN.alpha_at(ticker, fp)
            assert a == alpha and type(a) is type(alpha)
            assert G.StyleGAN.is_feedback_batch(i, tb, ff) == feedback
            assert ((i == tb) and G.StyleGAN.is_checkpoint_epoch(epoch, ep, cf)) == ckpt


SAMPLE_YAML = """# reference configs/sample.yaml (values restated)
output_dir: '/data/hzh/checkpoints/StyleGAN.pytorch/ckp_celeba_6'
structure: 'linear'
device_id: ('3')
dataset:
  img_dir: '/home/hzh/data/img_align_celeba'
  folder: False
  resolution: 128
sched:
  epochs: [2,4,4,4,4,8]
"""
FFHQ1024_YAML = """structure: 'linear'
checkpoint_factor: 4
feedback_factor: 4
dataset:
  resolution: 1024
model:
  gen:
    mapping_layers: 8
    truncation_psi: -1.
sched:
  epochs: [8,16,32,32,64,64]
"""


def test_config_tree_merges_reference_yamls(tmp_path):
    p = tmp_path / "sample.yaml"; p.write_text(SAMPLE_YAML)
    opt = default_cfg()
    opt.merge_from_file(str(p))
    opt.merge_from_list(["output_dir", str(tmp_path / "out"), "dataset.img_dir", "synthetic:64"])     # the non-YAML override
    opt.freeze()
    assert opt.structure == "linear" and opt.device_id == "3" and opt.dataset.resolution == 128 and opt.dataset.folder is False
    assert opt.sched.epochs == [2, 4, 4, 4, 4, 8] and opt.sched.batch_sizes[:3] == [128, 128, 128]
    assert opt.loss == "logistic" and opt.use_ema is True and opt.model.gen.mapping_layers == 4
    assert dict(**opt.model.g_optim) == dict(learning_rate=0.003, beta_1=0, beta_2=0.99, eps=1e-8)     # ** expansion as train.py:93-94
    assert isinstance(opt.model.gen, dict) and opt.output_dir.endswith("out")
    with pytest.raises(AttributeError):
        opt.loss = "hinge"                                   # frozen
    q = tmp_path / "ffhq.yaml"; q.write_text(FFHQ1024_YAML)
    o2 = default_cfg(); o2.merge_from_file(str(q))
    assert o2.model.gen.truncation_psi == -1.0 and o2.model.gen.mapping_layers == 8 and o2.dataset.resolution == 1024
    bad = tmp_path / "bad.yaml"; bad.write_text("no_such_key: 1\n")
    with pytest.raises(KeyError):
        default_cfg().merge_from_file(str(bad))
    bad.write_text("d_repeats: 'two'\n")
    with pytest.raises(ValueError):
        default_cfg().merge_from_file(str(bad))
    assert isinstance(CfgNode({"a": {"b": 1}}).a, CfgNode)
