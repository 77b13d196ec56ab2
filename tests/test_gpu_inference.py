"""-m gpu: the inference path -- the Generator used the way the reference's generate scripts use it
(generate_samples.py:99-110, generate_mixing_figure.py:17-43, generate_truncation_figure.py:22-34): ``gen(z, depth=,
alpha=)``, ``gen.g_mapping``, ``gen.g_synthesis(dlatents, depth=, alpha=)``, ``gen.truncation.avg_latent`` -- on the HIP
kernels, against outputs recorded from the reference (tests/golden/inference_mid.npz, make_golden_inference.py)."""
import os
import random

import numpy as np
import pytest
import torch

import golden_util as gu
from gpu_util import DEV, assert_close, build_mid, load_into, mid_params, pin_noise

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a)).double()


def pool(img):
    return torch.nn.functional.avg_pool2d(img.detach().double().cpu(), 2)


def noises(batch):
    return [gu.seeded((batch, 1, 4 * 2 ** (i // 2), 4 * 2 ** (i // 2)), 100 + i, torch.float64) for i in range(12)]


def test_generate_scripts_flow_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "inference_mid.npz"))
    gp, _ = mid_params(torch.float64)
    gen, _ = build_mid()
    load_into(gen, gp)
    out_depth, latent_size = 5, 512
    tol = 2e-3                                            # fixture images are stored in fp16 (2^-11 relative)
    with torch.no_grad():
        # ---- generate_samples.py: train-mode forward under no_grad (the script never calls eval())
        pin_noise(gen, noises(1))
        torch.manual_seed(11); random.seed(11)
        point = torch.randn(1, latent_size)
        point = (point / point.norm()) * (latent_size ** 0.5)
        assert torch.equal(point, torch.from_numpy(g["sample_point"]))
        img = gen(point.to(DEV), depth=out_depth, alpha=1)
        assert img.shape == (1, 3, 128, 128)
        assert_close(pool(img), T(g["sample_img"]), tol, "generate_samples image")
        assert_close(gen.truncation.avg_latent, T(g["sample_avg_after"]), 1e-5, "W average after the sample")
        # ---- generate_mixing_figure.py
        src_seeds, dst_seeds, style_ranges = [639, 701], [888, 829], [range(0, 4), range(4, 8)]
        assert gen.g_mapping.latent_size == latent_size
        src = torch.from_numpy(np.stack([np.random.RandomState(s).randn(latent_size) for s in src_seeds]).astype(np.float32))
        dst = torch.from_numpy(np.stack([np.random.RandomState(s).randn(latent_size) for s in dst_seeds]).astype(np.float32))
        pin_noise(gen, noises(2))
        src_dl, dst_dl = gen.g_mapping(src.to(DEV)), gen.g_mapping(dst.to(DEV))
        assert src_dl.shape == (2, 12, 512)
        assert_close(src_dl[:, 0], T(g["mix_src_dlat0"]), 1e-4, "mapping output")
        assert_close(pool(gen.g_synthesis(src_dl, depth=out_depth, alpha=1)), T(g["mix_src_img"]), tol, "mixing: source images")
        assert_close(pool(gen.g_synthesis(dst_dl, depth=out_depth, alpha=1)), T(g["mix_dst_img"]), tol, "mixing: destination images")
        src_np, dst_np = src_dl.cpu().numpy(), dst_dl.cpu().numpy()
        for row in range(2):
            row_dl = np.stack([dst_np[row]] * 2)
            row_dl[:, style_ranges[row]] = src_np[:, style_ranges[row]]
            imgs = gen.g_synthesis(torch.from_numpy(row_dl).to(DEV), depth=out_depth, alpha=1)
            assert_close(pool(imgs), T(g[f"mix_row{row}_img"]), tol, f"mixing: row {row}")
        # ---- generate_truncation_figure.py
        seeds, psis = [91, 388], [1, 0.5, -0.5]
        lat = torch.from_numpy(np.stack([np.random.RandomState(s).randn(latent_size) for s in seeds]).astype(np.float32))
        dl = gen.g_mapping(lat.to(DEV)).detach().cpu().numpy()
        avg = gen.truncation.avg_latent.cpu().numpy()
        pin_noise(gen, noises(3))
        for row, d in enumerate(list(dl)):
            row_dl = (d[np.newaxis] - avg) * np.reshape(psis, [-1, 1, 1]) + avg
            imgs = gen.g_synthesis(torch.from_numpy(row_dl.astype(np.float32)).to(DEV), depth=out_depth, alpha=1)
            assert_close(pool(imgs), T(g[f"trunc_row{row}_img"]), tol, f"truncation: row {row}")


def test_checkpoint_interchange_with_reference_keys(tmp_path):
    """``torch.save(gen.state_dict())`` / ``load_state_dict`` round trip with the reference's key set (train.py:24-29,106-126,
    generate_samples.py:84): keys and shapes are the reference's, strict loading works both ways."""
    gp, _ = mid_params(torch.float32)
    gen, _ = build_mid()
    load_into(gen, gp)
    sd = gen.state_dict()
    want = set(gp) | {k for k in sd if k.endswith(".kernel")}
    assert set(sd) == want, sorted(set(sd) ^ want)[:8]
    f = tmp_path / "gen.pth"
    torch.save(sd, f)
    gen2, _ = build_mid()
    gen2.load_state_dict(torch.load(f))                     # strict
    z = gu.seeded((2, 512), 5).to(DEV)
    pin_noise(gen, noises(2)); pin_noise(gen2, noises(2))
    gen.eval(); gen2.eval()
    with torch.no_grad():
        assert torch.equal(gen(z, 5, 1.0), gen2(z, 5, 1.0))
