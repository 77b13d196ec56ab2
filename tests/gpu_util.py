"""Helpers for the -m gpu parity tests: HIP path (stylegan.pytorch_amd) vs the CPU oracle on the same inputs."""
import torch

import golden_util as gu
from oracle import stylegan_oracle as O

DEV = "cuda:0"

# mid-size networks: every channel count is a multiple of 16 (MFMA granularity), 128x128 so both the fused and the
# non-fused up/down paths of the reference are exercised
MID = dict(resolution=128, fmap_base=1024, fmap_max=32, mapping_layers=2)
MID_DEPTH = 6


def rel_err(a, b):
    """rel-L2 of a (any device/dtype) against reference b, in fp64."""
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    den = torch.linalg.vector_norm(b).item()
    return torch.linalg.vector_norm(a - b).item() / (den + 1e-30)


def assert_close(a, b, tol, what="", floor=0.0):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, tuple(a.shape), tuple(b.shape))
    err = torch.linalg.vector_norm(a - b).item()
    den = torch.linalg.vector_norm(b).item()
    assert err <= tol * den + floor, f"{what}: rel-L2 {err / (den + 1e-30):.3e} > {tol:.1e} (|ref|={den:.3e})"


def mid_params(dtype=torch.float64):
    gp = O.make_generator_params(MID["resolution"], MID["mapping_layers"], 512, MID["fmap_base"], MID["fmap_max"], dtype=dtype)
    dp = O.make_discriminator_params(MID["resolution"], MID["fmap_base"], MID["fmap_max"], dtype=dtype)
    for p in (gp, dp):
        for k in list(p):
            rg = p[k].requires_grad
            p[k] = gu.fill_value(k, p[k].shape, dtype).requires_grad_(rg)
    return gp, dp


def mid_noises(batch, dtype=torch.float64, seed0=100):
    return [gu.seeded((batch, 1, 4 * 2 ** (i // 2), 4 * 2 ** (i // 2)), seed0 + i, dtype) for i in range(2 * MID_DEPTH)]


def load_into(module, params):
    """Copy oracle params (reference state_dict keys) into one of our modules."""
    sd = {k: v.detach().float() for k, v in params.items()}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith(".kernel") for k in missing), missing
    return module


def pin_noise(gen, noises):
    from stylegan.pytorch_amd.CustomLayers import NoiseLayer
    mods = [m for m in gen.modules() if isinstance(m, NoiseLayer)]
    assert len(mods) == len(noises)
    for m, n in zip(mods, noises):
        m.noise = n.float().to(DEV)


def build_mid(act_dtype=torch.float32):
    from stylegan.pytorch_amd.GAN import Discriminator, Generator
    gen = Generator(resolution=MID["resolution"], latent_size=512, mapping_layers=MID["mapping_layers"], blur_filter=[1, 2, 1],
                    truncation_psi=0.7, truncation_cutoff=8, fmap_base=MID["fmap_base"], fmap_max=MID["fmap_max"],
                    structure="linear", act_dtype=act_dtype).to(DEV)
    dis = Discriminator(resolution=MID["resolution"], num_channels=3, use_wscale=True, blur_filter=[1, 2, 1],
                        fmap_base=MID["fmap_base"], fmap_max=MID["fmap_max"], structure="linear", act_dtype=act_dtype).to(DEV)
    return gen, dis
