"""Deterministic parameter fill + tiny-model settings shared by the golden-vector generator
(tests/golden/make_golden.py, runs only where /root/reference exists) and by the tests.

Weights are never stored in fixtures: every tensor is regenerated from a seed derived from its
``state_dict`` key, so a fixture holds inputs/outputs only.
"""
import zlib

import torch

# tiny networks used for the network / step fixtures (reference kwargs: models/GAN.py:106,303)
TINY = dict(resolution=128, fmap_base=64, fmap_max=8, mapping_layers=2, latent_size=512)
TINY_DEPTH = 6          # log2(128) - 1


def _gen(name: str) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return g


def fill_value(name: str, shape, dtype=torch.float32) -> torch.Tensor:
    """Deterministic non-degenerate value for a state_dict entry (exercises noise weights and
    biases, which the reference initialises to zero)."""
    r = torch.randn(tuple(shape), generator=_gen(name), dtype=torch.float64)
    mapping = name.startswith("g_mapping")
    if name.endswith("noise.weight"):
        v = 0.3 * r
    elif name.endswith("init_block.const"):
        v = 1.0 + 0.5 * r
    elif name.endswith("init_block.bias"):
        v = 1.0 + 0.1 * r
    elif name.endswith("avg_latent"):
        v = 0.1 * r
    elif name.endswith(".bias"):
        v = (10.0 if mapping else 0.1) * r
    elif name.endswith(".weight"):
        v = (100.0 if mapping else 1.0) * r
    else:
        raise KeyError(name)
    return v.to(dtype)


def fill_state(keys_shapes, dtype=torch.float32):
    """keys_shapes: iterable of (name, shape).  Buffers named ``*.kernel`` are skipped."""
    return {k: fill_value(k, s, dtype) for k, s in keys_shapes if not k.endswith(".kernel")}


def seeded(shape, seed: int, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator(); g.manual_seed(seed)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float64).to(dtype)


def tensor_stats(t: torch.Tensor):
    t = t.detach().double().reshape(-1)
    return [float(t.sum()), float(t.abs().sum()), float(torch.linalg.vector_norm(t))]
