"""stylegan.pytorch_amd -- MI355X (gfx950) native StyleGAN generator/discriminator training hot path.

Same Python surface as the reference (huangzh13/StyleGAN.pytorch ``models/``): ``CustomLayers``, ``Blocks``,
``GAN`` (GMapping, GSynthesis, Generator, Discriminator, StyleGAN) and ``Losses``; the arithmetic runs in the
hand-written HIP kernels of ``libsgx_hip.so`` (C ABI: include/sgx.h).  There is no CPU or PyTorch-eager fallback.
"""
from . import native  # noqa: F401

__all__ = ["native", "functional", "CustomLayers", "Blocks", "GAN", "Losses", "optim"]
