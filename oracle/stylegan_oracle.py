"""CPU oracle for the StyleGAN G+D training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``stylegan/pytorch_amd`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
there only as the checker / the timed CPU baseline -- never as the product path.

What it is: a functional, state-dict-driven restatement (plain torch CPU ops, fp32 or fp64)
of the arithmetic the reference executes for one generator / discriminator forward and for
one full training iteration.  Every function cites the reference ``file:line`` it follows
(paths relative to the reference root).  The reference is pure Python on top of PyTorch
(``requirements.txt:5`` -- ``torch``, unpinned; this container: 2.10.0), so the arithmetic
lives in ATen; the oracle restates the *composition* and calls the same ATen primitives on CPU.

Parity pinning: the reference ships no tests or golden vectors for this path
(``test/test_Blocks.py:13-21`` and ``test/test_CustomLayers.py`` are empty), so the oracle is
pinned against outputs of the reference itself, executed on CPU in the build container by
``tests/golden/make_golden.py`` and committed as fixtures under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function below against those fixtures.

Parameters are passed as a flat ``dict`` with exactly the reference ``state_dict`` key names
(SURVEY.md A.4), so a reference checkpoint drives the oracle unchanged.
"""
from __future__ import annotations

import math
import random as _pyrandom
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

LRELU_SLOPE = 0.2            # models/GAN.py:67-68,150-151,346-347
SQRT2 = math.sqrt(2.0)


# --------------------------------------------------------------------------------------
# structure helpers
# --------------------------------------------------------------------------------------
def nf(stage: int, fmap_base: int = 8192, fmap_decay: float = 1.0, fmap_max: int = 512) -> int:
    """Feature-map count per stage (models/GAN.py:138-139, :323-324)."""
    return min(int(fmap_base / (2.0 ** (stage * fmap_decay))), fmap_max)


def he_w_mul(fan_in: int, gain: float, lrmul: float = 1.0) -> float:
    """Runtime weight multiplier under use_wscale=True (models/CustomLayers.py:84-91,120-127)."""
    return gain * fan_in ** (-0.5) * lrmul


# --------------------------------------------------------------------------------------
# primitive layers (models/CustomLayers.py)
# --------------------------------------------------------------------------------------
def pixel_norm(x: Tensor, eps: float = 1e-8) -> Tensor:
    """models/CustomLayers.py:22-23."""
    return x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + eps)


def leaky_relu(x: Tensor) -> Tensor:
    return F.leaky_relu(x, LRELU_SLOPE)


def activation(x: Tensor, kind: str = "lrelu") -> Tensor:
    """The two nonlinearities the reference names (models/GAN.py:67-68,150-151,346-347): 'lrelu' = LeakyReLU(0.2),
    'relu'.  (The reference maps 'relu' to the function torch.relu, which its nn.Sequential containers reject at
    construction; at layer level an nn.ReLU module works and is what tests/golden/flags.npz records.)"""
    return leaky_relu(x) if kind == "lrelu" else torch.relu(x)


class Flags:
    """The non-default layer options of the networks (models/GAN.py:105-108,303-305; config.py:60-75): LayerEpilogue
    stages, nonlinearity, blur filter.  Defaults = the reference's defaults."""

    def __init__(self, use_noise=True, use_pixel_norm=False, use_instance_norm=True, use_styles=True, act="lrelu",
                 blur_taps=(1.0, 2.0, 1.0)):
        self.use_noise, self.use_pixel_norm, self.use_instance_norm, self.use_styles = use_noise, use_pixel_norm, use_instance_norm, use_styles
        self.act, self.blur_taps = act, tuple(float(t) for t in blur_taps)


DEFAULT_FLAGS = Flags()


def upscale2d(x: Tensor) -> Tensor:
    """Nearest-neighbour x2 replicate (models/CustomLayers.py:27-36)."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def blur3(x: Tensor, taps: Sequence[float] = (1.0, 2.0, 1.0), normalize: bool = True) -> Tensor:
    """BlurLayer: depthwise outer(f, f) [normalised] correlation, zero padded by (K-1)//2 (models/CustomLayers.py:251-276;
    default f = [1,2,1]).  (Its ``flip`` option raises in the reference, :262, and is not restated.)"""
    k = torch.tensor(taps, dtype=torch.float32)               # the buffer is built in fp32 whatever the module dtype (:256-260)
    k = k[:, None] * k[None, :]
    if normalize:
        k = k / k.sum()
    k = k.to(x.dtype)[None, None].expand(x.shape[1], 1, -1, -1)
    return F.conv2d(x, k, padding=int((len(taps) - 1) / 2), groups=x.shape[1])


def downscale2d(x: Tensor) -> Tensor:
    """factor-2 Downscale2d == [0.5,0.5] (x) [0.5,0.5] stride-2 blur == 2x2 mean
    (models/CustomLayers.py:48-76; the two branches are the same arithmetic)."""
    return F.avg_pool2d(x, 2)


def eq_linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], gain: float, lrmul: float = 1.0) -> Tensor:
    """EqualizedLinear.forward with use_wscale=True (models/CustomLayers.py:99-103)."""
    w_mul = he_w_mul(weight.shape[1], gain, lrmul)
    b = None if bias is None else bias * lrmul
    return F.linear(x, weight * w_mul, b)


def fused_up_weight(w_scaled: Tensor) -> Tensor:
    """3x3 -> 4x4 transposed-conv kernel of the fused upscale path (models/CustomLayers.py:146-150)."""
    w = w_scaled.permute(1, 0, 2, 3)
    w = F.pad(w, [1, 1, 1, 1])
    return w[:, :, 1:, 1:] + w[:, :, :-1, 1:] + w[:, :, 1:, :-1] + w[:, :, :-1, :-1]


def fused_down_weight(w_scaled: Tensor) -> Tensor:
    """3x3 -> 4x4 stride-2 kernel of the fused downscale path (models/CustomLayers.py:159-162)."""
    w = F.pad(w_scaled, [1, 1, 1, 1])
    return (w[:, :, 1:, 1:] + w[:, :, :-1, 1:] + w[:, :, 1:, :-1] + w[:, :, :-1, :-1]) * 0.25


def eq_conv2d(x: Tensor, weight: Tensor, bias: Optional[Tensor], gain: float = SQRT2, *,
              up: bool = False, down: bool = False, blur_after: bool = False,
              blur_taps: Sequence[float] = (1.0, 2.0, 1.0)) -> Tensor:
    """EqualizedConv2d.forward, all five paths (models/CustomLayers.py:137-180).

    up:   conv0_up of a GSynthesisBlock (``intermediate`` = BlurLayer when blur_after)
    down: conv1_down of a DiscriminatorBlock
    The fused / non-fused switch follows the reference's size tests (:143, :158).
    """
    k = weight.shape[2]
    w = weight * he_w_mul(weight.shape[1] * k * k, gain)
    have_conv = False
    if up and min(x.shape[2:]) * 2 >= 128:                                   # :143-152
        x = F.conv_transpose2d(x, fused_up_weight(w), stride=2, padding=1)
        have_conv = True
    elif up:                                                                  # :153-154
        x = upscale2d(x)
    pool_after = False
    if down and min(x.shape[2:]) >= 128:                                      # :158-165
        x = F.conv2d(x, fused_down_weight(w), stride=2, padding=1)
        have_conv = True
    elif down:                                                                # :166-168
        pool_after = True
    if not have_conv and not blur_after and not pool_after:                   # :170-171
        return F.conv2d(x, w, bias, padding=k // 2)
    if not have_conv:                                                         # :172-173
        x = F.conv2d(x, w, None, padding=k // 2)
    if blur_after:                                                            # :175-176
        x = blur3(x, blur_taps)
    if pool_after:
        x = downscale2d(x)
    if bias is not None:                                                      # :178-179
        x = x + bias.view(1, -1, 1, 1)
    return x


def instance_norm(x: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.InstanceNorm2d defaults: biased variance, no affine (models/CustomLayers.py:232-233)."""
    mu = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps)


def layer_epilogue(x: Tensor, noise: Optional[Tensor], noise_weight: Optional[Tensor], style_w: Optional[Tensor],
                   style_b: Optional[Tensor], dlatent: Optional[Tensor], flags: Flags = DEFAULT_FLAGS) -> Tensor:
    """noise -> activation -> [pixel norm] -> [instance norm] -> [style mod] (models/CustomLayers.py:219-248; default flags
    of models/GAN.py:108: use_noise, use_instance_norm, use_styles, no pixel norm, LeakyReLU)."""
    if flags.use_noise:
        x = x + noise_weight.view(1, -1, 1, 1) * noise                        # :199
    x = activation(x, flags.act)
    if flags.use_pixel_norm:                                                  # :230-231
        x = pixel_norm(x)
    if flags.use_instance_norm:                                               # :232-233
        x = instance_norm(x)
    if not flags.use_styles:                                                  # :237-238,:246-247
        return x
    style = eq_linear(dlatent, style_w, style_b, gain=1.0)                    # :205-207,211
    c = x.shape[1]
    s = style.view(-1, 2, c, 1, 1)                                            # :213-214
    return x * (s[:, 0] + 1.0) + s[:, 1]                                      # :215


def minibatch_stddev(x: Tensor, group_size: int = 4) -> Tensor:
    """StddevLayer.forward with num_new_features=1 (models/CustomLayers.py:294-305)."""
    b, c, h, w = x.shape
    g = min(group_size, b)
    y = x.reshape(g, -1, 1, c, h, w)
    y = y - y.mean(0, keepdim=True)
    y = (y * y).mean(0, keepdim=True)
    y = (y + 1e-8) ** 0.5
    y = y.mean([3, 4, 5], keepdim=True).squeeze(3)
    y = y.expand(g, -1, -1, h, w).clone().reshape(b, 1, h, w)
    return torch.cat([x, y], dim=1)


def truncation_update(avg: Tensor, last: Tensor, beta: float = 0.995) -> Tensor:
    """models/CustomLayers.py:316-317."""
    return beta * avg + (1.0 - beta) * last


def truncation_apply(avg: Tensor, dlat: Tensor, psi: float = 0.7, max_layer: int = 8) -> Tensor:
    """models/CustomLayers.py:319-323."""
    interp = torch.lerp(avg.expand_as(dlat), dlat, psi)
    mask = (torch.arange(dlat.shape[1]) < max_layer).view(1, -1, 1)
    return torch.where(mask, interp, dlat)


# --------------------------------------------------------------------------------------
# networks (models/GAN.py, models/Blocks.py)
# --------------------------------------------------------------------------------------
def g_mapping(p: Params, z: Tensor, mapping_layers: int, prefix: str = "g_mapping.map.", act: str = "lrelu") -> Tensor:
    """GMapping.forward without the broadcast (models/GAN.py:72-96): PixelNorm, then
    mapping_layers x (EqualizedLinear lrmul=0.01 gain=sqrt2, LeakyReLU)."""
    x = pixel_norm(z)
    for i in range(mapping_layers):
        x = activation(eq_linear(x, p[f"{prefix}dense{i}.weight"], p[f"{prefix}dense{i}.bias"],
                                 gain=SQRT2, lrmul=0.01), act)
    return x


def _epi(p: Params, pre: str, x: Tensor, noise: Tensor, dlat: Tensor, flags: Flags = DEFAULT_FLAGS) -> Tensor:
    return layer_epilogue(x, noise, p.get(pre + "top_epi.noise.weight"), p.get(pre + "style_mod.lin.weight"),
                          p.get(pre + "style_mod.lin.bias"), dlat, flags)


def g_synthesis(p: Params, dlatents: Tensor, depth: int, alpha: float, noises: List[Tensor],
                prefix: str = "g_synthesis.", flags: Flags = DEFAULT_FLAGS) -> Tensor:
    """GSynthesis.forward, structure 'linear' (models/GAN.py:175-208) with InputBlock
    (models/Blocks.py:47-60) and GSynthesisBlock (models/Blocks.py:83-88).

    ``noises[2*i], noises[2*i+1]``: the [B,1,R,R] noise maps for layer pair i (resolution 4*2^i).
    """
    b = dlatents.shape[0]
    pre = prefix + "init_block."
    x = p[pre + "const"].expand(b, -1, -1, -1) + p[pre + "bias"].view(1, -1, 1, 1)   # Blocks.py:51-52
    x = _epi(p, pre + "epi1.", x, noises[0], dlatents[:, 0], flags)
    x = eq_conv2d(x, p[pre + "conv.weight"], p[pre + "conv.bias"])
    x = _epi(p, pre + "epi2.", x, noises[1], dlatents[:, 1], flags)

    def block(i: int, x: Tensor) -> Tensor:
        bp = f"{prefix}blocks.{i}."
        x = eq_conv2d(x, p[bp + "conv0_up.weight"], p[bp + "conv0_up.bias"], up=True, blur_after=True, blur_taps=flags.blur_taps)
        x = _epi(p, bp + "epi1.", x, noises[2 * (i + 1)], dlatents[:, 2 * (i + 1)], flags)
        x = eq_conv2d(x, p[bp + "conv1.weight"], p[bp + "conv1.bias"])
        return _epi(p, bp + "epi2.", x, noises[2 * (i + 1) + 1], dlatents[:, 2 * (i + 1) + 1], flags)

    def to_rgb(i: int, x: Tensor) -> Tensor:
        return eq_conv2d(x, p[f"{prefix}to_rgb.{i}.weight"], p[f"{prefix}to_rgb.{i}.bias"], gain=1.0)

    if depth == 0:
        return to_rgb(0, x)                                                    # GAN.py:204
    for i in range(depth - 1):                                                 # GAN.py:196-197
        x = block(i, x)
    residual = to_rgb(depth - 1, upscale2d(x))                                 # GAN.py:199 (nearest x2)
    straight = to_rgb(depth, block(depth - 1, x))                              # GAN.py:200
    return alpha * straight + (1 - alpha) * residual                           # GAN.py:202


def generator(p: Params, z: Tensor, depth: int, alpha: float, noises: List[Tensor], *,
              mapping_layers: int, num_layers: int, training: bool = True,
              latents2: Optional[Tensor] = None, mixing_cutoff: Optional[int] = None,
              truncation_psi: float = 0.7, truncation_cutoff: int = 8, dlatent_avg_beta: float = 0.995,
              flags: Flags = DEFAULT_FLAGS, labels: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """Generator.forward (models/GAN.py:254-297).  RNG is explicit: ``latents2`` / ``mixing_cutoff``
    are the style-mixing draws of :282 / :286-288 (None = mixing disabled).  Returns
    (images, new avg_latent or None)."""
    if labels is not None:                                                                 # GAN.py:264-268 (conditional)
        z = torch.cat([z, p["class_embedding.weight"][labels]], 1)
    dl = g_mapping(p, z, mapping_layers, act=flags.act).unsqueeze(1).expand(-1, num_layers, -1)   # GAN.py:98-99
    new_avg = None
    has_trunc = truncation_psi > 0 and "truncation.avg_latent" in p
    if training:
        if has_trunc:
            new_avg = truncation_update(p["truncation.avg_latent"], dl[0, 0].detach(), dlatent_avg_beta)  # :277-278
        if latents2 is not None:
            dl2 = g_mapping(p, latents2, mapping_layers, act=flags.act).unsqueeze(1).expand(-1, num_layers, -1)
            idx = torch.arange(num_layers).view(1, -1, 1)
            dl = torch.where(idx < mixing_cutoff, dl, dl2)                                 # :289
        if has_trunc:
            dl = truncation_apply(new_avg, dl, truncation_psi, truncation_cutoff)          # :292-293
    return g_synthesis(p, dl, depth, alpha, noises, flags=flags), new_avg


def draw_mixing(z_shape, depth: int, style_mixing_prob: float = 0.9) -> Tuple[Tensor, int]:
    """The RNG consumption of models/GAN.py:282-288 in the reference's order: CPU ``torch.randn``
    first, then ``random.random()`` and (only if < prob) ``random.randint``."""
    latents2 = torch.randn(z_shape)
    cur_layers = 2 * (depth + 1)
    cutoff = _pyrandom.randint(1, cur_layers) if _pyrandom.random() < style_mixing_prob else cur_layers
    return latents2, cutoff


def discriminator_block(p: Params, bp: str, x: Tensor, flags: Flags = DEFAULT_FLAGS) -> Tensor:
    """DiscriminatorBlock (models/Blocks.py:137-146): conv0 -> act -> blur -> conv1_down -> act."""
    x = activation(eq_conv2d(x, p[bp + "conv0.weight"], p[bp + "conv0.bias"]), flags.act)
    x = blur3(x, flags.blur_taps)
    return activation(eq_conv2d(x, p[bp + "conv1_down.weight"], p[bp + "conv1_down.bias"], down=True), flags.act)


def discriminator(p: Params, img: Tensor, depth: int, alpha: float, total_depth: int,
                  prefix: str = "", flags: Flags = DEFAULT_FLAGS, labels: Optional[Tensor] = None) -> Tensor:
    """Discriminator.forward, structure 'linear' (models/GAN.py:413-442), DiscriminatorBlock
    (models/Blocks.py:137-146), DiscriminatorTop (models/Blocks.py:117-134).  ``total_depth`` is
    ``self.depth`` = log2(resolution)-1; module lists are indexed from the highest resolution."""
    def from_rgb(i: int, x: Tensor) -> Tensor:
        return eq_conv2d(x, p[f"{prefix}from_rgb.{i}.weight"], p[f"{prefix}from_rgb.{i}.bias"])

    def block(i: int, x: Tensor) -> Tensor:
        return discriminator_block(p, f"{prefix}blocks.{i}.", x, flags)

    if labels is not None:                                                     # conditional: GAN.py:415-421 / :431-436
        idx = total_depth - depth - 1 if depth > 0 else total_depth - 1        # embeddings[-1] at depth 0
        emb = p[f"{prefix}embeddings.{idx}.weight"][labels].view(img.shape[0], -1, img.shape[2], img.shape[3])
        img = torch.cat([img, emb], dim=1)
    if depth > 0:
        residual = from_rgb(total_depth - depth, F.avg_pool2d(img, 2))         # GAN.py:423-424
        straight = block(total_depth - depth - 1, from_rgb(total_depth - depth - 1, img))  # :425-426
        x = alpha * straight + (1 - alpha) * residual                          # :427
        for i in range(total_depth - depth, total_depth - 1):                  # :429-430
            x = block(i, x)
    else:
        x = from_rgb(total_depth - 1, img)                                     # :438 (from_rgb[-1])
    fp = prefix + "final_block."
    x = minibatch_stddev(x)
    x = activation(eq_conv2d(x, p[fp + "conv.weight"], p[fp + "conv.bias"]), flags.act)
    x = x.reshape(x.shape[0], -1)                                              # View(-1): NCHW order
    x = activation(eq_linear(x, p[fp + "dense0.weight"], p[fp + "dense0.bias"], gain=SQRT2), flags.act)
    return eq_linear(x, p[fp + "dense1.weight"], p[fp + "dense1.bias"], gain=1.0)


# --------------------------------------------------------------------------------------
# training step (models/GAN.py:557-659, models/Losses.py:192-229, models/__init__.py:13-40)
# --------------------------------------------------------------------------------------
def progressive_down_sampling(real: Tensor, depth: int, alpha: float, total_depth: int) -> Tensor:
    """StyleGAN.__progressive_down_sampling, structure 'linear' (models/GAN.py:575-589)."""
    f = int(2 ** (total_depth - depth - 1))
    pf = int(2 ** (total_depth - depth))
    ds = F.avg_pool2d(real, f) if f > 1 else real
    prior = upscale2d(F.avg_pool2d(real, pf)) if depth > 0 else ds
    return alpha * ds + (1 - alpha) * prior


def images_u8_to_float(u8_hwc: Tensor, flip: Optional[Sequence[bool]] = None) -> Tensor:
    """uint8 [B,H,W,3] -> fp32 [B,3,H,W]: the reference's transform chain without Resize, data/transforms.py:27-32
    (``RandomHorizontalFlip(), ToTensor(), Normalize((.5,.5,.5),(.5,.5,.5))``) with the flip decisions given.

    The arithmetic lives in torchvision, which requirements.txt:4 names unpinned and which is NOT installed in this
    image -- so this function cannot be pinned against torchvision itself.  It is pinned bit-exactly
    against tests/golden/images_u8.npz (tests/golden/make_golden_images.py: PNG bytes decoded by PIL, then torchvision's
    published ``hflip`` / ``to_tensor`` / ``normalize`` restated operation by operation, independently of this file) and by
    known values (0 -> -1, 255 -> 1, 128 -> 1/255): tests/test_oracle_golden.py."""
    x = u8_hwc
    if flip is not None:
        x = torch.stack([img.flip(1) if f else img for img, f in zip(x, flip)])
    t = x.permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)
    return (t - 0.5) / 0.5


def r1_penalty(p: Params, real: Tensor, depth: int, alpha: float, total_depth: int) -> Tensor:
    """LogisticGAN.R1Penalty (models/Losses.py:197-211): SUM over batch and pixels of grad^2."""
    real = real.detach().requires_grad_(True)
    logit = discriminator(p, real, depth, alpha, total_depth)
    (g,) = torch.autograd.grad(logit, real, torch.ones_like(logit), create_graph=True, retain_graph=True)
    return (g * g).sum()


def logistic_d_loss(p: Params, real: Tensor, fake: Tensor, depth: int, alpha: float, total_depth: int,
                    r1_gamma: float = 10.0) -> Tensor:
    """LogisticGAN.dis_loss (models/Losses.py:213-224)."""
    r = discriminator(p, real, depth, alpha, total_depth)
    f = discriminator(p, fake, depth, alpha, total_depth)
    loss = F.softplus(f).mean() + F.softplus(-r).mean()
    if r1_gamma != 0.0:
        loss = loss + r1_penalty(p, real, depth, alpha, total_depth) * (r1_gamma * 0.5)
    return loss


def logistic_g_loss(p: Params, fake: Tensor, depth: int, alpha: float, total_depth: int) -> Tensor:
    """LogisticGAN.gen_loss (models/Losses.py:226-229)."""
    return F.softplus(-discriminator(p, fake, depth, alpha, total_depth)).mean()


def gan_dis_loss(kind: str, r: Tensor, f: Tensor) -> Tensor:
    """Discriminator loss heads of the non-default losses on the [B,1] predictions r = D(real), f = D(fake):
    'standard-gan' (models/Losses.py:107-126: (BCE(r,1) + BCE(f,0)) / 2 on the squeezed logits), 'hinge' (:141-148),
    'relativistic-hinge' (:159-174).  Pinned by tests/golden/losses.npz (the reference's classes on an identity D)."""
    if kind in ("standard-gan", "conditional-loss"):       # ConditionalGANLoss.dis_loss (models/Losses.py:61-88): the same head
        r, f = r.squeeze(), f.squeeze()
        return (F.binary_cross_entropy_with_logits(r, torch.ones_like(r)) + F.binary_cross_entropy_with_logits(f, torch.zeros_like(f))) / 2
    if kind == "hinge":
        return F.relu(1 - r).mean() + F.relu(1 + f).mean()
    if kind == "relativistic-hinge":
        return F.relu(1 - (r - f.mean())).mean() + F.relu(1 + (f - r.mean())).mean()
    raise KeyError(kind)


def gan_gen_loss(kind: str, r: Optional[Tensor], f: Tensor) -> Tensor:
    """Generator loss heads: 'standard-gan' BCE(f,1) (the evident intent of models/Losses.py:130-134, whose tuple
    unpacking of the [B,1] output cannot execute), 'hinge' -mean(f) (:150-151), 'relativistic-hinge' (:176-189)."""
    if kind in ("standard-gan", "conditional-loss"):       # ConditionalGANLoss.gen_loss (models/Losses.py:90-93)
        f = f.squeeze()
        return F.binary_cross_entropy_with_logits(f, torch.ones_like(f))
    if kind == "hinge":
        return -f.mean()
    if kind == "relativistic-hinge":
        return F.relu(1 + (r - f.mean())).mean() + F.relu(1 - (f - r.mean())).mean()
    raise KeyError(kind)


class AdamState:
    """torch.optim.Adam restated (models/GAN.py:529-533; defaults lr .003, betas (0,.99), eps 1e-8,
    no weight decay / amsgrad).  Parameters whose grad is None are skipped, as torch does."""

    def __init__(self, lr=0.003, beta1=0.0, beta2=0.99, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m: Dict[str, Tensor] = {}
        self.v: Dict[str, Tensor] = {}
        self.t: Dict[str, int] = {}

    def step(self, params: Params, grads: Dict[str, Optional[Tensor]]) -> None:
        for k, g in grads.items():
            if g is None:
                continue
            if k not in self.m:
                self.m[k] = torch.zeros_like(params[k]); self.v[k] = torch.zeros_like(params[k]); self.t[k] = 0
            self.t[k] += 1
            t = self.t[k]
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            params[k].data.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def clip_grad_norm(grads: Dict[str, Optional[Tensor]], max_norm: float = 10.0) -> float:
    """nn.utils.clip_grad_norm_ (models/GAN.py:651): global L2 norm, coef = max/(norm+1e-6) clamped to 1."""
    gs = [g for g in grads.values() if g is not None]
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in gs]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in gs:
        g.mul_(coef)
    return float(total)


def ema_update(shadow: Params, src: Params, beta: float, names: Sequence[str]) -> None:
    """update_average over named_parameters only (models/__init__.py:31-36)."""
    for k in names:
        shadow[k].data.copy_(beta * shadow[k].data + (1.0 - beta) * src[k].data)


def d_step(gp: Params, dp: Params, d_opt: AdamState, z: Tensor, real_full: Tensor, depth: int, alpha: float, *,
           total_depth: int, mapping_layers: int, noises: List[Tensor], latents2=None, mixing_cutoff=None,
           truncation_psi: float = 0.7, loss: str = "logistic", flags: Flags = DEFAULT_FLAGS,
           labels: Optional[Tensor] = None) -> Tuple[float, Dict[str, Optional[Tensor]]]:
    """StyleGAN.optimize_discriminator, d_repeats=1 (models/GAN.py:591-622).  ``loss``: 'logistic' (+R1) or one of the
    ``gan_dis_loss`` kinds."""
    real = progressive_down_sampling(real_full, depth, alpha, total_depth)
    fake, new_avg = generator(gp, z, depth, alpha, noises, mapping_layers=mapping_layers,
                              num_layers=2 * total_depth, latents2=latents2, mixing_cutoff=mixing_cutoff,
                              truncation_psi=truncation_psi, flags=flags, labels=labels)
    if new_avg is not None:
        gp["truncation.avg_latent"] = new_avg.detach()
    fake = fake.detach()
    names = [k for k, v in dp.items() if v.requires_grad]
    if loss == "logistic":
        assert flags is DEFAULT_FLAGS and labels is None
        loss = logistic_d_loss(dp, real, fake, depth, alpha, total_depth)
    else:
        loss = gan_dis_loss(loss, discriminator(dp, real, depth, alpha, total_depth, flags=flags, labels=labels),
                            discriminator(dp, fake, depth, alpha, total_depth, flags=flags, labels=labels))
    gl = torch.autograd.grad(loss, [dp[k] for k in names], allow_unused=True)
    grads = dict(zip(names, gl))
    d_opt.step(dp, grads)
    return float(loss.detach()), grads


def g_step(gp: Params, dp: Params, g_opt: AdamState, z: Tensor, depth: int, alpha: float, *,
           total_depth: int, mapping_layers: int, noises: List[Tensor], latents2=None, mixing_cutoff=None,
           truncation_psi: float = 0.7, shadow: Optional[Params] = None, ema_decay: float = 0.999,
           loss: str = "logistic", real_full: Optional[Tensor] = None, flags: Flags = DEFAULT_FLAGS,
           labels: Optional[Tensor] = None) -> Tuple[float, Dict[str, Optional[Tensor]]]:
    """StyleGAN.optimize_generator (models/GAN.py:624-659) incl. grad clip and EMA.  ``real_full`` is read by the
    relativistic loss only (:635-641: the real batch at the current depth)."""
    fake, new_avg = generator(gp, z, depth, alpha, noises, mapping_layers=mapping_layers,
                              num_layers=2 * total_depth, latents2=latents2, mixing_cutoff=mixing_cutoff,
                              truncation_psi=truncation_psi, flags=flags, labels=labels)
    if new_avg is not None:
        gp["truncation.avg_latent"] = new_avg.detach()
    names = [k for k, v in gp.items() if v.requires_grad]
    if loss == "logistic":
        assert flags is DEFAULT_FLAGS and labels is None
        loss = logistic_g_loss(dp, fake, depth, alpha, total_depth)
    else:
        r = None
        if loss == "relativistic-hinge":
            r = discriminator(dp, progressive_down_sampling(real_full, depth, alpha, total_depth), depth, alpha, total_depth, flags=flags)
        loss = gan_gen_loss(loss, r, discriminator(dp, fake, depth, alpha, total_depth, flags=flags, labels=labels))
    gl = torch.autograd.grad(loss, [gp[k] for k in names], allow_unused=True)
    grads = {k: (None if g is None else g.clone()) for k, g in zip(names, gl)}
    clip_grad_norm(grads, 10.0)
    g_opt.step(gp, grads)
    if shadow is not None:
        ema_update(shadow, gp, ema_decay, names)
    return float(loss.detach()), grads


# --------------------------------------------------------------------------------------
# progressive schedule (models/GAN.py:730-797) -- integer/float bookkeeping, bit-exact target
# --------------------------------------------------------------------------------------
def schedule(num_images: int, epochs: Sequence[int], batch_sizes: Sequence[int],
             fade_in_percentage: Sequence[float], total_depth: int, start_depth: int = 0,
             feedback_factor: int = 10, checkpoint_factor: int = 10, structure: str = "linear"):
    """Yield (depth, epoch, i, ticker, step, alpha, feedback, checkpoint_after_epoch) exactly as the
    loops of StyleGAN.train compute them (models/GAN.py:727-803).  ``total_batches`` is the
    drop_last DataLoader length (data/__init__.py:43-50)."""
    if structure == "fixed":
        start_depth = total_depth - 1                                                    # :727-728
    step = 1
    for d in range(start_depth, total_depth):                                           # :730
        ticker = 1                                                                       # :735
        total_batches = num_images // batch_sizes[d]
        for epoch in range(1, epochs[d] + 1):                                            # :741
            fade_point = int((fade_in_percentage[d] / 100) * epochs[d] * total_batches)  # :748-749
            for i in range(1, total_batches + 1):                                        # :751
                alpha = ticker / fade_point if ticker <= fade_point else 1               # :753
                feedback = (i % int(total_batches / feedback_factor + 1) == 0) or i == 1  # :774
                ckpt = (i == total_batches) and (epoch % checkpoint_factor == 0 or epoch == 1
                                                 or epoch == epochs[d])                  # :803
                yield (d, epoch, i, ticker, step, alpha, feedback, ckpt)
                ticker += 1                                                              # :796
                step += 1                                                                # :797


# --------------------------------------------------------------------------------------
# parameter construction with the reference key names / shapes
# --------------------------------------------------------------------------------------
def make_generator_params(resolution: int, mapping_layers: int = 8, latent_size: int = 512,
                          fmap_base: int = 8192, fmap_max: int = 512, truncation: bool = True,
                          dtype=torch.float32) -> Params:
    """Random-init generator parameters with the reference's shapes and init distributions
    (models/CustomLayers.py:92,128-129; models/Blocks.py:34-35; models/CustomLayers.py:188)."""
    rl2 = int(math.log2(resolution))
    p: Params = {}
    for i in range(mapping_layers):
        p[f"g_mapping.map.dense{i}.weight"] = torch.randn(512, latent_size if i == 0 else 512, dtype=dtype) / 0.01
        p[f"g_mapping.map.dense{i}.bias"] = torch.zeros(512, dtype=dtype)

    def epi(pre: str, c: int):
        p[pre + "top_epi.noise.weight"] = torch.zeros(c, dtype=dtype)
        p[pre + "style_mod.lin.weight"] = torch.randn(2 * c, 512, dtype=dtype)
        p[pre + "style_mod.lin.bias"] = torch.zeros(2 * c, dtype=dtype)

    c1 = nf(1, fmap_base, 1.0, fmap_max)
    pre = "g_synthesis.init_block."
    p[pre + "const"] = torch.ones(1, c1, 4, 4, dtype=dtype)
    p[pre + "bias"] = torch.ones(c1, dtype=dtype)
    epi(pre + "epi1.", c1)
    p[pre + "conv.weight"] = torch.randn(c1, c1, 3, 3, dtype=dtype)
    p[pre + "conv.bias"] = torch.zeros(c1, dtype=dtype)
    epi(pre + "epi2.", c1)
    p["g_synthesis.to_rgb.0.weight"] = torch.randn(3, c1, 1, 1, dtype=dtype)
    p["g_synthesis.to_rgb.0.bias"] = torch.zeros(3, dtype=dtype)
    for res in range(3, rl2 + 1):
        cin, cout = nf(res - 2, fmap_base, 1.0, fmap_max), nf(res - 1, fmap_base, 1.0, fmap_max)
        bp = f"g_synthesis.blocks.{res - 3}."
        p[bp + "conv0_up.weight"] = torch.randn(cout, cin, 3, 3, dtype=dtype)
        p[bp + "conv0_up.bias"] = torch.zeros(cout, dtype=dtype)
        epi(bp + "epi1.", cout)
        p[bp + "conv1.weight"] = torch.randn(cout, cout, 3, 3, dtype=dtype)
        p[bp + "conv1.bias"] = torch.zeros(cout, dtype=dtype)
        epi(bp + "epi2.", cout)
        p[f"g_synthesis.to_rgb.{res - 2}.weight"] = torch.randn(3, cout, 1, 1, dtype=dtype)
        p[f"g_synthesis.to_rgb.{res - 2}.bias"] = torch.zeros(3, dtype=dtype)
    for v in p.values():
        v.requires_grad_(True)
    if truncation:
        p["truncation.avg_latent"] = torch.zeros(512, dtype=dtype)
    return p


def make_discriminator_params(resolution: int, fmap_base: int = 8192, fmap_max: int = 512,
                              dtype=torch.float32) -> Params:
    """Random-init discriminator parameters (models/GAN.py:350-378, models/Blocks.py:117-146)."""
    rl2 = int(math.log2(resolution))
    p: Params = {}
    i = 0
    for res in range(rl2, 2, -1):
        cin, cout = nf(res - 1, fmap_base, 1.0, fmap_max), nf(res - 2, fmap_base, 1.0, fmap_max)
        bp = f"blocks.{i}."
        p[bp + "conv0.weight"] = torch.randn(cin, cin, 3, 3, dtype=dtype)
        p[bp + "conv0.bias"] = torch.zeros(cin, dtype=dtype)
        p[bp + "conv1_down.weight"] = torch.randn(cout, cin, 3, 3, dtype=dtype)
        p[bp + "conv1_down.bias"] = torch.zeros(cout, dtype=dtype)
        p[f"from_rgb.{i}.weight"] = torch.randn(cin, 3, 1, 1, dtype=dtype)
        p[f"from_rgb.{i}.bias"] = torch.zeros(cin, dtype=dtype)
        i += 1
    c2 = nf(2, fmap_base, 1.0, fmap_max)
    p[f"from_rgb.{i}.weight"] = torch.randn(c2, 3, 1, 1, dtype=dtype)
    p[f"from_rgb.{i}.bias"] = torch.zeros(c2, dtype=dtype)
    p["final_block.conv.weight"] = torch.randn(c2, c2 + 1, 3, 3, dtype=dtype)
    p["final_block.conv.bias"] = torch.zeros(c2, dtype=dtype)
    p["final_block.dense0.weight"] = torch.randn(c2, c2 * 16, dtype=dtype)
    p["final_block.dense0.bias"] = torch.zeros(c2, dtype=dtype)
    p["final_block.dense1.weight"] = torch.randn(1, c2, dtype=dtype)
    p["final_block.dense1.bias"] = torch.zeros(1, dtype=dtype)
    for v in p.values():
        v.requires_grad_(True)
    return p


def noise_shapes(batch: int, depth: int) -> List[Tuple[int, int, int, int]]:
    """Shapes of the per-layer noise maps consumed by a forward at ``depth`` in execution order
    (models/CustomLayers.py:193)."""
    out = []
    for i in range(depth + 1):
        r = 4 * 2 ** i
        out += [(batch, 1, r, r), (batch, 1, r, r)]
    return out


def flops_per_image(depth: int, resolution: int = 1024, mapping_layers: int = 8) -> float:
    """Algorithmic conv+GEMM FLOPs (2/MAC) of one full logistic+R1 G+D iteration per image at
    progressive ``depth`` -- SURVEY.md section 6 table, recomputed from the closed form of
    SURVEY.md 8(d).  Used only for reporting."""
    table_1024_8 = {0: 1.60e9, 1: 13.10e9, 2: 59.05e9, 3: 242.87e9, 4: 518.72e9, 5: 692.85e9,
                    6: 867.73e9, 7: 1044.05e9, 8: 1223.27e9}
    return table_1024_8[depth]
