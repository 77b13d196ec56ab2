"""Networks and the training-step wrapper with the reference's surface (reference models/GAN.py):
``GMapping``, ``GSynthesis``, ``Generator``, ``Discriminator``, ``StyleGAN`` -- same constructor kwargs, forward
signatures, attribute names and ``state_dict`` keys, so ``train.py``-style drivers, the generate scripts and
checkpoints interchange.  All arithmetic of the forward/backward hot path runs in libsgx_hip.so.

Extra (non-reference) knobs are keyword-only and default to the reference behaviour:
``act_dtype`` (torch.float32 | torch.bfloat16 storage of activations between kernels; accumulation stays fp32).
"""
import contextlib
import copy
import datetime
import os
import random
import time
import timeit
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import Losses
from . import functional as F
from .Blocks import DiscriminatorBlock, DiscriminatorTop, GSynthesisBlock, InputBlock
from .CustomLayers import EqualizedConv2d, EqualizedLinear, PixelNormLayer, Truncation, act_code, apply_act
from .data import get_data_loader
from . import native
from .native import ACT_LRELU
from .optim import FusedAdam, clip_and_step, ema_update


class DeferredLoss:
    """What ``optimize_discriminator`` / ``optimize_generator`` return in place of the reference's ``loss.item()``
    (models/GAN.py:620,:659): the value is copied to pinned host memory asynchronously and the host only waits for it when
    somebody actually reads it (``float(x)``, ``"%f" % x``, ``f"{x:.3f}"``, arithmetic, comparison).  The training loop
    can thus enqueue the next half-iteration while the GPU is still finishing this one."""

    __slots__ = ("_host", "_event", "_scale", "_value", "__weakref__")

    def __init__(self, dev_scalar, scale=1.0, stream=None):
        """``stream``: the stream the scalar is produced on, if not the current one (data parallel: the update stream)."""
        self._scale, self._value = float(scale), None
        if dev_scalar.is_cuda:
            with torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext():
                t = dev_scalar.detach().reshape(1).float()
                view, slot = native.RING.take(4)                    # a pinned slot allocated once (no hipHostMalloc per loss)
                self._host = view.view(torch.float32)
                self._host.copy_(t, non_blocking=True)
                self._event = native.RING.mark(slot, owner=self)
        else:
            self._host, self._event = dev_scalar.detach().reshape(1).float().clone(), None

    def item(self):
        if self._value is None:
            if self._event is not None:
                self._event.synchronize()
            self._value = float(self._host[0]) * self._scale
        return self._value

    __float__ = item

    def __format__(self, spec): return format(self.item(), spec)
    def __repr__(self): return repr(self.item())
    def __add__(self, o): return self.item() + float(o)
    __radd__ = __add__
    def __sub__(self, o): return self.item() - float(o)
    def __rsub__(self, o): return float(o) - self.item()
    def __mul__(self, o): return self.item() * float(o)
    __rmul__ = __mul__
    def __truediv__(self, o): return self.item() / float(o)
    def __lt__(self, o): return self.item() < float(o)
    def __gt__(self, o): return self.item() > float(o)
    def __eq__(self, o): return self.item() == float(o)
    def __abs__(self): return abs(self.item())
    def __neg__(self): return -self.item()
    def __hash__(self): return hash(self.item())


# Generator output: to_rgb, the nearest upsample of the previous resolution's RGB image and the fade-in lerp as one kernel
# (functional.RgbOutFadeFn); 0 = the three separate passes (A/B).
FUSE_RGB_FADE = os.environ.get("SGX_FUSE_RGB_FADE", "1") != "0"


def _nonlinearity(name):
    """(activation module, gain) of the reference's ``nonlinearity`` argument (models/GAN.py:67-68,150-151,346-347).  The
    reference maps 'relu' to the FUNCTION ``torch.relu`` and then puts it into ``nn.Sequential`` / module attributes, which
    raises TypeError at construction (both networks; checked against the reference) -- the evident intent, a ReLU module,
    is what is built here."""
    return {'relu': (nn.ReLU(), np.sqrt(2)), 'lrelu': (nn.LeakyReLU(negative_slope=0.2), np.sqrt(2))}[name]


def _conv_weights(module):
    """The 3x3 convolution parameters of a network (cached list; the module tree is static)."""
    ws = module.__dict__.get("_sgx_conv_weights")
    if ws is None:
        ws = [m.weight for m in module.modules() if isinstance(m, EqualizedConv2d)]
        module.__dict__["_sgx_conv_weights"] = ws
    return ws


def update_average(model_tgt, model_src, beta):
    """EMA of the generator weights into the shadow copy -- reference models/__init__.py:13-40."""
    ema_update(model_tgt, model_src, beta)


class GMapping(nn.Module):
    """Mapping network -- reference models/GAN.py:37-100: PixelNorm, then ``mapping_layers`` x
    (EqualizedLinear lrmul 0.01 -> LeakyReLU 0.2), broadcast to [B, dlatent_broadcast, dlatent_size]."""

    def __init__(self, latent_size=512, dlatent_size=512, dlatent_broadcast=None, mapping_layers=8, mapping_fmaps=512,
                 mapping_lrmul=0.01, mapping_nonlinearity='lrelu', use_wscale=True, normalize_latents=True, **kwargs):
        super().__init__()
        self.latent_size = latent_size
        self.mapping_fmaps = mapping_fmaps
        self.dlatent_size = dlatent_size
        self.dlatent_broadcast = dlatent_broadcast
        act, gain = _nonlinearity(mapping_nonlinearity)
        self._act = act_code(act)
        layers = []
        if normalize_latents:
            layers.append(('pixel_norm', PixelNormLayer()))
        layers.append(('dense0', EqualizedLinear(self.latent_size, self.mapping_fmaps, gain=gain, lrmul=mapping_lrmul,
                                                 use_wscale=use_wscale)))
        layers.append(('dense0_act', act))
        for layer_idx in range(1, mapping_layers):
            fmaps_in = self.mapping_fmaps
            fmaps_out = self.dlatent_size if layer_idx == mapping_layers - 1 else self.mapping_fmaps
            layers.append(('dense{:d}'.format(layer_idx),
                           EqualizedLinear(fmaps_in, fmaps_out, gain=gain, lrmul=mapping_lrmul, use_wscale=use_wscale)))
            layers.append(('dense{:d}_act'.format(layer_idx), act))
        self.map = nn.Sequential(OrderedDict(layers))
        self.mapping_layers = mapping_layers
        for m in self.map.modules():
            if isinstance(m, EqualizedLinear):
                m.first_order_only = True                 # fused three-launch linear (no double backward reaches G)

    def forward(self, x):
        x = x.float()
        if hasattr(self.map, 'pixel_norm'):
            x = self.map.pixel_norm(x)
        for i in range(self.mapping_layers):
            dense = getattr(self.map, 'dense{:d}'.format(i))
            if self._act == ACT_LRELU:
                x = dense(x, act=ACT_LRELU)                                  # bias + LeakyReLU fused after the GEMM
            else:
                x = apply_act(dense(x), self._act)
        if self.dlatent_broadcast is not None:
            x = x.unsqueeze(1).expand(-1, self.dlatent_broadcast, -1)
        return x


class GSynthesis(nn.Module):
    """Synthesis network with progressive growing -- reference models/GAN.py:103-208."""

    def __init__(self, dlatent_size=512, num_channels=3, resolution=1024, fmap_base=8192, fmap_decay=1.0, fmap_max=512,
                 use_styles=True, const_input_layer=True, use_noise=True, nonlinearity='lrelu', use_wscale=True,
                 use_pixel_norm=False, use_instance_norm=True, blur_filter=None, structure='linear', act_dtype=torch.float32,
                 **kwargs):
        super().__init__()

        def nf(stage):
            return min(int(fmap_base / (2.0 ** (stage * fmap_decay))), fmap_max)

        self.structure = structure
        resolution_log2 = int(np.log2(resolution))
        assert resolution == 2 ** resolution_log2 and resolution >= 4
        self.depth = resolution_log2 - 1
        self.num_layers = resolution_log2 * 2 - 2
        self.num_styles = self.num_layers if use_styles else 1
        self.act_dtype = act_dtype
        act, gain = _nonlinearity(nonlinearity)

        self.init_block = InputBlock(nf(1), dlatent_size, const_input_layer, gain, use_wscale, use_noise, use_pixel_norm,
                                     use_instance_norm, use_styles, act)
        rgb_converters = [EqualizedConv2d(nf(1), num_channels, 1, gain=1, use_wscale=use_wscale)]
        blocks = []
        for res in range(3, resolution_log2 + 1):
            last_channels = nf(res - 2)
            channels = nf(res - 1)
            blocks.append(GSynthesisBlock(last_channels, channels, blur_filter, dlatent_size, gain, use_wscale, use_noise,
                                          use_pixel_norm, use_instance_norm, use_styles, act))
            rgb_converters.append(EqualizedConv2d(channels, num_channels, 1, gain=1, use_wscale=use_wscale))
        self.blocks = nn.ModuleList(blocks)
        self.to_rgb = nn.ModuleList(rgb_converters)
        self.temporaryUpsampler = lambda x: F.nchw_view(F.call(F.Up2Fn, F.nhwc(x), 1.0))

    def forward(self, dlatents_in, depth=0, alpha=0., labels_in=None):
        assert depth < self.depth, "Requested output depth cannot be produced"
        F.prepack(_conv_weights(self))                                      # all stale MFMA operand packs: one launch
        dl = self._styles(dlatents_in.float(), depth)                        # per-layer style vectors (or dlatents)
        dt = self.act_dtype
        nblocks = len(self.blocks) if self.structure == 'fixed' else depth
        prev_arena = F.NOISE_ARENA
        if any(m.noise is None for m in self._noise_layers()):
            b = dlatents_in.shape[0]                                         # one randn for all the per-layer noise maps
            F.NOISE_ARENA = F.NoiseArena(b * sum(2 * (4 << k) ** 2 for k in range(nblocks + 1)), dlatents_in.device)
        try:
            return self._forward(dl, dt, depth, alpha)
        finally:
            F.NOISE_ARENA = prev_arena

    def _styles(self, dlatents, depth):
        """Per-layer inputs of the epilogues: with the default flags, the style vectors of all active layers from ONE
        launch (``GroupedStyleFn``) wrapped as ``PreStyle``; otherwise the L contiguous dlatents (``SplitLayersFn``)."""
        nblocks = len(self.blocks) if self.structure == 'fixed' else depth
        epis = [self.init_block.epi1, self.init_block.epi2]
        for blk in self.blocks[:nblocks]:
            epis += [blk.epi1, blk.epi2]
        B = dlatents.shape[0]
        ok = (getattr(self.init_block, "const_input_layer", False) and B <= 32 and dlatents.shape[2] % 64 == 0
              and all(getattr(e, "_fusable", False) and e.style_mod is not None for e in epis))
        if not ok:
            return F.call(F.SplitLayersFn, dlatents)
        lins = [e.style_mod.lin for e in epis]
        meta = tuple((i, float(l.w_mul), float(l.b_mul)) for i, l in enumerate(lins))
        lm = dlatents.transpose(0, 1).contiguous()                           # [L, B, D], one copy
        styles = F.call(F.GroupedStyleFn, lm, meta, *[l.weight for l in lins], *[l.bias for l in lins])
        return [F.PreStyle(s) for s in styles] + [None] * (dlatents.shape[1] - len(styles))

    def _noise_layers(self):
        ls = self.__dict__.get("_sgx_noise_layers")
        if ls is None:
            from .CustomLayers import NoiseLayer
            ls = self.__dict__["_sgx_noise_layers"] = [m for m in self.modules() if isinstance(m, NoiseLayer)]
        return ls

    def _forward(self, dl, dt, depth, alpha):
        if self.structure == 'fixed':
            x = self.init_block.forward_nhwc(dl[0:2], dt)
            rgb, last = self.to_rgb[-1], (self.blocks[-1] if len(self.blocks) else None)
            fuse_last = (F.FUSE_EPI_RGB and last is not None and last.epi2._fusable and rgb.weight.shape[0] == 3
                         and F.epi_rgb_out_ok(rgb.weight.shape[1], self.act_dtype))
            for i, block in enumerate(self.blocks):
                if fuse_last and block is last:
                    # the last epilogue inside to_rgb (functional.EpiRgbOutFn; the same kernel the 'linear' structure ends with)
                    y2, (ebias, noise, nw, style, part) = block.forward_nhwc(x, dl[2 * (i + 1):2 * (i + 2)], defer_epi2=True)
                    return F.nchw_view(F.call(F.EpiRgbOutFn, y2, ebias, noise, nw, style, rgb.weight, rgb.scaled_bias(), float(rgb.w_mul), None, 1.0, part))
                x = block.forward_nhwc(x, dl[2 * (i + 1):2 * (i + 2)])
            images = rgb.forward_nhwc(x)
        elif self.structure == 'linear':
            x = self.init_block.forward_nhwc(dl[0:2], dt)
            if depth > 0:
                for i, block in enumerate(self.blocks[:depth - 1]):
                    x = block.forward_nhwc(x, dl[2 * (i + 1):2 * (i + 2)])
                # reference GAN.py:199 applies to_rgb AFTER the nearest upsample; a 1x1 conv commutes with
                # replication, so convert at the low resolution (4x fewer bytes) and upsample the RGB image.
                prev = self.to_rgb[depth - 1]
                if F.RGB_FORK and prev.weight.shape[0] == 3 and prev.weight.shape[1] == x.shape[3]:
                    # x feeds the newest block AND this to_rgb: RgbOutForkFn joins its two gradients in to_rgb's data-gradient pass
                    x, low = F.call(F.RgbOutForkFn, x, prev.weight, prev.scaled_bias(), prev.w_mul)
                else:
                    low = prev.forward_nhwc(x)                                                    # RGB at the previous resolution
                rgb = self.to_rgb[depth]
                last = self.blocks[depth - 1]
                if (F.FUSE_EPI_RGB and FUSE_RGB_FADE and last.epi2._fusable and rgb.weight.shape[0] == 3 and low.dtype == torch.float32
                        and F.epi_rgb_out_ok(rgb.weight.shape[1], x.dtype)):
                    # the last epilogue inside to_rgb (+ upsample of ``low`` + fade-in lerp): one pass over conv1's output
                    y2, (ebias, noise, nw, style, part) = last.forward_nhwc(x, dl[2 * depth:2 * (depth + 1)], defer_epi2=True)
                    return F.nchw_view(F.call(F.EpiRgbOutFn, y2, ebias, noise, nw, style, rgb.weight, rgb.scaled_bias(), float(rgb.w_mul), low,
                                              alpha if isinstance(alpha, torch.Tensor) else float(alpha), part))
                xs = last.forward_nhwc(x, dl[2 * depth:2 * (depth + 1)])
                if FUSE_RGB_FADE and rgb.weight.shape[0] == 3 and low.dtype == torch.float32:
                    # to_rgb + upsample of ``low`` + fade-in lerp in one pass over xs (GAN.py:199-202)
                    images = F.call(F.RgbOutFadeFn, xs, rgb.weight, rgb.scaled_bias(), float(rgb.w_mul), low, alpha)
                else:
                    images = F.fade(rgb.forward_nhwc(xs), F.call(F.Up2Fn, low, 1.0), alpha)       # GAN.py:202
            else:
                images = self.to_rgb[0].forward_nhwc(x)
        else:
            raise KeyError("Unknown structure: ", self.structure)
        return F.nchw_view(images)


class Generator(nn.Module):
    """Style-based generator -- reference models/GAN.py:211-297 (mapping, W moving average, style mixing,
    truncation, synthesis).  The RNG draws of the mixing step happen in the reference's order (:282-288)."""

    def __init__(self, resolution, latent_size=512, dlatent_size=512, conditional=False, n_classes=0, truncation_psi=0.7,
                 truncation_cutoff=8, dlatent_avg_beta=0.995, style_mixing_prob=0.9, **kwargs):
        super().__init__()
        if conditional:
            assert n_classes > 0, "Conditional generation requires n_class > 0"
            self.class_embedding = nn.Embedding(n_classes, latent_size)
            latent_size *= 2
        self.conditional = conditional
        self.style_mixing_prob = style_mixing_prob
        self.num_layers = (int(np.log2(resolution)) - 1) * 2
        self.g_mapping = GMapping(latent_size, dlatent_size, dlatent_broadcast=self.num_layers, **kwargs)
        self.g_synthesis = GSynthesis(resolution=resolution, **kwargs)
        if truncation_psi > 0:
            self.truncation = Truncation(avg_latent=torch.zeros(dlatent_size), max_layer=truncation_cutoff,
                                         threshold=truncation_psi, beta=dlatent_avg_beta)
        else:
            self.truncation = None

    _mixing_override = None          # (latents2 device tensor, cutoff int or device tensor) supplied by a step graph
    # Data parallel: called with the W-average buffer right after ``truncation.update`` and BEFORE the truncation is applied.
    # The reference updates with GLOBAL sample 0 (:278) = rank 0's local sample 0, so every rank must truncate THIS forward's
    # dlatents with rank 0's value (StyleGAN installs a broadcast from rank 0 here).
    _avg_latent_hook = None

    def draw_mixing_host(self, shape, depth):
        """The host-side random draws of style mixing, in the reference's order: (latents2 CPU tensor, cutoff int)."""
        latents2 = torch.randn(shape)                                                    # CPU RNG first (:282)
        cur_layers = 2 * (depth + 1)
        cutoff = random.randint(1, cur_layers) if random.random() < self.style_mixing_prob else cur_layers
        return latents2, cutoff

    def draw_mixing(self, shape, depth, device):
        latents2, cutoff = self.draw_mixing_host(shape, depth)
        if device.type == "cuda":
            # through persistent pinned staging (native.upload): no host wait on the queue, and no hipHostMalloc per call either
            # (``.pin_memory()`` here cost 21 ms per call at batch 128 whenever the host ran ahead of the GPU)
            latents2 = native.upload(latents2.contiguous(), device)[0]
        return latents2, cutoff

    def forward(self, latents_in, depth, alpha, labels_in=None):
        if self.conditional:
            assert labels_in is not None, "Conditional discriminatin requires labels"
            latents_in = torch.cat([latents_in, self.class_embedding(labels_in)], 1)
        mixing = self.training and self.style_mixing_prob is not None and self.style_mixing_prob > 0
        if mixing:
            # host RNG (CPU randn first, then the mixing coin and cutoff: models/GAN.py:282-287) -- or the values a
            # captured-step wrapper drew in that same order and staged in static device tensors.  The mapping network is
            # row-wise, so both latent sets go through it as ONE batch of 2B rows (half the launches, same rows).
            latents2, mixing_cutoff = self._mixing_override or self.draw_mixing(latents_in.shape, depth, latents_in.device)
            both = self.g_mapping(torch.cat([latents_in, latents2.to(latents_in.dtype)], 0))
            dlatents_in, dlatents2 = both[:latents_in.shape[0]], both[latents_in.shape[0]:]
        else:
            dlatents_in = self.g_mapping(latents_in)
        if self.training:
            if self.truncation is not None:
                self.truncation.update(dlatents_in[0, 0].detach())                       # sample 0 only (:278)
                if self._avg_latent_hook is not None:
                    self._avg_latent_hook(self.truncation.avg_latent)
            if mixing:
                layer_idx = torch.arange(self.num_layers, device=latents_in.device).view(1, -1, 1)
                dlatents_in = torch.where(layer_idx < mixing_cutoff, dlatents_in, dlatents2)
            if self.truncation is not None:
                dlatents_in = self.truncation(dlatents_in)
        return self.g_synthesis(dlatents_in, depth, alpha)


class Discriminator(nn.Module):
    """Progressive discriminator -- reference models/GAN.py:300-444."""

    def __init__(self, resolution, num_channels=3, conditional=False, n_classes=0, fmap_base=8192, fmap_decay=1.0,
                 fmap_max=512, nonlinearity='lrelu', use_wscale=True, mbstd_group_size=4, mbstd_num_features=1,
                 blur_filter=None, structure='linear', act_dtype=torch.float32, **kwargs):
        super().__init__()
        if conditional:
            # reference :326-330: the label embedding joins the image as num_channels extra channels of the from_rgb input
            assert n_classes > 0, "Conditional Discriminator requires n_class > 0"
            num_channels *= 2
        embeddings = []

        def nf(stage):
            return min(int(fmap_base / (2.0 ** (stage * fmap_decay))), fmap_max)

        self.conditional = conditional
        self.mbstd_num_features = mbstd_num_features
        self.mbstd_group_size = mbstd_group_size
        self.structure = structure
        self.act_dtype = act_dtype
        resolution_log2 = int(np.log2(resolution))
        assert resolution == 2 ** resolution_log2 and resolution >= 4
        self.depth = resolution_log2 - 1
        act, gain = _nonlinearity(nonlinearity)

        blocks, from_rgb = [], []
        for res in range(resolution_log2, 2, -1):
            blocks.append(DiscriminatorBlock(nf(res - 1), nf(res - 2), gain=gain, use_wscale=use_wscale,
                                             activation_layer=act, blur_kernel=blur_filter))
            from_rgb.append(EqualizedConv2d(num_channels, nf(res - 1), kernel_size=1, gain=gain, use_wscale=use_wscale))
            if conditional:                                                 # :360-363
                r = 2 ** res
                embeddings.append(nn.Embedding(n_classes, (num_channels // 2) * r * r))
        if conditional:                                                     # :365-368
            embeddings.append(nn.Embedding(n_classes, (num_channels // 2) * 4 * 4))
            self.embeddings = nn.ModuleList(embeddings)
        self.blocks = nn.ModuleList(blocks)
        self.final_block = DiscriminatorTop(self.mbstd_group_size, self.mbstd_num_features, in_channels=nf(2),
                                            intermediate_channels=nf(2), gain=gain, use_wscale=use_wscale,
                                            activation_layer=act)
        from_rgb.append(EqualizedConv2d(num_channels, nf(2), kernel_size=1, gain=gain, use_wscale=use_wscale))
        self.from_rgb = nn.ModuleList(from_rgb)
        self.temporaryDownsampler = lambda x: F.nchw_view(F.call(F.Pool2Fn, F.nhwc(x), 0.25))

    def prepack(self, depth=None, img_shape=None):
        """Operand packs of every convolution whose weights changed since they were last packed (one launch), plus -- given the
        depth and the NHWC image shape -- the composed from_rgb/conv0 pack of the newest block.  The step calls this BEFORE it forks
        the fake branch to its own stream, so both branches find the packs made."""
        F.prepack(_conv_weights(self))
        if depth is not None and img_shape is not None and self.structure == 'linear' and depth > 0 and not self.conditional:
            top, top_rgb = self.blocks[self.depth - depth - 1], self.from_rgb[self.depth - depth - 1]
            if top.fused_from_rgb_ok(img_shape, top_rgb, self.act_dtype):
                F.rgb_packs(top.conv0.weight, top.conv0.w_mul, top_rgb.weight, top_rgb.w_mul, top_rgb.scaled_bias())

    def forward(self, images_in, depth, alpha=1., labels_in=None):
        assert depth < self.depth, "Requested output depth cannot be produced"
        img = F.nhwc(images_in, torch.float32)                              # [B,R,R,3] fp32
        self.prepack(depth, img.shape)                                      # all stale MFMA operand packs: one launch
        dt = self.act_dtype
        if self.conditional:
            # :395-400,:415-421,:431-436: embedding [B, 3*R*R] viewed as [B,3,R,R], concatenated to the image channels
            assert labels_in is not None, "Conditional Discriminator requires labels"
            idx = 0 if self.structure == 'fixed' else (self.depth - depth - 1 if depth > 0 else -1)
            b, r1, r2, _ = img.shape
            emb = self.embeddings[idx](labels_in).float().view(b, -1, r1, r2).permute(0, 2, 3, 1)
            img = torch.cat([img, emb], dim=3).contiguous()
        # LeakyReLU backward without a pass of its own along the block chain (Blocks.DiscriminatorBlock.forward_nhwc): a
        # block's final activation is un-done by the NEXT block's conv0 data-gradient kernel (mask in its store), the newest
        # block's by the fade-in lerp's backward (one scale-and-mask pass); the last block feeds the minibatch-stddev head
        # and keeps its own pass.
        fuse = os.environ.get("SGX_FUSE_ACT_BWD", "1") != "0"

        def chain(x, blocks, x_masked):
            for i, block in enumerate(blocks):
                defer = fuse and i < len(blocks) - 1 and block._act == ACT_LRELU
                x = block.forward_nhwc(x, x_masked=x_masked, defer_out=defer)
                x_masked = defer
            return x
        if self.structure == 'fixed':
            x = chain(self.from_rgb[0].forward_nhwc(img, out_dtype=dt), list(self.blocks), False)
        elif self.structure == 'linear':
            if depth > 0:
                # (1-alpha) of the residual branch rides in from_rgb's weight scale and bias when alpha is a host number (under
                # graph replay it is device memory): its backward then needs no scaling pass over the activation
                pre = fuse and not isinstance(alpha, torch.Tensor) and not self.conditional
                top, top_rgb = self.blocks[self.depth - depth - 1], self.from_rgb[self.depth - depth - 1]
                res_rgb = self.from_rgb[self.depth - depth]
                # (the image feeds the newest block AND, pooled, the residual branch: PoolForkFn joins its two gradients in one pass)
                if F.POOL_FORK:
                    img, pimg = F.call(F.PoolForkFn, img, 0.25)
                else:
                    pimg = F.call(F.Pool2Fn, img, 0.25)
                if fuse and top._act == ACT_LRELU and not self.conditional and F.fade_rgb_ok(res_rgb, top.conv1_down.weight.shape[0], dt):
                    # round 5: the residual as a recipe -- evaluated inside the store of the newest block's stride-2 convolution (with the lerp)
                    residual = F.RgbResidual(pimg, res_rgb, float(1 - alpha) if pre else 1.0, dt)
                else:
                    residual = res_rgb.forward_nhwc(pimg, out_dtype=dt, out_scale=float(1 - alpha) if pre else 1.0)
                # the fade-in lerp in the store of the newest block's stride-2 convolution (functional.ConvDownFadeFn) where alpha
                # is a host number (the residual then already carries its 1 - alpha) and the shape has that kernel
                fade_arg = None
                if fuse and top._act == ACT_LRELU and not self.conditional:
                    # (alpha in device memory -- graph replay: the kernel reads [alpha, 1 - alpha] itself and the residual is not prescaled)
                    fade_arg = (residual, alpha, None) if isinstance(alpha, torch.Tensor) else ((residual, float(alpha), 1.0) if pre else None)
                if not self.conditional and top.fused_from_rgb_ok(img.shape, top_rgb, dt):
                    # from_rgb and the newest block's conv0 (no activation between them) as ONE convolution of the image, with
                    # the LeakyReLU and the blur in its store (functional.RgbConvBlurFn)
                    out = top.forward_from_image(img, top_rgb, defer_out=fuse, fade=fade_arg)
                else:
                    out = top.forward_nhwc(top_rgb.forward_nhwc(img, out_dtype=dt), defer_out=fuse, fade=fade_arg)
                straight, lerped = out if fade_arg is not None else (out, False)
                if not lerped and isinstance(residual, F.RgbResidual):
                    residual = residual.materialize()                                    # (no lerp-in-the-store kernel for this shape)
                x = straight if lerped else F.fade(straight, residual, alpha,            # GAN.py:427
                                                   a_act=fuse and top._act == ACT_LRELU, b_prescaled=pre)
                x = chain(x, list(self.blocks[(self.depth - depth):]), False)
            else:
                x = self.from_rgb[-1].forward_nhwc(img, out_dtype=dt)
        else:
            raise KeyError("Unknown structure: ", self.structure)
        return self.final_block.forward_nhwc(x)


class StyleGAN:
    """Wrapper around the Generator and the Discriminator: optimizers, loss, EMA and the two optimisation steps --
    reference models/GAN.py:447-659 -- and ``train``, the progressive-growing schedule loop (:682-826)."""

    def __init__(self, structure, resolution, num_channels, latent_size, g_args, d_args, g_opt_args, d_opt_args,
                 conditional=False, n_classes=0, loss="relativistic-hinge", drift=0.001, d_repeats=1, use_ema=False,
                 ema_decay=0.999, device=torch.device("cpu"), act_dtype=torch.float32, data_parallel=None,
                 use_graphs=False):
        assert structure in ['fixed', 'linear']
        if conditional:
            assert n_classes > 0, "Conditional GANs require n_classes > 0"
        self.structure = structure
        self.depth = int(np.log2(resolution)) - 1
        self.latent_size = latent_size
        self.device = torch.device(device)
        self.d_repeats = d_repeats
        self.conditional = conditional
        self.n_classes = n_classes
        self.use_ema = use_ema
        self.ema_decay = ema_decay
        self.dp = data_parallel                       # parallel.DataParallelGroup or None
        # hipGraph replay of the two half-iterations (one graph per (kind, depth, shapes)); eager when off, with
        # data parallelism, with labels or with d_repeats != 1
        self.use_graphs = bool(use_graphs)
        self._step_graphs = {}
        # Stream structure of a half-iteration (A/B and per-box calibration, bench.py): the fake branch of the D step on an
        # auxiliary stream / the weight-gradient kernels on a side stream.  Extra streams buy GPU overlap of the latency-bound
        # low-resolution kernels and cost host time per fork (measured on one box at batch 4, eager: 19.7 ms host-bound with
        # both, 16.0 / 16.1 with one of them, 16.3 with neither; no effect at batch 32) -- which wins depends on the host.
        self.aux_stream = os.environ.get("SGX_AUX_STREAM", "1") not in ("0", "")
        self.param_stream = os.environ.get("SGX_PARAM_STREAM", "1") not in ("0", "")
        if self.device.type != "cuda":
            raise RuntimeError("stylegan.pytorch_amd runs on MI355X only: device must be a cuda (ROCm) device; the "
                               "reference's CPU path is the oracle, not a fallback of this package")

        self.gen = Generator(num_channels=num_channels, resolution=resolution, structure=self.structure,
                             conditional=self.conditional, n_classes=self.n_classes, act_dtype=act_dtype,
                             **g_args).to(self.device)
        self.dis = Discriminator(num_channels=num_channels, resolution=resolution, structure=self.structure,
                                 conditional=self.conditional, n_classes=self.n_classes, act_dtype=act_dtype,
                                 **d_args).to(self.device)
        self.__setup_gen_optim(**g_opt_args)
        self.__setup_dis_optim(**d_opt_args)
        self.drift = drift
        self.loss = self.__setup_loss(loss)
        if self.use_ema:
            self.gen_shadow = copy.deepcopy(self.gen)
            self.ema_updater = update_average
            self.ema_updater(self.gen_shadow, self.gen, beta=0)
        self._grad_buckets = {}                       # (kind, depth) -> dist.GradBuckets of that network's active parameters
        if self.dp is not None and self.gen.truncation is not None:
            self.gen._avg_latent_hook = lambda buf: self.dp.broadcast(buf, src=0)     # (after the deepcopy: the shadow has none)

    def __setup_gen_optim(self, learning_rate, beta_1, beta_2, eps):
        self.gen_optim = FusedAdam(self.gen.parameters(), lr=learning_rate, betas=(float(beta_1), float(beta_2)), eps=eps)

    def __setup_dis_optim(self, learning_rate, beta_1, beta_2, eps):
        self.dis_optim = FusedAdam(self.dis.parameters(), lr=learning_rate, betas=(float(beta_1), float(beta_2)), eps=eps)

    def __setup_loss(self, loss):
        if isinstance(loss, str):
            loss = loss.lower()
            # data parallel: every head is a batch mean -> 1/world under the gradient all-reduce(SUM)
            mean_scale = 1.0 / self.dp.world_size if self.dp is not None else 1.0
            if self.conditional:                                            # reference :548-551
                assert loss in ["conditional-loss"]
                return Losses.ConditionalGANLoss(self.dis, mean_scale=mean_scale)
            assert loss in ["logistic", "hinge", "standard-gan", "relativistic-hinge"], "Unknown loss function"
            if loss == "logistic":
                return Losses.LogisticGAN(self.dis, mean_scale=mean_scale)
            if loss == "hinge":
                return Losses.HingeGAN(self.dis, mean_scale=mean_scale)
            if loss == "standard-gan":
                return Losses.StandardGAN(self.dis, mean_scale=mean_scale)
            # the relativistic loss subtracts the mean prediction of the WHOLE batch: a differentiable all-reduce
            return Losses.RelativisticAverageHingeGAN(self.dis, mean_scale=mean_scale,
                                                      batch_mean=self.dp.global_mean if self.dp is not None else None)
        return loss

    def progressive_down_sampling(self, real_batch, depth, alpha):
        """Real images at the current depth with the fade-in blend -- reference models/GAN.py:557-589."""
        if self.structure == 'fixed':
            return real_batch
        x = F.nhwc(real_batch, torch.float32)
        levels = self.depth - depth - 1                         # AvgPool2d(2**levels) == levels x (2x2 mean)
        for _ in range(levels):
            x = F.call(F.Pool2Fn, x, 0.25)
        if depth > 0:
            if FUSE_RGB_FADE and x.shape[3] == 3 and x.dtype == torch.float32 and not x.requires_grad:
                x = F.downsample_fade_rgb(x, alpha)                    # pool, upsample and lerp in one pass (:575-586)
            else:
                prior = F.call(F.Up2Fn, F.call(F.Pool2Fn, x, 0.25), 1.0)
                x = F.fade(x, prior, alpha)
        else:
            x = F.fade(x, x, alpha)                                       # prior == current at depth 0 (:583-584)
        return F.nchw_view(x)

    # name-mangled alias so code written against the reference's private helper keeps working
    _StyleGAN__progressive_down_sampling = progressive_down_sampling

    def _zero_grads(self, kind, depth):
        """``optim.zero_grad()`` of the reference step (:616,:648).  Data parallel, from the second iteration at a depth on:
        the gradients of the network's active parameters are views into flat buckets (dist.GradBuckets), zero-filled here and
        accumulated into by the backward, so that the all-reduce runs on the buckets in place."""
        optim = self.dis_optim if kind == "d" else self.gen_optim
        optim.zero_grad()                                            # set_to_none: inactive resolutions keep grad None
        gb = self._grad_buckets.get((kind, int(depth))) if self.dp is not None else None
        if gb is not None:
            gb.attach()

    def _sched_begin(self, kind, depth, param_stream):
        """Data parallel, eager: bucket-level overlap of the gradient all-reduce with this backward (dist.BucketScheduler).  The
        first backward at a depth RECORDS how often and in which order the parameters' gradients are written; later ones fire
        a bucket's all-reduce as soon as its last gradient is final."""
        from . import dist as D
        self.__dict__["_sched_active"] = None
        if self.dp is None or not self.dp.overlap_buckets or torch.cuda.is_current_stream_capturing():
            return
        if F.GRAD_NOTE is None:
            F.GRAD_NOTE = D.note_grad_write
        net = self.dis if kind == "d" else self.gen
        if not net.__dict__.get("_sgx_grad_hooks"):
            D.install_grad_hooks(net.parameters())
            net.__dict__["_sgx_grad_hooks"] = True
        scheds = self.__dict__.setdefault("_bucket_scheds", {})
        key = (kind, int(depth))
        sched = scheds.get(key)
        gb = self._grad_buckets.get(key)
        if sched is None or (not sched.recording and (gb is None or sched.gb is not gb or not gb.attached())):
            sched = scheds[key] = D.BucketScheduler(self.dp)   # (re-)record
        if not sched.recording:
            sched.begin(param_stream)
        self.__dict__["_sched_active"] = sched
        D.set_active_scheduler(sched)

    def _sched_abort(self):
        """The backward raised: no scheduler may stay installed (a later, unrelated backward would feed it)."""
        if self.dp is not None:
            from . import dist as D
            D.set_active_scheduler(None)
            self.__dict__["_sched_active"] = None

    def _note_active_grads(self, kind, depth):
        """After a backward: remember the active set of this (network, depth) as a flat bucket layout for the next iteration
        (in gradient-ready order when the backward was recorded by a BucketScheduler)."""
        if self.dp is None:
            return
        from . import dist as D
        D.set_active_scheduler(None)
        sched = self.__dict__.get("_sched_active")
        net = self.dis if kind == "d" else self.gen
        active = [p for p in net.parameters() if p.grad is not None]
        gb = self._grad_buckets.get((kind, int(depth)))
        recorded = sched is not None and sched.recording
        if active and (gb is None or not gb.matches(active) or recorded) and not torch.cuda.is_current_stream_capturing():
            for key in [k for k in self._grad_buckets if k[0] == kind and k[1] != int(depth)
                        and not any(g.depth == k[1] and g.graph is not None for g in self._step_graphs.values())]:
                del self._grad_buckets[key]                    # progressive growing moves on: drop the previous depth's ~100 MB
                self.__dict__.get("_bucket_scheds", {}).pop(key, None)
            new = None
            if sched is not None and sched.recording:
                new = sched.layout(self.dp.bucket_elems, only=active, canonical=list(net.parameters()))
                if new is None or not new.matches(active):     # a parameter got its gradient without a note: no early firing
                    new = None
                    self.__dict__.get("_bucket_scheds", {}).pop((kind, int(depth)), None)
            self._grad_buckets[(kind, int(depth))] = new if new is not None else D.GradBuckets(active, self.dp.bucket_elems)
            self.__dict__["_sched_active"] = None              # this iteration's gradients are not in the new buckets yet

    def _aux_stream(self):
        """Second compute stream for work that is independent of the main chain (the D-step generator forward)."""
        if not self.aux_stream:
            return None
        st = self.__dict__.get("_aux_compute_stream")
        if st is None:
            st = self.__dict__["_aux_compute_stream"] = torch.cuda.Stream(device=self.device)
            # the fake branch's gradients reach AccumulateGrad nodes that live on the main stream: intended (the engine
            # synchronises the two), so silence torch's once-per-process note about it
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
        return st

    def _param_stream(self, two_branches=False):
        """Side stream for the weight-gradient kernels (one per StyleGAN; they serialise among themselves, which is what
        makes accumulating into ``.grad`` from the two branches of the D step safe)."""
        if not self.param_stream:
            # With two backward branches (D step: fake on the auxiliary stream, real on the main one) the accumulating
            # launches of BOTH must still share one stream: the main stream itself (only the fake branch forks to it).
            return torch.cuda.current_stream() if two_branches else None
        # (Measured and not kept: leaving the side stream out of CAPTURED steps, where each fork/join is a cross-queue
        # edge of the replayed graph -- 17.65 vs 17.5-17.66 ms/step, no gain once the accumulation stays race-free.)
        st = self.__dict__.get("_param_side_stream")
        if st is None:
            st = self.__dict__["_param_side_stream"] = torch.cuda.Stream(device=self.device)
        return st

    def _graphable(self, labels):
        # with data parallelism the all-reduce stays eager between two graphs; the W-average broadcast cannot
        return (self.use_graphs and labels is None and self.d_repeats == 1 and self.structure == "linear"
                and (self.dp is None or (self.gen.truncation is None                     # (neither can the relativistic
                                         and not isinstance(self.loss, Losses.RelativisticAverageHingeGAN))))  # loss's mean)

    def _d_grads(self, noise, real_batch, depth, alpha, labels=None):
        """Discriminator half-iteration, part 1: losses and local gradients; returns the (device) loss."""
        real_samples = self.progressive_down_sampling(real_batch, depth, alpha)

        def make_fakes():
            self._wait_update("g")                    # data parallel: G's all-reduce + Adam + EMA may still be in flight
            with torch.no_grad():                     # the reference builds and drops this graph (.detach(), :607)
                out = self.gen(noise, depth, alpha, labels)
            return out

        # The generator forward that makes the fakes and the D(real) forward are independent chains of (at batch 4, mostly
        # latency-bound) kernels: the first is issued on an auxiliary stream -- in the reference's program order, so the
        # host RNG draws keep their order -- and the main stream only joins it when LogisticGAN asks for the fakes, after
        # D(real).  Other losses get the tensor up front.
        aux = self._aux_stream()
        lazy = isinstance(self.loss, Losses.LogisticGAN)
        self._wait_update("d")
        if aux is not None and lazy:
            # the whole fake branch -- generator forward AND D(fake) forward, hence also D(fake)'s backward, which autograd
            # runs on the stream of its forward -- lives on the auxiliary stream; the real branch (D(real), the R1
            # gradient pass, their backward) on the main one.  D's operand packs are made before the fork.
            self.dis.prepack(depth, (real_samples.shape[0], real_samples.shape[2], real_samples.shape[3], real_samples.shape[1]))
            cur = torch.cuda.current_stream()
            aux.wait_stream(cur)
            with torch.cuda.stream(aux):
                f_preds = self.dis(make_fakes(), depth, alpha)

            def fake_logits():
                cur.wait_stream(aux)
                f_preds.record_stream(cur)
                return f_preds
            loss = self.loss.dis_loss(real_samples, None, depth, alpha, fake_logits=fake_logits)
        elif self.conditional:                                              # reference :611-613
            loss = self.loss.dis_loss(real_samples, make_fakes(), labels, depth, alpha)
        else:
            loss = self.loss.dis_loss(real_samples, make_fakes if lazy else make_fakes(), depth, alpha)
        self._zero_grads("d", depth)
        side = self._param_stream(two_branches=aux is not None and lazy)
        self._sched_begin("d", depth, side)
        # conv weight / bias gradients accumulate inside the finishing kernel, on a side stream next to the backward chain
        try:
            with F.accumulate_param_grads(), F.param_grad_stream(side):
                loss.backward()
        except BaseException:
            self._sched_abort()
            raise
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        self._note_active_grads("d", depth)
        return loss.detach()

    def _reduce(self, kind):
        if self.dp is None:
            return
        sched = self.__dict__.pop("_sched_active", None)
        if sched is not None and not sched.recording and sched.gb is not None and sched.gb.attached():
            try:
                sched.finish()                                       # most buckets were all-reduced during the backward
            except RuntimeError:
                # a gradient was written more often than recorded: this step's gradients are unusable (a bucket may have left
                # early).  Drop the schedule AND the layout of this (network, depth) so the next backward records afresh, then
                # let the caller see the error.
                for key in [k for k, v in self.__dict__.get("_bucket_scheds", {}).items() if v is sched]:
                    self._bucket_scheds.pop(key, None)
                    self._grad_buckets.pop(key, None)
                raise
            self.__dict__["_last_sched"] = sched
            return
        gb = next((g for (k, _), g in self._grad_buckets.items() if k == kind and g.attached()), None)
        if gb is not None:
            self.dp.all_reduce_buckets(gb)                           # the gradients live in the flat buckets: in place
        else:
            self.dp.all_reduce_grads((self.dis if kind == "d" else self.gen).parameters())

    def _d_reduce(self):
        self._reduce("d")

    def _d_update(self):
        self.dis_optim.step()

    def _g_grads(self, noise, real_batch, depth, alpha, labels=None):
        real_samples = None
        if not isinstance(self.loss, (Losses.LogisticGAN, Losses.HingeGAN, Losses.StandardGAN, Losses.ConditionalGANLoss)):
            real_samples = self.progressive_down_sampling(real_batch, depth, alpha)   # only the relativistic loss reads it
        self._wait_update("g")
        fake_samples = self.gen(noise, depth, alpha, labels)
        self._wait_update("d")                        # data parallel: D's all-reduce + Adam overlapped the G forward above
        # the reference also back-propagates into D's parameters here and discards the result at the next
        # dis_optim.zero_grad(); skipping those weight gradients changes no observable value
        d_all = self.__dict__.get("_sgx_dis_params")               # the module tree is static: walk it once
        if d_all is None:
            d_all = self.__dict__["_sgx_dis_params"] = list(self.dis.parameters())
        d_params = [p for p in d_all if p.requires_grad]
        for p in d_params:
            p.requires_grad_(False)
        try:
            if self.conditional:                                            # reference :644-646
                loss = self.loss.gen_loss(real_samples, fake_samples, labels, depth, alpha)
            else:
                loss = self.loss.gen_loss(real_samples, fake_samples, depth, alpha)
            self._zero_grads("g", depth)
            side = self._param_stream()
            self._sched_begin("g", depth, side)
            try:
                with F.accumulate_param_grads(), F.param_grad_stream(side):
                    loss.backward()
            except BaseException:
                self._sched_abort()
                raise
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            self._note_active_grads("g", depth)
        finally:
            for p in d_params:
                p.requires_grad_(True)
        return loss.detach()

    def _g_reduce(self):
        self._reduce("g")                                                             # before the clip: global norm

    def _g_update(self):
        clip_and_step(self.gen_optim, max_norm=10.)                                   # :651-652 without a host sync
        if self.use_ema:
            self.ema_updater(self.gen_shadow, self.gen, self.ema_decay)

    # Data parallel, eager mode: the gradient all-reduce and the parameter update of one network run on their own stream
    # while the main stream already works on the part of the next half-iteration that does not read those parameters
    # (D's update || the generator forward of the G step;  G's update + EMA || D(real) forward of the next D step).
    # Whoever reads the parameters next waits for the event (`_wait_update`).  Gradient tensors stay alive until the next
    # zero_grad of the same optimizer, which comes after that wait.
    def _async_update(self, kind, loss):
        """-> the GLOBAL loss (sum of the ranks' partial losses: the mean terms carry 1/N, the R1 term is a batch sum), a
        tensor produced on the update stream (``self._loss_stream`` tells ``optimize_*`` where to read it)."""
        upd = self.__dict__.get("_update_stream")
        if upd is None:
            upd = self.__dict__["_update_stream"] = torch.cuda.Stream(device=self.device)
        upd.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(upd):
            (self._d_reduce if kind == "d" else self._g_reduce)()
            loss.record_stream(upd)
            loss = self.dp.all_reduce_scalar(loss)
            (self._d_update if kind == "d" else self._g_update)()
        ev = torch.cuda.Event()
        ev.record(upd)
        self.__dict__.setdefault("_pending_updates", {})[kind] = ev
        self.__dict__["_loss_stream"] = upd
        return loss

    def _wait_update(self, kind):
        ev = self.__dict__.get("_pending_updates", {}).pop(kind, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _d_body(self, noise, real_batch, depth, alpha, labels=None):
        loss = self._d_grads(noise, real_batch, depth, alpha, labels)
        if self.dp is not None and not torch.cuda.is_current_stream_capturing():
            return self._async_update("d", loss)
        self._d_reduce()
        self._d_update()
        return loss

    def _g_body(self, noise, real_batch, depth, alpha, labels=None):
        loss = self._g_grads(noise, real_batch, depth, alpha, labels)
        if self.dp is not None and not torch.cuda.is_current_stream_capturing():
            return self._async_update("g", loss)
        self._g_reduce()
        self._g_update()
        return loss

    def _graphed(self, kind, noise, real_batch, depth, alpha):
        self._wait_update("d"); self._wait_update("g")          # leftovers of eager calls (the graphs update in line)
        key = (kind, int(depth), tuple(noise.shape), tuple(real_batch.shape), tuple(real_batch.stride()), real_batch.dtype,
               self.aux_stream, self.param_stream)                     # a captured graph bakes its stream structure in
        g = self._step_graphs.get(key)
        if g is None:
            g = self._step_graphs[key] = _StepGraph(self, kind, depth)
        return g.run(noise, real_batch, alpha)

    # What optimize_* return.  False (default): a Python ``float``, exactly as the reference's ``loss.item()`` (models/GAN.py:620,
    # :659) -- which waits for the GPU.  True: a ``DeferredLoss`` (async copy to pinned memory, the host only waits when the
    # value is read), so that a loop that reads the losses rarely -- ``train`` below sets it for its own duration, bench.py
    # sets it -- enqueues the next half-iteration while the GPU still runs this one.
    deferred_losses = False

    # True: the eager step takes the fade-in alpha from DEVICE memory, exactly as the hipGraph-replayed step must (a captured
    # graph cannot bake a per-iteration scalar into kernel arguments): the fade-in lerp reads [alpha, 1-alpha] from a device
    # tensor and from_rgb's (1-alpha) is not folded into its weights.  Same arithmetic as the replay, launch by launch -- what
    # the graph-vs-eager parity tests compare against (host alpha rounds the residual branch once instead of twice in bf16).
    alpha_on_device = False

    def _alpha_arg(self, alpha):
        if not self.alpha_on_device or isinstance(alpha, torch.Tensor):
            return alpha
        return native.upload(torch.tensor([float(alpha), 1.0 - float(alpha)], dtype=torch.float32), self.device)[0]

    def _loss_out(self, deferred):
        return deferred if self.deferred_losses else deferred.item()

    def optimize_discriminator(self, noise, real_batch, depth, alpha, labels=None):
        """One discriminator update -- reference models/GAN.py:591-622."""
        if self._graphable(labels):
            return self._loss_out(self._graphed("d", noise, real_batch, depth, alpha))
        loss_val = None
        alpha = self._alpha_arg(alpha)
        for _ in range(self.d_repeats):
            loss = self._d_body(noise, real_batch, depth, alpha, labels)
            if loss_val is not None and self.__dict__.get("_loss_stream") is not None:   # data parallel and d_repeats > 1
                torch.cuda.current_stream().wait_stream(self.__dict__.pop("_loss_stream"))
            loss_val = loss if loss_val is None else loss_val + loss
        return self._loss_out(DeferredLoss(loss_val, 1.0 / self.d_repeats, stream=self.__dict__.pop("_loss_stream", None)))

    def optimize_generator(self, noise, real_batch, depth, alpha, labels=None):
        """One generator update incl. gradient clipping and EMA -- reference models/GAN.py:624-659."""
        if self._graphable(labels):
            return self._loss_out(self._graphed("g", noise, real_batch, depth, alpha))
        loss = self._g_body(noise, real_batch, depth, self._alpha_arg(alpha), labels)
        return self._loss_out(DeferredLoss(loss, stream=self.__dict__.pop("_loss_stream", None)))

    # ------------------------------------------------------------------------------------------------------------
    # The progressive-growing schedule -- reference models/GAN.py:730-803.  Pure host arithmetic, kept in the reference's
    # own expression forms because the target is BIT-exact agreement of (depth, alpha, feedback ticks, checkpoint epochs)
    # with the reference loop (tests/test_train_schedule.py replays tests/golden/schedule.npz, recorded from it).
    @staticmethod
    def fade_point_of(fade_in_percentage, epochs, total_batches):
        return int((fade_in_percentage / 100) * epochs * total_batches)                  # :748-749

    @staticmethod
    def alpha_at(ticker, fade_point):
        return ticker / fade_point if ticker <= fade_point else 1                         # :753

    @staticmethod
    def is_feedback_batch(i, total_batches, feedback_factor):
        return i % int(total_batches / feedback_factor + 1) == 0 or i == 1               # :774

    @staticmethod
    def is_checkpoint_epoch(epoch, epochs, checkpoint_factor):
        return epoch % checkpoint_factor == 0 or epoch == 1 or epoch == epochs            # :803

    @staticmethod
    def create_grid(samples, scale_factor, img_file):
        """Sample sheet of one feedback tick -- reference models/GAN.py:660-680 (nearest upscale, then a square-ish grid of
        per-image min/max normalised tiles with a one-pixel border, written as PNG).  torchvision is not on the MI355X image:
        the grid is assembled here and written with PIL; without PIL the tensor is saved next to the requested name."""
        samples = samples.detach().float().cpu()
        if scale_factor > 1:
            samples = torch.nn.functional.interpolate(samples, scale_factor=scale_factor)
        n, c, h, w = samples.shape
        nrow = max(1, int(np.sqrt(n)))                                  # images per row, as save_image(nrow=...)
        ncol = (n + nrow - 1) // nrow
        pad = 1
        grid = torch.full((c, ncol * (h + pad) + pad, nrow * (w + pad) + pad), 128.0)
        for k in range(n):
            t = samples[k]
            lo, hi = float(t.min()), float(t.max())
            t = (t - lo) / max(hi - lo, 1e-5)
            y, x = (k // nrow) * (h + pad) + pad, (k % nrow) * (w + pad) + pad
            grid[:, y:y + h, x:x + w] = t
        try:
            from PIL import Image
            arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
            Image.fromarray(arr[:, :, 0] if c == 1 else arr).save(img_file)
        except ImportError:
            torch.save(grid, img_file + ".pt")

    # ``train`` replays captured half-iterations (see there): "auto" (default) decides per depth from three timed eager iterations --
    # replay where the host's enqueue time is the step time (the 4x4 / 8x8 depths at batch 128 and the top depths at batch 8 / 4 / 2:
    # +25 % / +24 % and +25 % / +54 % / +74 % img/s in profiles/r05_bench_sweep_*.json), eager launches where the GPU is the limit (replay
    # measured 7-19 % SLOWER at depth 2 and 4: a captured half-iteration re-packs every weight it uses); True / False (SGX_TRAIN_GRAPHS=1 / 0)
    train_graphs = {"0": False, "1": True}.get(os.environ.get("SGX_TRAIN_GRAPHS", "auto"), "auto")

    def train(self, dataset, num_workers, epochs, batch_sizes, fade_in_percentage, logger, output,
              num_samples=36, start_depth=0, feedback_factor=100, checkpoint_factor=1):
        """The reference's training driver (models/GAN.py:682-826), same signature and side effects: per depth a fresh
        shuffled drop-last loader at that depth's batch size, the fade-in ``alpha`` ticker, one discriminator and one
        generator update per batch, a loss line + sample sheet every feedback tick, checkpoints of G / D / both optimizers
        (/ the EMA shadow) at the checkpoint epochs.  The losses come back as ``DeferredLoss``; the host only waits for
        them on feedback ticks (where the log line formats them), so between ticks it runs ahead of the GPU."""
        assert self.depth <= len(epochs), "epochs not compatible with depth"
        assert self.depth <= len(batch_sizes), "batch_sizes not compatible with depth"
        assert self.depth <= len(fade_in_percentage), "fade_in_percentage not compatible with depth"
        self.gen.train()
        self.dis.train()
        if self.use_ema:
            self.gen_shadow.train()
        was_deferred, self.deferred_losses = self.deferred_losses, True        # the log line of a feedback tick reads them
        # Round 5: the loop REPLAYS its half-iterations as hipGraphs wherever the step can be captured (``_graphable``: unconditional,
        # d_repeats 1, 'linear' structure; per depth and batch shape, after two eager calls).  At the top depths the reference's schedule
        # shrinks the batch to 8 / 4 / 2 where the launch count is largest (567-689 launches per iteration), and the eager loop is bound by
        # the host's enqueue time there (bench.py --sweep); the replayed step is bound by the GPU.  Same arithmetic as
        # the eager step with ``alpha_on_device`` (tests/test_gpu_graphs.py).  ``train_graphs``: "auto" (per depth, measured), True, False.
        was_graphs = getattr(self, "use_graphs", False)                     # (a schedule-only stand-in has no launch state: tests/test_train_schedule.py)
        self.use_graphs = bool(was_graphs or self.train_graphs is True)
        self._train_forced_graphs = self.use_graphs                          # the caller (or SGX_TRAIN_GRAPHS=1) asked for replay: no per-depth probe
        try:
            self._train_loop(dataset, num_workers, epochs, batch_sizes, fade_in_percentage, logger, output, num_samples,
                             start_depth, feedback_factor, checkpoint_factor)
        finally:
            self.deferred_losses = was_deferred
            self.use_graphs = was_graphs

    def _train_loop(self, dataset, num_workers, epochs, batch_sizes, fade_in_percentage, logger, output, num_samples,
                    start_depth, feedback_factor, checkpoint_factor):
        t_begin = time.time()
        fixed_input = torch.randn(num_samples, self.latent_size).to(self.device)       # CPU RNG, as the reference (:719)
        fixed_labels = None
        if self.conditional:
            fixed_labels = torch.linspace(0, self.n_classes - 1, num_samples).to(torch.int64).to(self.device)
        logger.info("Starting the training process ... \n")
        if self.structure == 'fixed':
            start_depth = self.depth - 1
        step = 1
        for current_depth in range(start_depth, self.depth):
            current_res = np.power(2, current_depth + 2)
            # captured steps of the depths already trained are not needed again: give their private memory pools back
            graphs = getattr(self, "_step_graphs", {})
            for key in [k for k in graphs if k[1] != current_depth]:
                del graphs[key]
            # "auto": this depth starts with eager launches; iterations 2..4 are timed (host enqueue time against wall time incl. the GPU)
            probe = None
            if self.train_graphs == "auto" and getattr(self, "device", None) is not None and self.device.type == "cuda" \
                    and not getattr(self, "_train_forced_graphs", False):
                self.use_graphs = False
                probe = {"seen": 0, "n": 0, "enq": 0.0, "all": 0.0}
            logger.info("Currently working on depth: %d", current_depth + 1)
            logger.info("Current resolution: %d x %d" % (current_res, current_res))
            ticker = 1
            data = get_data_loader(dataset, batch_sizes[current_depth], num_workers)
            n_epochs = epochs[current_depth]
            for epoch in range(1, n_epochs + 1):
                t_epoch = timeit.default_timer()
                logger.info("Epoch: [%d]" % epoch)
                total_batches = len(data)
                fade_point = self.fade_point_of(fade_in_percentage[current_depth], n_epochs, total_batches)
                for i, batch in enumerate(data, 1):
                    alpha = self.alpha_at(ticker, fade_point)
                    if self.conditional:
                        images, labels = batch
                        labels = labels.to(self.device)
                    else:
                        images, labels = batch, None
                    images = images.to(self.device)
                    gan_input = torch.randn(images.shape[0], self.latent_size).to(self.device)
                    # the probe times ONLY the two optimize_* calls of a full-size iteration (the loader, the H2D copy, the feedback tick's
                    # loss read -- the losses are deferred inside train() -- and sample grid lie outside the window: host work that replay
                    # does not remove): per probed iteration the host enqueue time of the two calls, then their wall time including the GPU
                    probing = probe is not None and probe["seen"] >= 1 and images.shape[0] == batch_sizes[current_depth]
                    if probing:
                        torch.cuda.synchronize(); t_probe = time.perf_counter()
                    # a ragged last batch of an epoch runs eagerly: its own captured graph would hold a private pool for one use per epoch
                    ragged = self.use_graphs and images.shape[0] != batch_sizes[current_depth]
                    if ragged:
                        self.use_graphs = False
                    try:
                        dis_loss = self.optimize_discriminator(gan_input, images, current_depth, alpha, labels)
                        gen_loss = self.optimize_generator(gan_input, images, current_depth, alpha, labels)
                    finally:
                        if ragged:
                            self.use_graphs = True
                    if probe is not None:
                        probe["seen"] += 1
                    if probing:
                        t_enq = time.perf_counter() - t_probe
                        torch.cuda.synchronize()
                        probe["enq"] += t_enq; probe["all"] += time.perf_counter() - t_probe; probe["n"] += 1
                        if probe["n"] == 3:
                            t_enq, t_all = probe["enq"], probe["all"]
                            if self.dp is not None:
                                # ONE decision for all ranks (the slowest rank's times): a rank replaying [graph | all-reduce | update]
                                # next to a rank on the eager bucket schedule would issue different collectives
                                t_enq, t_all = self.dp.host_max([t_enq, t_all])
                            self.use_graphs = bool(t_enq >= 0.85 * t_all)          # the host is the limit: replay from here on
                            logger.info("Depth %d: %s launches (host enqueue %.1f ms of %.1f ms per iteration)" % (
                                current_depth + 1, "hipGraph replay of the captured half-iterations" if self.use_graphs else "eager",
                                t_enq / 3 * 1e3, t_all / 3 * 1e3))
                            probe = None
                    if self.is_feedback_batch(i, total_batches, feedback_factor):
                        elapsed = str(datetime.timedelta(seconds=time.time() - t_begin)).split('.')[0]
                        logger.info("Elapsed: [%s] Step: %d  Batch: %d  D_Loss: %f  G_Loss: %f"
                                    % (elapsed, step, i, dis_loss, gen_loss))
                        os.makedirs(os.path.join(output, 'samples'), exist_ok=True)
                        gen_img_file = os.path.join(output, 'samples', "gen_" + str(current_depth) + "_" + str(epoch) + "_"
                                                    + str(i) + ".png")
                        self._wait_update("d"); self._wait_update("g")     # data parallel: G's Adam + EMA may still be in flight
                        with torch.no_grad():
                            sampler = self.gen_shadow if self.use_ema else self.gen
                            self.create_grid(
                                samples=sampler(fixed_input, current_depth, alpha, labels_in=fixed_labels).detach(),
                                scale_factor=int(np.power(2, self.depth - current_depth - 1)) if self.structure == 'linear' else 1,
                                img_file=gen_img_file)
                    ticker += 1
                    step += 1
                elapsed = str(datetime.timedelta(seconds=timeit.default_timer() - t_epoch)).split('.')[0]
                logger.info("Time taken for epoch: %s\n" % elapsed)
                if self.is_checkpoint_epoch(epoch, n_epochs, checkpoint_factor):
                    # data parallel: the last batch's all-reduce + Adam (+ EMA) run on the update stream; the state_dict copies
                    # below are ordered on the current stream, so it joins the update stream first (no torn / stale state)
                    self._wait_update("d"); self._wait_update("g")
                    save_dir = os.path.join(output, 'models')
                    os.makedirs(save_dir, exist_ok=True)
                    tag = str(current_depth) + "_" + str(epoch) + ".pth"
                    gen_save_file = os.path.join(save_dir, "GAN_GEN_" + tag)
                    torch.save(self.gen.state_dict(), gen_save_file)
                    logger.info("Saving the model to: %s\n" % gen_save_file)
                    torch.save(self.dis.state_dict(), os.path.join(save_dir, "GAN_DIS_" + tag))
                    torch.save(self.gen_optim.state_dict(), os.path.join(save_dir, "GAN_GEN_OPTIM_" + tag))
                    torch.save(self.dis_optim.state_dict(), os.path.join(save_dir, "GAN_DIS_OPTIM_" + tag))
                    if self.use_ema:
                        gen_shadow_save_file = os.path.join(save_dir, "GAN_GEN_SHADOW_" + tag)
                        torch.save(self.gen_shadow.state_dict(), gen_shadow_save_file)
                        logger.info("Saving the model to: %s\n" % gen_shadow_save_file)
        logger.info('Training completed.\n')



# data parallel under hipGraph replay: the update after the all-reduce as eager launches (1) or as a second captured graph (0)
DP_EAGER_UPDATE = os.environ.get("SGX_DP_EAGER_UPDATE", "1") != "0"


class _StepGraph:
    """One half-iteration (kind 'd' or 'g') at one depth and one batch shape as a replayable hipGraph.

    The first ``WARMUP`` calls run eagerly on the capture stream (they are real iterations); the next call captures the
    half-iteration and every call from then on replays it.  Everything that changes from iteration to iteration lives
    in static device tensors written before the replay: latents, images, the fade-in alpha, the style-mixing latents and
    cutoff (drawn on the host in the reference's RNG order) and Adam's bias-correction scalars.  The noise inputs are
    drawn inside the graph by torch's graph-safe CUDA generator."""

    WARMUP = 2

    def __init__(self, sg, kind, depth):
        self.sg, self.kind, self.depth = sg, kind, int(depth)
        self.calls = 0
        self.graph = None
        self.stream = torch.cuda.Stream(device=sg.device)
        self.z = self.real = self.loss = None
        dev = sg.device
        self.ab = torch.zeros(2, dtype=torch.float32, device=dev)                 # [alpha, 1 - alpha]
        self.ab_host = torch.zeros(2, dtype=torch.float32).pin_memory()
        self.cutoff = torch.zeros(1, dtype=torch.int64, device=dev)
        self.cutoff_host = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.latents2 = self.latents2_host = None
        self.adam_entries = []
        self.graph_update = None
        self.split = False                                     # data parallel: [graph: grads] -> eager all-reduce -> update (graph or eager)
        self.done = torch.cuda.Event()

    def _mixing(self):
        gen = self.sg.gen
        return gen.training and gen.style_mixing_prob is not None and gen.style_mixing_prob > 0

    def _stage(self, noise, real_batch, alpha):
        """Host RNG in the eager order, then every per-iteration input into its static tensor (stream ordered)."""
        if self.z is None:
            self.z = torch.empty_like(noise)
            self.real = torch.empty_like(real_batch)                               # keeps the (NHWC) strides
            if self._mixing():
                self.latents2 = torch.empty_like(noise)
                self.latents2_host = torch.empty(noise.shape, dtype=noise.dtype).pin_memory()
        if self._mixing():
            l2, cut = self.sg.gen.draw_mixing_host(noise.shape, self.depth)
            native._host_copy(self.latents2_host, l2); self.cutoff_host[0] = cut
            self.latents2.copy_(self.latents2_host, non_blocking=True)
            self.cutoff.copy_(self.cutoff_host, non_blocking=True)
        self.ab_host[0] = float(alpha); self.ab_host[1] = 1.0 - float(alpha)
        self.ab.copy_(self.ab_host, non_blocking=True)
        self.z.copy_(noise); self.real.copy_(real_batch)

    def _body(self, part="all"):
        """part 'all': the whole half-iteration; 'grads' / 'update': its two halves around the (eager) all-reduce."""
        sg = self.sg
        if part == "update":
            return (sg._d_update if self.kind == "d" else sg._g_update)()
        sg.gen._mixing_override = (self.latents2, self.cutoff) if self._mixing() else None
        try:
            name = {"all": "_body", "grads": "_grads"}[part]
            return getattr(sg, "_" + self.kind + name)(self.z, self.real, self.depth, self.ab)
        finally:
            sg.gen._mixing_override = None

    def _changed_params(self):
        sg = self.sg
        if self.kind == "d":
            return list(sg.dis.parameters())
        return list(sg.gen.parameters()) + (list(sg.gen_shadow.parameters()) if sg.use_ema else [])

    def run(self, noise, real_batch, alpha):
        sg = self.sg
        cur = torch.cuda.current_stream()
        # The pinned staging buffers (mixing latents, alpha, Adam scalars) are rewritten below and re-read by this graph's
        # copy nodes at replay time: the previous call of THIS graph must have finished.  (The other half-iteration's
        # graph runs in between, so the host still runs one half-iteration ahead of the GPU.)
        self.done.synchronize()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._stage(noise, real_batch, alpha)
            loss_stream = None
            if self.graph is None and self.calls < self.WARMUP:
                loss = self._body()                                                # eager, on the capture stream
                # data parallel: the eager body hands back a loss produced on the update stream (all-reduced there)
                loss_stream = sg.__dict__.pop("_loss_stream", None)
            else:
                if self.graph is None:
                    err = None
                    try:
                        self._capture()
                    except Exception as e:                                         # noqa: BLE001 -- stay correct, go eager
                        err = e
                        torch.cuda.synchronize()
                        native.lib().sgx_clear_error()                             # the failed capture leaves a sticky error
                    # data parallel: ONE outcome for all ranks -- a rank that replays [graph | all-reduce | update] next to a rank that
                    # fell back to the eager bucket schedule would issue different collectives (hang or mismatched reductions)
                    ok = err is None if sg.dp is None else sg.dp.all_ok(err is None)
                    if not ok:
                        import sys
                        why = f"{type(err).__name__}: {err}" if err is not None else "it failed on another rank"
                        print(f"stylegan.pytorch_amd: hipGraph capture of the {self.kind}-step failed ({why}); "
                              "continuing eagerly", file=sys.stderr)
                        self.graph = self.graph_update = None
                        self.split = False
                        sg.use_graphs = False
                        torch.cuda.synchronize()
                        self._undo_failed_capture()
                        loss = self._body()
                        out = DeferredLoss(loss, stream=sg.__dict__.pop("_loss_stream", None))
                        self.done.record()
                        self.calls += 1
                        cur.wait_stream(self.stream)
                        return out
                else:
                    FusedAdam.graph_advance(self.adam_entries)
                self.graph.replay()
                if self.split:                                                     # data parallel: grads | all-reduce | update
                    for p, g in self.grads:
                        if p.grad is not g:
                            p.grad = g
                    (sg._d_reduce if self.kind == "d" else sg._g_reduce)()
                    self.loss_global = sg.dp.all_reduce_scalar(self.loss)           # partial losses -> the global loss
                    if self.graph_update is not None:
                        self.graph_update.replay()
                    else:
                        self._body("update")                                       # 2-4 launches: cheaper than a second graph launch
                F.bump_weight_generation(self._changed_params())                   # eager users must re-pack these
                for p, g in self.grads:
                    if p.grad is not g:
                        p.grad = g
                loss = self.loss_global if self.split else self.loss
            out = DeferredLoss(loss, stream=loss_stream)
            self.done.record()
        self.calls += 1
        cur.wait_stream(self.stream)
        return out

    def _undo_failed_capture(self):
        """A capture RECORDS launches, it does not run them: whatever the aborted capture produced is uninitialised memory
        behind valid-looking handles, and host-side counters it advanced are one step ahead.  Before the eager retry:
        forget every weight pack (their tags match the current weights but the pack kernels never ran, and their events
        belong to the dead capture), put Adam's per-parameter step counts back, and drop the gradient tensors the capture
        allocated (so the eager backward writes fresh ones instead of accumulating into garbage)."""
        F.clear_pack_cache()
        for st, val in self.__dict__.pop("_adam_steps_before_capture", []):
            st["step"] = torch.tensor(val)
        sg = self.sg
        nets = [sg.dis] if self.kind == "d" else [sg.gen]
        for net in nets:
            for p in net.parameters():
                p.grad = None
        self.adam_entries = []

    def _capture(self):
        sg = self.sg
        opt = sg.dis_optim if self.kind == "d" else sg.gen_optim
        opt.ensure_state()
        self._adam_steps_before_capture = [(st, float(st["step"])) for st in opt.state.values() if "step" in st]
        native.reserve_capture_staging(1 << 20)                # descriptor tables are staged in pre-allocated pinned memory
        F.clear_pack_cache()                                   # the graph packs every weight it uses itself
        # No destructor may run while the launches are being recorded: a garbage-collected StyleGAN of an earlier depth / test takes
        # its captured graphs, events and streams down with it (hipGraphExecDestroy, hipEventDestroy ...), and such a call from this
        # thread in the middle of a stream capture invalidates it ("operation failed due to a previous error during capture"); with a
        # process group alive RCCL's watchdog thread then throws hipErrorCapturedEvent and the process aborts before any eager retry
        # can help (profiles/r05_capture_gc_guard.txt: 6 of 11 runs of tests/test_gpu_graphs.py aborted without this guard, 0 of 10
        # with it).  Collect first, then keep the collector off until the capture has ended.
        import gc
        gc_was = False
        split = sg.dp is not None
        net = sg.dis if self.kind == "d" else sg.gen
        try:
            if os.environ.get("SGX_CAPTURE_GC_GUARD", "1") != "0":  # (0: A/B of the guard itself)
                gc.collect()
                gc_was = gc.isenabled()
                gc.disable()                                   # (inside the try: whatever raises below, `finally` turns it back on)
            torch.cuda.synchronize()
            opt._capture_log = []
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="relaxed"):
                self.loss = self._body("grads" if split else "all")
            self.split = split
            if split and DP_EAGER_UPDATE:
                # RCCL stays outside the graph: [graph: losses + backward] -> eager bucketed all-reduce -> eager clip / Adam / EMA
                # (a handful of launches; a second graph per half-iteration costs its launch latency twice per step).
                pass
            elif split:
                # SGX_DP_EAGER_UPDATE=0: [graph: losses + backward] -> eager bucketed all-reduce -> [graph:
                # clip / Adam / EMA].  The second capture needs this iteration's gradient tensors to exist (they are
                # allocated by the first graph's capture and only hold values after a replay), so nothing is replayed
                # in between: a capture records launches, it does not run them.
                self.graph_update = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_update, stream=self.stream, capture_error_mode="relaxed"):
                    self._body("update")
        finally:
            self.adam_entries, opt._capture_log = (getattr(opt, "_capture_log", None) or []), None
            if gc_was:
                gc.enable()
        # the gradient tensors the graph writes (static addresses): re-attached after every replay so that .grad shows
        # this iteration's gradients even if an eager call in between replaced them
        self.grads = [(p, p.grad) for p in net.parameters() if p.grad is not None]
