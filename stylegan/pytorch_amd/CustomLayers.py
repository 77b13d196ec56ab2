"""Primitive layers with the reference's class names, constructor arguments and ``state_dict`` keys
(reference models/CustomLayers.py), computed by the gfx950 kernels of libsgx_hip.so.

Every ``forward`` keeps the reference contract: logical NCHW fp32 tensors in and out.  Blocks call the
``*_nhwc`` methods instead, which stay in the library's native NHWC layout (and compute dtype) between kernels.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import functional as F
from .native import ACT_LRELU, ACT_NONE, ACT_RELU, EPI_ACT, EPI_NORM

MBSTD_CPAD = 32          # the stddev channel is appended in a zero-padded group of 32 channels (MFMA K granularity)


def act_code(activation_layer):
    """ACT_LRELU / ACT_RELU for the two nonlinearities the reference offers (models/GAN.py:67-68,150-151,346-347:
    ``nn.LeakyReLU(negative_slope=0.2)`` | ``torch.relu``)."""
    if isinstance(activation_layer, nn.LeakyReLU) and abs(activation_layer.negative_slope - 0.2) < 1e-12:
        return ACT_LRELU
    if activation_layer is torch.relu or isinstance(activation_layer, nn.ReLU):
        return ACT_RELU
    raise NotImplementedError(f"activation {activation_layer!r}: the reference offers 'lrelu' (0.2) and 'relu'")


def apply_act(x, act):
    """Standalone activation of an NHWC tensor (the un-fused path: ReLU networks, non-default layer stacks)."""
    return F.call(F.BiasActFn, x, None, 1.0, act) if act else x


class PixelNormLayer(nn.Module):
    """x * rsqrt(mean(x^2, dim=1) + eps) -- reference models/CustomLayers.py:17-23.  The training path applies it
    to the latent z [B, C] (models/GAN.py:75-76); 4-D inputs are normalised over the channel dim as well."""

    def __init__(self, epsilon=1e-8):
        super().__init__()
        self.epsilon = epsilon

    def forward(self, x):
        assert abs(self.epsilon - 1e-8) < 1e-20, "kernel is built for the reference's epsilon"
        if x.dim() == 2:
            return F.call(F.PixelNormFn, x)
        shape = x.shape                                         # [B, C, ...] -> rows of C
        y = F.call(F.PixelNormFn, x.movedim(1, -1).reshape(-1, shape[1]))
        return y.reshape(shape[0], *shape[2:], shape[1]).movedim(-1, 1)


class Upscale2d(nn.Module):
    """Nearest-neighbour replicate -- reference models/CustomLayers.py:26-45."""

    def __init__(self, factor=2, gain=1):
        super().__init__()
        assert isinstance(factor, int) and factor >= 1
        self.gain = gain
        self.factor = factor

    @staticmethod
    def upscale2d(x, factor=2, gain=1):
        assert x.dim() == 4
        if factor == 1:
            return x * gain if gain != 1 else x
        assert factor == 2, "the kernel implements the factor-2 case the networks use"
        return F.nchw_view(F.call(F.Up2Fn, F.nhwc(x), float(gain)))

    def forward(self, x):
        return self.upscale2d(x, factor=self.factor, gain=self.gain)


class BlurLayer(nn.Module):
    """Depthwise separable blur, zero padded -- reference models/CustomLayers.py:251-276.  The kernel buffer keeps
    the reference's name/shape; the HIP kernel implements the default normalised [1,2,1] filter (stride 1)."""

    def __init__(self, kernel=None, normalize=True, flip=False, stride=1):
        super().__init__()
        if kernel is None:
            kernel = [1, 2, 1]
        k = torch.tensor(kernel, dtype=torch.float32)
        k = k[:, None] * k[None, :]
        k = k[None, None]
        if normalize:
            k = k / k.sum()
        if flip:
            k = k.flip(2, 3)
        self.register_buffer('kernel', k)
        self.stride = stride
        self._is_121 = (list(kernel) == [1, 2, 1]) and normalize and stride == 1
        self._is_box2 = (len(kernel) == 2 and stride == 2 and abs(float(k.sum()) - 1.0) < 1e-6
                         and float((k - k.mean()).abs().max()) < 1e-7)
        self._taps = None                        # (buffer version, K, taps tuple): host copy of ``kernel`` for the generic path

    def _host_taps(self):
        ver = self.kernel._version
        if self._taps is None or self._taps[0] != ver:
            k = self.kernel.detach().float().cpu()
            self._taps = (ver, int(k.shape[2]), tuple(float(v) for v in k.reshape(-1)))
        return self._taps[1], self._taps[2]

    def forward_nhwc(self, x):
        if self._is_121:
            return F.call(F.BlurFn, x)
        if self._is_box2:
            return F.call(F.Pool2Fn, x, 0.25)
        if self.stride != 1:
            raise NotImplementedError("BlurLayer: stride 2 is built for the 2x2 box of Downscale2d only")
        K, taps = self._host_taps()                                    # any other filter: generic K x K kernel (:266-275)
        if K > 7:
            raise NotImplementedError("BlurLayer: filters longer than 7 taps are not built")
        pad = int((K - 1) / 2)
        return F.call(F.BlurGenFn, x, taps, K, pad, x.shape[1] + 2 * pad - K + 1, x.shape[2] + 2 * pad - K + 1)

    def forward(self, x):
        return F.nchw_view(self.forward_nhwc(F.nhwc(x)))


class Downscale2d(nn.Module):
    """2x2 mean -- reference models/CustomLayers.py:48-76 (both of its branches are this arithmetic)."""

    def __init__(self, factor=2, gain=1):
        super().__init__()
        assert isinstance(factor, int) and factor >= 1
        self.factor = factor
        self.gain = gain
        if factor == 2:
            f = [np.sqrt(gain) / factor] * factor
            self.blur = BlurLayer(kernel=f, normalize=False, stride=factor)
        else:
            self.blur = None

    def forward_nhwc(self, x):
        assert self.factor == 2, "the kernel implements the factor-2 case the networks use"
        return F.call(F.Pool2Fn, x, 0.25 * float(self.gain))

    def forward(self, x):
        assert x.dim() == 4
        return F.nchw_view(self.forward_nhwc(F.nhwc(x)))


class EqualizedLinear(nn.Module):
    """Linear layer with equalized learning rate -- reference models/CustomLayers.py:79-103."""

    # True (set by the generator's modules): nothing differentiates twice through this layer, so it runs as the fused
    # three-launch LinearFn; False: the differentiable composite (the discriminator head sits under the R1 double backward)
    first_order_only = False

    def __init__(self, input_size, output_size, gain=2 ** 0.5, use_wscale=False, lrmul=1, bias=True):
        super().__init__()
        he_std = gain * input_size ** (-0.5)
        if use_wscale:
            init_std = 1.0 / lrmul
            self.w_mul = he_std * lrmul
        else:
            init_std = he_std / lrmul
            self.w_mul = lrmul
        self.weight = nn.Parameter(torch.randn(output_size, input_size) * init_std)
        if bias:
            self.bias = nn.Parameter(torch.zeros(output_size))
            self.b_mul = lrmul
        else:
            self.bias = None
            self.b_mul = 1

    def forward(self, x, act=ACT_NONE, weight=None):
        """``act`` fuses the LeakyReLU that follows the layer in the mapping network / D head; ``weight``
        substitutes a re-ordered view of the weight (D dense0 consumes NHWC-flattened features)."""
        w = self.weight if weight is None else weight
        if self.first_order_only and x.dim() == 2:
            return F.linear_fused(x.float(), w, self.bias, self.w_mul, self.b_mul, act)
        return F.linear(x.float(), w, self.bias, self.w_mul, self.b_mul, act)


class EqualizedConv2d(nn.Module):
    """Conv layer with equalized learning rate -- reference models/CustomLayers.py:106-180.

    All of the reference's paths are served by three MFMA kernels: plain 3x3; 4x4 stride-2 (the fused
    conv+downscale, which is also exactly conv3x3 -> avg_pool2 of the non-fused branch); 4x4 stride-2 transposed
    (the fused upscale+conv; the non-fused nearest-up -> conv3x3 branch is the same operator with the 3x3 kernel
    flipped).  The fused/non-fused *semantics* still switch on the input size exactly where the
    reference switches (:143), because the two up paths differ by that flip.
    """

    def __init__(self, input_channels, output_channels, kernel_size, stride=1, gain=2 ** 0.5, use_wscale=False,
                 lrmul=1, bias=True, intermediate=None, upscale=False, downscale=False):
        super().__init__()
        self.upscale = Upscale2d() if upscale else None
        self.downscale = Downscale2d() if downscale else None
        he_std = gain * (input_channels * kernel_size ** 2) ** (-0.5)
        self.kernel_size = kernel_size
        if use_wscale:
            init_std = 1.0 / lrmul
            self.w_mul = he_std * lrmul
        else:
            init_std = he_std / lrmul
            self.w_mul = lrmul
        self.weight = nn.Parameter(torch.randn(output_channels, input_channels, kernel_size, kernel_size) * init_std)
        if bias:
            self.bias = nn.Parameter(torch.zeros(output_channels))
            self.b_mul = lrmul
        else:
            self.bias = None
            self.b_mul = 1
        self.intermediate = intermediate
        assert stride == 1

    def scaled_bias(self):
        """The bias as the kernels consume it (b_mul is 1 for every convolution of the networks)."""
        if self.bias is None:
            return None
        return self.bias * self.b_mul if self.b_mul != 1 else self.bias

    def forward_nhwc(self, x, act=ACT_NONE, skip_bias=False, out_dtype=None, defer_act=False, x_masked=False, out_scale=1.0,
                     epi_stats=None, x_pre=None, sign_bits=False, x_pre_bits=None, fade=None):
        """x: NHWC.  ``skip_bias``: the caller folds the bias into the next kernel (generator epilogue).
        ``defer_act`` / ``x_masked``: the LeakyReLU backward of this layer is applied by its consumer / this layer's input is
        such an output and its data gradient leaves the kernel already masked (functional.ConvFn; discriminator chain only).
        ``out_scale`` (from_rgb only): the layer's output times a python-float factor, folded into its weight scale and bias
        (the fade-in coefficient of the residual branch: neither the forward nor the backward needs a scaling pass then).
        The parameter is consumed in place: w_mul, the 3x3 -> 4x4 kernel synthesis and the MFMA operand packing run in
        sgx_pack_weight (cached per parameter version), their adjoints in the weight-gradient finishing kernel."""
        bias = None if skip_bias else self.scaled_bias()
        cin, cout = self.weight.shape[1], self.weight.shape[0]
        if self.kernel_size == 1:
            assert self.upscale is None and self.downscale is None and self.intermediate is None
            if cin == 3 and x.shape[3] == 3:
                if out_scale != 1.0:
                    y = F.call(F.RgbInFn, x.float(), self.weight, None if bias is None else bias * float(out_scale),
                               self.w_mul * float(out_scale), out_dtype or torch.float32)
                else:
                    y = F.call(F.RgbInFn, x.float(), self.weight, bias, self.w_mul, out_dtype or torch.float32)
            elif cin % 3 == 0 and cin > 3 and x.shape[3] == cin and cout != 3:
                # conditional discriminator (reference models/GAN.py:326-330,415-421): from_rgb over [image, label
                # embedding] = the sum of the 3-channel kernels over the channel groups (the weight slices are tiny)
                y = None
                for k in range(0, cin, 3):
                    part = F.call(F.RgbInFn, x[..., k:k + 3].float().contiguous(), self.weight[:, k:k + 3].contiguous(),
                                  bias if k == 0 else None, self.w_mul, out_dtype or torch.float32)
                    y = part if y is None else F.call(F.AxpbyFn, y, part, 1.0, 1.0)
            elif cout == 3:
                y = F.call(F.RgbOutFn, x, self.weight, bias, self.w_mul)
            else:
                raise NotImplementedError("1x1 EqualizedConv2d is built for the to_rgb / from_rgb layers (3 channels on one side)")
            return F.call(F.BiasActFn, y, None, 1.0, act) if act else y
        assert self.kernel_size == 3
        if epi_stats is not None:
            # generator layers (Blocks.GSynthesisBlock / InputBlock): ``epi_stats`` = (epilogue bias, noise, noise weight) of
            # the LayerEpilogue that follows; the kernel that writes this layer's output also emits the partial instance-norm
            # statistics -> returns (y, partials)
            assert skip_bias and act == ACT_NONE and self.downscale is None
            ebias, noise, nw = epi_stats
            if self.upscale is not None:
                fused = min(x.shape[1], x.shape[2]) * 2 >= 128            # reference :143
                y = F.conv(x, self.weight, None, "U" if fused else "UF", self.w_mul)
                assert self.intermediate is not None and self.intermediate._is_121
                return F.call(F.BlurStatsFn, y, ebias, noise, nw)         # blur + statistics in one pass
            assert self.intermediate is None
            if F.conv_stats_nparts(x, self.weight.shape[0]) > 0:          # 3x3: statistics out of the convolution's store epilogue
                return F.conv(x, self.weight, None, "S", self.w_mul, ipad=x.shape[3], stats=(ebias, noise, nw))
            return F.conv(x, self.weight, None, "S", self.w_mul, ipad=x.shape[3]), None   # no fused kernel: separate pass
        if self.upscale is not None:
            fused = min(x.shape[1], x.shape[2]) * 2 >= 128                # reference :143
            mode = "U" if fused else "UF"
            if (self.intermediate is not None and self.intermediate._is_121
                    and F.conv_blur_ok(x, self.weight.shape[0], mode, False)):
                # transposed convolution and the blur after it in one kernel (the blur in the store epilogue)
                y = F.call(F.ConvBlurFn, x, self.weight, mode, float(self.w_mul), int(self.weight.shape[1]), False, None)
            else:
                y = F.conv(x, self.weight, None, mode, self.w_mul)
                if self.intermediate is not None:
                    y = self.intermediate.forward_nhwc(y)
            if bias is not None or act:
                y = F.call(F.BiasActFn, y, bias, 1.0, act)                  # bias after the blur (:178-179)
            return y
        if self.downscale is not None:
            assert self.intermediate is None                              # reference :167
            if fade is not None:
                # (residual branch, alpha, beta): the fade-in lerp of the discriminator's newest block in this layer's store
                # (functional.ConvDownFadeFn; the caller checked functional.conv_down_fade_ok)
                assert act == ACT_LRELU and not x_masked
                if x_pre is not None and getattr(x, "_sgx_pre_of", None) is not x_pre:
                    raise F.N.SgxError("conv+fade: x_pre given, but x is not the ActBlurPassFn output of that tensor")
                resid, alpha, beta = fade
                if not isinstance(alpha, torch.Tensor):
                    alpha, beta = float(alpha), float(beta)
                if isinstance(resid, F.RgbResidual):
                    # the residual branch from_rgb(pooled image) evaluated inside the store (functional.ConvDownFadeRgbFn)
                    lay = resid.layer
                    return F.call(F.ConvDownFadeRgbFn, x, self.weight, bias, resid.pimg, lay.weight, lay.bias, float(self.w_mul), int(self.weight.shape[1]),
                                  alpha, beta, lay.w_mul * float(resid.out_scale), float(lay.b_mul), float(resid.out_scale), x_pre, x_pre_bits)
                return F.call(F.ConvDownFadeFn, x, self.weight, bias, resid, float(self.w_mul), int(self.weight.shape[1]), alpha, beta, x_pre, x_pre_bits)
            return F.conv(x, self.weight, bias, "D", self.w_mul, act, defer_act=defer_act and act == ACT_LRELU,
                          x_pre=x_pre, x_pre_bits=x_pre_bits)             # bias after the 2x2 mean == bias in the fused store
        if self.intermediate is None:
            if sign_bits:
                # -> (y, sign bits of y or None): the discriminator block's conv0, whose output is the LeakyReLU-backward mask
                if F.conv_signbits_ok(x, self.weight.shape[0]):
                    return F.conv(x, self.weight, bias, "S", self.w_mul, act, ipad=x.shape[3], x_masked=x_masked, bits_out=True)
                return F.conv(x, self.weight, bias, "S", self.w_mul, act, ipad=x.shape[3], x_masked=x_masked), None
            return F.conv(x, self.weight, bias, "S", self.w_mul, act, ipad=x.shape[3], x_masked=x_masked)
        y = self.intermediate.forward_nhwc(F.conv(x, self.weight, None, "S", self.w_mul))
        return F.call(F.BiasActFn, y, bias, 1.0, act) if (bias is not None or act) else y

    def forward(self, x):
        if self.kernel_size == 1 and self.weight.shape[0] == 3:
            return F.nchw_view(self.forward_nhwc(F.nhwc(x)))
        xin = F.nhwc(x, torch.float32)
        return F.nchw_view(self.forward_nhwc(xin))


class NoiseLayer(nn.Module):
    """Per-pixel noise with a per-channel weight -- reference models/CustomLayers.py:183-200.  Inside the networks
    the addition is fused into the layer-epilogue kernel; this standalone forward is the reference formula."""

    def __init__(self, channels):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(channels))
        self.noise = None

    def sample(self, x_nhwc_shape, device):
        """Noise for one layer, [B,1,H,W] fp32: the preset ``.noise`` if set (reference :194-198), else a fresh draw
        from the device generator in the reference's order (:193)."""
        if self.noise is not None:
            return self.noise
        b, h, w, _ = x_nhwc_shape
        if F.NOISE_ARENA is not None:
            return F.NOISE_ARENA.take(b, h, w)                      # slice of the forward's single randn
        return torch.randn(b, 1, h, w, device=device, dtype=torch.float32)

    def forward(self, x, noise=None):
        if noise is None:
            noise = self.sample((x.size(0), x.size(2), x.size(3), x.size(1)), x.device)
        return x + self.weight.view(1, -1, 1, 1).to(x.dtype) * noise.to(x.dtype)   # standalone use only (not the hot path)


class StyleMod(nn.Module):
    """x * (style0 + 1) + style1 with style = EqualizedLinear(w) -- reference models/CustomLayers.py:203-216."""

    def __init__(self, latent_size, channels, use_wscale):
        super().__init__()
        self.lin = EqualizedLinear(latent_size, channels * 2, gain=1.0, use_wscale=use_wscale)
        self.lin.first_order_only = True

    def style(self, latent):
        return self.lin(latent)                                           # [B, 2C] fp32

    def forward(self, x, latent):
        s = self.style(latent).view(-1, 2, x.size(1), *([1] * (x.dim() - 2))).to(x.dtype)
        return x * (s[:, 0] + 1.) + s[:, 1]


class LayerEpilogue(nn.Module):
    """Things to do at the end of each generator layer -- reference models/CustomLayers.py:219-248.
    noise -> LeakyReLU -> InstanceNorm -> StyleMod run as ONE fused HIP op (two passes over the tensor)."""

    def __init__(self, channels, dlatent_size, use_wscale, use_noise, use_pixel_norm, use_instance_norm, use_styles,
                 activation_layer):
        super().__init__()
        from collections import OrderedDict
        layers = []
        if use_noise:
            layers.append(('noise', NoiseLayer(channels)))
        layers.append(('activation', activation_layer))
        if use_pixel_norm:
            layers.append(('pixel_norm', PixelNormLayer()))
        if use_instance_norm:
            layers.append(('instance_norm', nn.InstanceNorm2d(channels)))
        self.top_epi = nn.Sequential(OrderedDict(layers))
        self.style_mod = StyleMod(dlatent_size, channels, use_wscale=use_wscale) if use_styles else None
        self._act = act_code(activation_layer)
        self._use = (bool(use_noise), bool(use_pixel_norm), bool(use_instance_norm), bool(use_styles))
        # the reference's default stack (models/GAN.py:108): one fused op; every other combination runs the same kernels
        # stage by stage (forward_nhwc below)
        self._fusable = use_noise and use_instance_norm and use_styles and not use_pixel_norm and self._act == ACT_LRELU

    def _style(self, dlatents_in_slice):
        if isinstance(dlatents_in_slice, F.PreStyle):                   # computed with all the other layers' in one launch
            return dlatents_in_slice.style
        return self.style_mod.style(dlatents_in_slice)

    def noise_inputs(self, x_shape, device):
        """(noise [B,1,H,W], noise weight) of this layer for an NHWC activation of ``x_shape`` -- drawn here when the producer
        of the activation folds the statistics pass into its store and needs them before the epilogue runs."""
        noise_layer = self.top_epi.noise
        return noise_layer.sample(x_shape, device), noise_layer.weight

    def forward_nhwc(self, x, dlatents_in_slice, conv_bias=None, noise_in=None, pre_stats=None):
        """``noise_in`` / ``pre_stats`` (default stack only): the (noise, weight) pair already drawn by ``noise_inputs`` and the
        partial instance-norm statistics the producer of ``x`` computed with them (functional.BlurStatsFn / ConvFn)."""
        use_noise, use_pixel_norm, use_instance_norm, use_styles = self._use
        noise = nw = None
        if noise_in is not None:
            noise, nw = noise_in
        elif use_noise:
            noise, nw = self.noise_inputs(x.shape, x.device)
        style = self._style(dlatents_in_slice) if use_styles else None
        if self._fusable:
            return F.call(F.GEpilogueFn, x, conv_bias, noise, nw, style, EPI_ACT | EPI_NORM, pre_stats)
        assert pre_stats is None
        # Non-default stacks (reference :224-246: noise -> activation -> [pixel norm] -> [instance norm] -> [style]), on the
        # same kernels: the fused op with its stages switched by flags, split in two around a ReLU / a pixel norm.
        lrelu = self._act == ACT_LRELU
        norm = EPI_NORM if use_instance_norm else 0
        if lrelu and not use_pixel_norm:
            return F.call(F.GEpilogueFn, x, conv_bias, noise, nw, style, EPI_ACT | norm)
        x = F.call(F.GEpilogueFn, x, conv_bias, noise, nw, None, EPI_ACT if lrelu else 0)      # x + bias + noise [-> lrelu]
        if not lrelu:
            x = apply_act(x, self._act)
        if use_pixel_norm:
            shape, dt = x.shape, x.dtype
            x = F.call(F.PixelNormFn, x.reshape(-1, shape[3])).reshape(shape).to(dt)          # over channels, per pixel (:22-23)
        if norm or style is not None:
            x = F.call(F.GEpilogueFn, x, None, None, None, style, norm)
        return x

    def forward(self, x, dlatents_in_slice=None):
        return F.nchw_view(self.forward_nhwc(F.nhwc(x), dlatents_in_slice))


class View(nn.Module):
    """reference models/CustomLayers.py:279-285."""

    def __init__(self, *shape):
        super().__init__()
        self.shape = shape

    def forward(self, x):
        return x.reshape(x.size(0), *self.shape)


class StddevLayer(nn.Module):
    """Minibatch standard deviation -- reference models/CustomLayers.py:288-305 (group_size 4, one new feature)."""

    def __init__(self, group_size=4, num_new_features=1):
        super().__init__()
        self.group_size = group_size
        self.num_new_features = num_new_features

    def forward_nhwc(self, x):
        """[B,H,W,C] -> [B,H,W,C+MBSTD_CPAD]; channel C is the statistic, the rest is zero padding."""
        if self.group_size != 4 or self.num_new_features != 1:
            raise NotImplementedError("StddevLayer: the kernel implements group_size=4, num_new_features=1")
        return F.call(F.MbstdFn, x, x.shape[3] + MBSTD_CPAD)

    def forward(self, x):
        y = self.forward_nhwc(F.nhwc(x))
        return F.nchw_view(y)[:, :x.size(1) + 1]


class Truncation(nn.Module):
    """W-space truncation trick + moving average of W -- reference models/CustomLayers.py:308-323.
    [B, layers, 512] fp32 bookkeeping, two tiny elementwise ops."""

    def __init__(self, avg_latent, max_layer=8, threshold=0.7, beta=0.995):
        super().__init__()
        self.max_layer = max_layer
        self.threshold = threshold
        self.beta = beta
        self.register_buffer('avg_latent', avg_latent)

    def update(self, last_avg):
        self.avg_latent.copy_(self.beta * self.avg_latent + (1. - self.beta) * last_avg)

    def forward(self, x):
        assert x.dim() == 3
        interp = torch.lerp(self.avg_latent.expand_as(x), x, self.threshold)
        do_trunc = (torch.arange(x.size(1), device=x.device) < self.max_layer).view(1, -1, 1)
        return torch.where(do_trunc, interp, x)
