"""Per-resolution blocks with the reference's names and ``state_dict`` keys (reference models/Blocks.py).

``forward`` keeps the reference contract (logical NCHW fp32); the networks call ``forward_nhwc`` and stay in the
kernels' NHWC layout / compute dtype from the first layer to the last.
"""
import os

import torch
import torch.nn as nn

from . import functional as F
from .CustomLayers import (BlurLayer, EqualizedConv2d, EqualizedLinear, LayerEpilogue, StddevLayer, View, act_code, apply_act)
from .native import ACT_LRELU, ACT_NONE


# Instance-norm statistics of a generator layer epilogue out of the kernel that PRODUCES its input (A/B switch):
# bit 0: the blur after conv0_up (epi1), bit 1: the 3x3 convolution conv1 (epi2).  Measured alone on the GPU
# (tools/convstats_probe.py, profiles/r03_convstats_probe.txt): the convolution's statistics epilogue costs it 60-80 us at
# batch 32 and saves a 1-tensor read pass -- +136 us per layer at 1024^2, +27 at 512^2 and 256^2, a loss below 2^27 elements
# (the finalize over one partial per tile slot is latency-bound: -45 us at batch 4, 1024^2) -> on from FUSE_EPI_STATS_MIN elements.
# The blur variant is slower than blur + statistics pass at every size (175 registers: two waves per SIMD) -> off by default.
FUSE_EPI_STATS = int(os.environ.get("SGX_FUSE_EPI_STATS", "2"))
FUSE_EPI_STATS_MIN = int(os.environ.get("SGX_FUSE_EPI_STATS_MIN", str(1 << 27)))


# Discriminator block backward: the blur (and the LeakyReLU mask) between conv0 and conv1_down folded into conv1_down's data
# gradient (functional.ConvFn x_pre / ConvBlurFn).  0: the separate blur-and-mask pass (A/B; the kernel-level switch is
# SGX_CONV_UP_BLUR, which also covers the generator's conv0_up -> blur).
FUSE_BLUR_BWD = os.environ.get("SGX_FUSE_BLUR_BWD", "1") != "0"


def _lat(d, k):
    """Layer k of a block's pair of dlatents: a [B, 2, D] tensor (reference signature) or a pair of [B, D] tensors."""
    return d[k] if isinstance(d, (tuple, list)) else d[:, k]


class InputBlock(nn.Module):
    """The 4x4 block: learned constant (+bias) -> epilogue -> conv3x3 -> epilogue -- reference models/Blocks.py:17-60."""

    def __init__(self, nf, dlatent_size, const_input_layer, gain, use_wscale, use_noise, use_pixel_norm,
                 use_instance_norm, use_styles, activation_layer):
        super().__init__()
        self.const_input_layer = const_input_layer
        self.nf = nf
        if self.const_input_layer:
            self.const = nn.Parameter(torch.ones(1, nf, 4, 4))
            self.bias = nn.Parameter(torch.ones(nf))
        else:
            self.dense = EqualizedLinear(dlatent_size, nf * 16, gain=gain / 4, use_wscale=use_wscale)
            self.dense.first_order_only = True
        self.epi1 = LayerEpilogue(nf, dlatent_size, use_wscale, use_noise, use_pixel_norm, use_instance_norm,
                                  use_styles, activation_layer)
        self.conv = EqualizedConv2d(nf, nf, 3, gain=gain, use_wscale=use_wscale)
        self.epi2 = LayerEpilogue(nf, dlatent_size, use_wscale, use_noise, use_pixel_norm, use_instance_norm,
                                  use_styles, activation_layer)

    def forward_nhwc(self, dlatents_in_range, dtype=torch.float32):
        first = _lat(dlatents_in_range, 0)
        b = (first.style if isinstance(first, F.PreStyle) else first).size(0)
        if self.const_input_layer:
            # const [1,C,4,4] -> NHWC [B,4,4,C]; its bias is folded into the epilogue kernel (Blocks.py:51-52)
            x = self.const.permute(0, 2, 3, 1).to(dtype).expand(b, -1, -1, -1).contiguous()
            bias = self.bias
        else:
            x = self.dense(_lat(dlatents_in_range, 0)).view(b, self.nf, 4, 4).permute(0, 2, 3, 1).to(dtype).contiguous()
            bias = None
        x = self.epi1.forward_nhwc(x, _lat(dlatents_in_range, 0), conv_bias=bias)
        x = self.conv.forward_nhwc(x, skip_bias=True)
        return self.epi2.forward_nhwc(x, _lat(dlatents_in_range, 1), conv_bias=self.conv.scaled_bias())

    def forward(self, dlatents_in_range):
        return F.nchw_view(self.forward_nhwc(dlatents_in_range))


class GSynthesisBlock(nn.Module):
    """upscale-conv (+blur) -> epilogue -> conv3x3 -> epilogue -- reference models/Blocks.py:63-88."""

    def __init__(self, in_channels, out_channels, blur_filter, dlatent_size, gain, use_wscale, use_noise,
                 use_pixel_norm, use_instance_norm, use_styles, activation_layer):
        super().__init__()
        blur = BlurLayer(blur_filter) if blur_filter else None
        self.conv0_up = EqualizedConv2d(in_channels, out_channels, kernel_size=3, gain=gain, use_wscale=use_wscale,
                                        intermediate=blur, upscale=True)
        self.epi1 = LayerEpilogue(out_channels, dlatent_size, use_wscale, use_noise, use_pixel_norm,
                                  use_instance_norm, use_styles, activation_layer)
        self.conv1 = EqualizedConv2d(out_channels, out_channels, kernel_size=3, gain=gain, use_wscale=use_wscale)
        self.epi2 = LayerEpilogue(out_channels, dlatent_size, use_wscale, use_noise, use_pixel_norm,
                                  use_instance_norm, use_styles, activation_layer)

    def forward_nhwc(self, x, dlatents_in_range, defer_epi2=False):
        """``defer_epi2`` (the LAST block of a forward, default epilogue stack): epi2 is not applied -- returns (conv1's output,
        (epilogue bias, noise, noise weight, style, producer statistics or None)) for ``functional.EpiRgbOutFn``, which applies the
        epilogue inside the to_rgb convolution that is its only consumer."""
        up = self.conv0_up
        if (FUSE_EPI_STATS & 1) and self.epi1._fusable and up.intermediate is not None and up.intermediate._is_121:
            # the blur after the upscale-conv also emits the epilogue's instance-norm statistics (one pass less over the tensor)
            b, h, w, _ = x.shape
            nin = self.epi1.noise_inputs((b, 2 * h, 2 * w, up.weight.shape[0]), x.device)
            x, part = up.forward_nhwc(x, skip_bias=True, epi_stats=(up.scaled_bias(),) + nin)
            x = self.epi1.forward_nhwc(x, _lat(dlatents_in_range, 0), conv_bias=up.scaled_bias(), noise_in=nin, pre_stats=part)
        else:
            x = up.forward_nhwc(x, skip_bias=True)                        # transposed conv + blur; bias folded below
            x = self.epi1.forward_nhwc(x, _lat(dlatents_in_range, 0), conv_bias=up.scaled_bias())
        defer_epi2 = defer_epi2 and self.epi2._fusable
        if (FUSE_EPI_STATS & 2) and self.epi2._fusable and x.numel() >= FUSE_EPI_STATS_MIN:
            nin = self.epi2.noise_inputs(x.shape, x.device)               # conv1 keeps the shape
            x, part = self.conv1.forward_nhwc(x, skip_bias=True, epi_stats=(self.conv1.scaled_bias(),) + nin)
            if defer_epi2:
                return x, (self.conv1.scaled_bias(), nin[0], nin[1], self.epi2._style(_lat(dlatents_in_range, 1)), part)
            return self.epi2.forward_nhwc(x, _lat(dlatents_in_range, 1), conv_bias=self.conv1.scaled_bias(), noise_in=nin, pre_stats=part)
        if defer_epi2:
            nin = self.epi2.noise_inputs(x.shape, x.device)
            x = self.conv1.forward_nhwc(x, skip_bias=True)
            return x, (self.conv1.scaled_bias(), nin[0], nin[1], self.epi2._style(_lat(dlatents_in_range, 1)), None)
        x = self.conv1.forward_nhwc(x, skip_bias=True)
        return self.epi2.forward_nhwc(x, _lat(dlatents_in_range, 1), conv_bias=self.conv1.scaled_bias())

    def forward(self, x, dlatents_in_range):
        return F.nchw_view(self.forward_nhwc(F.nhwc(x), dlatents_in_range))


class DiscriminatorTop(nn.Module):
    """stddev -> conv3x3 -> lrelu -> flatten -> dense -> lrelu -> dense -- reference models/Blocks.py:91-134
    (an nn.Sequential there; same child names, hence the same state_dict keys)."""

    def __init__(self, mbstd_group_size, mbstd_num_features, in_channels, intermediate_channels, gain, use_wscale,
                 activation_layer, resolution=4, in_channels2=None, output_features=1, last_gain=1):
        super().__init__()
        if mbstd_group_size > 1:
            self.stddev_layer = StddevLayer(mbstd_group_size, mbstd_num_features)
        else:
            self.stddev_layer = None
        if in_channels2 is None:
            in_channels2 = in_channels
        nfeat = mbstd_num_features if mbstd_group_size > 1 else 0
        self.conv = EqualizedConv2d(in_channels + nfeat, in_channels2, kernel_size=3, gain=gain, use_wscale=use_wscale)
        self.act0 = activation_layer
        self.view = View(-1)
        self.dense0 = EqualizedLinear(in_channels2 * resolution * resolution, intermediate_channels, gain=gain,
                                      use_wscale=use_wscale)
        self.act1 = activation_layer
        self.dense1 = EqualizedLinear(intermediate_channels, output_features, gain=last_gain, use_wscale=use_wscale)
        self.resolution = resolution
        self._act = act_code(activation_layer)        # LeakyReLU(0.2) rides in the kernels' stores; ReLU runs as its own pass

    def forward_nhwc(self, x):
        fused = ACT_LRELU if self._act == ACT_LRELU else ACT_NONE
        post = ACT_NONE if fused else self._act
        if self.stddev_layer is not None:
            x = self.stddev_layer.forward_nhwc(x)
        x = apply_act(self.conv.forward_nhwc(x, act=fused), post)        # [B,4,4,C]
        b = x.shape[0]
        flat = x.permute(0, 3, 1, 2).reshape(b, -1).float()               # View(-1) flattens NCHW (index c*16+h*4+w)
        y = apply_act(self.dense0(flat, act=fused), post)
        return self.dense1(y)

    def forward(self, x):
        return self.forward_nhwc(F.nhwc(x))


class DiscriminatorBlock(nn.Module):
    """conv3x3 -> lrelu -> blur -> conv-downscale -> lrelu -- reference models/Blocks.py:137-146."""

    def __init__(self, in_channels, out_channels, gain, use_wscale, activation_layer, blur_kernel):
        super().__init__()
        self.conv0 = EqualizedConv2d(in_channels, in_channels, kernel_size=3, gain=gain, use_wscale=use_wscale)
        self.act0 = activation_layer
        self.blur = BlurLayer(kernel=blur_kernel)
        self.conv1_down = EqualizedConv2d(in_channels, out_channels, kernel_size=3, gain=gain, use_wscale=use_wscale,
                                          downscale=True)
        self.act1 = activation_layer
        self._act = act_code(activation_layer)

    def fused_from_rgb_ok(self, img_shape, from_rgb, dtype):
        """True if ``from_rgb -> conv0 -> LeakyReLU -> blur`` of this block has the composed 3-channel kernel for an NHWC image of
        ``img_shape`` (functional.RgbConvBlurFn: bf16, 16 or 32 channels, the default blur and activation; A/B: SGX_RGBCONV=0)."""
        c0 = self.conv0
        return (FUSE_BLUR_BWD and self.blur._is_121 and self._act == ACT_LRELU and len(img_shape) == 4 and img_shape[3] == 3
                and from_rgb.kernel_size == 1 and tuple(from_rgb.weight.shape[1:]) == (3, 1, 1) and c0.b_mul == 1 and from_rgb.b_mul == 1
                and from_rgb.weight.shape[0] == c0.weight.shape[1] == c0.weight.shape[0]
                and F.rgbconv_ok(img_shape[0], img_shape[1], img_shape[2], c0.weight.shape[0], dtype))

    def forward_from_image(self, img, from_rgb, defer_out=False, fade=None):
        """The block applied to ``from_rgb(img)`` with from_rgb and conv0 composed into one convolution of the fp32 NHWC image
        (reference models/GAN.py:425 + models/Blocks.py:139-146): neither from_rgb's output nor conv0's pre-activation exists in
        memory; the backward gets the activation mask from the sign bits the kernel writes."""
        c0 = self.conv0
        xb, zbits = F.rgbconv_blur(img, c0.weight, c0.scaled_bias(), from_rgb.weight, from_rgb.scaled_bias(), c0.w_mul, from_rgb.w_mul)
        if fade is not None and F.conv_down_fade_ok(xb, self.conv1_down.weight.shape[0]):
            return self.conv1_down.forward_nhwc(xb, act=ACT_LRELU, x_pre=None, x_pre_bits=zbits, fade=fade), True
        y = self.conv1_down.forward_nhwc(xb, act=ACT_LRELU, defer_act=defer_out and self._act == ACT_LRELU, x_pre=None, x_pre_bits=zbits)
        return (y, False) if fade is not None else y

    def forward_nhwc(self, x, x_masked=False, defer_out=False, fade=None):
        """``fade`` = (residual branch, alpha, beta) (newest block, LeakyReLU networks): the fade-in lerp rides in conv1_down's store
        where that kernel exists -> returns (output, True) with the lerp applied, else (block output, False).
        ``x_masked``: x is the previous block's LeakyReLU output whose activation backward was deferred to this block
        (conv0's data gradient leaves its kernel already masked); ``defer_out``: this block's final activation backward is
        applied by the consumer of its output (the next block, or the fade-in lerp).  Set by Discriminator.forward for the
        LeakyReLU networks: one elementwise pass less per block and backward."""
        x_masked = x_masked and self._act == ACT_LRELU
        defer_out = defer_out and self._act == ACT_LRELU
        if self.blur._is_121 and self._act == ACT_LRELU and FUSE_BLUR_BWD:
            # LeakyReLU folded into the blur pass; backward: the blur and the activation's mask belong to conv1_down's data
            # gradient (ConvFn x_pre): one kernel where that wins, else the blur-and-mask pass -- which reads the mask as the
            # SIGN BITS that conv0's store wrote next to z (1 bit per element instead of 16) where conv0 has that variant
            z, zbits = self.conv0.forward_nhwc(x, act=ACT_NONE, x_masked=x_masked, sign_bits=True)
            x = F.act_blur_pass(z)
            if fade is not None and F.conv_down_fade_ok(x, self.conv1_down.weight.shape[0]):
                return self.conv1_down.forward_nhwc(x, act=ACT_LRELU, x_pre=z, x_pre_bits=zbits, fade=fade), True
            y = self.conv1_down.forward_nhwc(x, act=ACT_LRELU, defer_act=defer_out, x_pre=z, x_pre_bits=zbits)
            return (y, False) if fade is not None else y
        z = self.conv0.forward_nhwc(x, act=ACT_NONE, x_masked=x_masked)   # bias fused in the conv store; pre-activation
        if self.blur._is_121 and self._act == ACT_LRELU:
            x = F.call(F.ActBlurFn, z)                                      # LeakyReLU folded into the blur pass (both ways)
        else:                                                               # ReLU / another blur filter: stage by stage
            x = self.blur.forward_nhwc(apply_act(z, self._act))
        if self._act == ACT_LRELU:
            y = self.conv1_down.forward_nhwc(x, act=ACT_LRELU, defer_act=defer_out)
        else:
            y = apply_act(self.conv1_down.forward_nhwc(x, act=ACT_NONE), self._act)
        return (y, False) if fade is not None else y

    def forward(self, x):
        return F.nchw_view(self.forward_nhwc(F.nhwc(x)))
