#!/usr/bin/env python3
"""Row-streaming forward of the composed first layer against the LDS-tile kernel (sgx_rgbconv_tune(1, 0, 0), an independent implementation of
the same result) and against torch fp64, by image region.   python tools/rgbconv_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as TF  # noqa: E402

from stylegan.pytorch_amd import functional as F  # noqa: E402
from stylegan.pytorch_amd import native as N  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    for B, H, W, C in ((4, 128, 128, 16), (2, 1024, 1024, 16), (4, 64, 64, 32), (3, 48, 192, 16)):
        w0 = torch.randn(C, C, 3, 3, device=dev); b0 = torch.randn(C, device=dev); wr = torch.randn(C, 3, 1, 1, device=dev); br = torch.randn(C, device=dev)
        s0, sr = (2.0 / (C * 9)) ** 0.5, (1.0 / 3) ** 0.5
        img = torch.randn(B, H, W, 3, device=dev).clamp(-1, 1)
        with torch.no_grad():
            N.check(N.lib().sgx_rgbconv_tune(-1, 0, 0), "sgx_rgbconv_tune")
            y, bits = F.RgbConvBlurFn.apply(img, w0, b0, wr, br, s0, sr)
            # every row-block size (6 nit - 2 rows; the host picks by launch size: partial last blocks, one-block images) writes the same bits
            for nit in (6, 5, 4, 3, 2, 1):
                N.check(N.lib().sgx_rgbconv_tune(-1, nit, 0), "sgx_rgbconv_tune")
                yn, bn = F.RgbConvBlurFn.apply(img, w0, b0, wr, br, s0, sr)
                same = torch.equal(yn.view(torch.int16), y.view(torch.int16)) and torch.equal(bn, bits)
                print(f"  B{B} {H}x{W} C{C} rows per block {6 * nit - 2}: {'identical to the default' if same else 'DIFFERENT from the default'}")
            N.check(N.lib().sgx_rgbconv_tune(1, 0, 0), "sgx_rgbconv_tune")
            y1, bits1 = F.RgbConvBlurFn.apply(img, w0, b0, wr, br, s0, sr)
            N.check(N.lib().sgx_rgbconv_tune(-1, 0, 0), "sgx_rgbconv_tune")
            x = img.double().permute(0, 3, 1, 2)
            t = TF.conv2d(x, wr.double() * sr, br.double())
            t = TF.conv2d(t, w0.double() * s0, b0.double(), padding=1)
            a = TF.leaky_relu(t, 0.2)
            k = torch.tensor([1., 2., 1.], device=dev, dtype=torch.float64); k = (k[:, None] * k[None, :] / 16.0)[None, None].repeat(C, 1, 1, 1)
            ref = TF.conv2d(a, k, padding=1, groups=C).permute(0, 2, 3, 1)
        def rel(u, v):
            return float((u.double() - v.double()).norm() / v.double().norm())
        inner = (slice(None), slice(2, H - 2), slice(2, W - 2))
        print(f"B{B} {H}x{W} C{C}: stream vs fp64 {rel(y, ref):.3e} (interior {rel(y[inner], ref[inner]):.3e}); tile vs fp64 {rel(y1, ref):.3e} (interior "
              f"{rel(y1[inner], ref[inner]):.3e}); stream vs tile max |d| {float((y.float() - y1.float()).abs().max()):.3e}, "
              f"rows 0/1/-1 {rel(y[:, 0], ref[:, 0]):.2e}/{rel(y[:, 1], ref[:, 1]):.2e}/{rel(y[:, -1], ref[:, -1]):.2e} cols 0/-1 {rel(y[:, :, 0], ref[:, :, 0]):.2e}/{rel(y[:, :, -1], ref[:, :, -1]):.2e}; "
              f"bits differ {int((bits != bits1).sum())} of {bits.numel()}")


if __name__ == "__main__":
    main()
