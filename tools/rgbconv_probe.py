#!/usr/bin/env python3
"""Times the kernels of csrc/rgbconv.hip alone (the discriminator's composed first layer) at the benchmark shapes, with the
ablation switches of the forward kernel (set through sgx_rgbconv_tune).   python tools/rgbconv_probe.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylegan.pytorch_amd import functional as F  # noqa: E402
from stylegan.pytorch_amd import native as N  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    C, R = 16, 1024
    w0 = torch.randn(C, C, 3, 3, device=dev); b0 = torch.randn(C, device=dev); wr = torch.randn(C, 3, 1, 1, device=dev); br = torch.randn(C, device=dev)
    for B in [int(a) for a in sys.argv[1:]] or [32, 4]:
        img = torch.randn(B, R, R, 3, device=dev)
        gz = torch.randn(B, R, R, C, device=dev).bfloat16()
        px = B * R * R
        rows = []
        with torch.no_grad():
            for dbg, what in ((0, "forward + act + blur + bits"), (8, "  no sign bits"), (4, "  no output stores"), (12, "  no stores at all")):
                N.check(N.lib().sgx_rgbconv_tune(-1, 0, dbg), "sgx_rgbconv_tune")
                us = timeit(lambda: F.RgbConvBlurFn.apply(img, w0, b0, wr, br, 0.1, 0.5))
                rows.append((what, us, px * 46.0 / us / 1e6))
            N.check(N.lib().sgx_rgbconv_tune(-1, 0, 0), "sgx_rgbconv_tune")
            for nit in (3, 4, 5, 6, 8):
                N.check(N.lib().sgx_rgbconv_tune(-1, nit, 0), "sgx_rgbconv_tune")
                us = timeit(lambda: F.RgbConvBlurFn.apply(img, w0, b0, wr, br, 0.1, 0.5))
                rows.append((f"  {6 * nit - 2} rows per wave", us, px * 46.0 / us / 1e6))
            N.check(N.lib().sgx_rgbconv_tune(-1, 0, 0), "sgx_rgbconv_tune")
            for v, what in ((1, "forward, LDS-tile kernel, persistent blocks"), (2, "forward, LDS-tile kernel, one tile per block")):
                N.check(N.lib().sgx_rgbconv_tune(v, 0, 0), "sgx_rgbconv_tune")
                us = timeit(lambda: F.RgbConvBlurFn.apply(img, w0, b0, wr, br, 0.1, 0.5))
                rows.append((what, us, px * 46.0 / us / 1e6))
            N.check(N.lib().sgx_rgbconv_tune(-1, 0, 0), "sgx_rgbconv_tune")
            us = timeit(lambda: F.RgbConvPlainFn.apply(img, w0, wr, br, 0.1, 0.5)); rows.append(("plain convolution (LDS tile kernel)", us, px * 44.0 / us / 1e6))
            us = timeit(lambda: F.RgbConvAdjFn.apply(gz, w0, wr, br, 0.1, 0.5)); rows.append(("image gradient", us, px * 44.0 / us / 1e6))
            us = timeit(lambda: F._rgb_wgrad(img, gz, True, w0, b0, wr, br, 0.1, 0.5, (True,) * 4)); rows.append(("weight gradients (3 launches)", us, px * 44.0 / us / 1e6))
        print(f"batch {B}, {R}x{R}, 3 -> {C} channels")
        for what, us, tbs in rows:
            print(f"  {what:40s} {us:8.1f} us   {tbs:5.2f} TB/s of algorithmic bytes")


if __name__ == "__main__":
    main()
