"""Transposed convolution + blur (+ mask) in one kernel vs the separate kernels, alone on the GPU.
usage: python tools/upblur_probe.py [B ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import golden_util as gu  # noqa: E402
from stylegan.pytorch_amd import functional as F, native as N  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for B in [int(v) for v in sys.argv[1:]] or [4, 32]:
    for cin, cout, H in [(32, 16, 512), (64, 32, 256), (128, 64, 128), (256, 128, 64), (512, 256, 32)]:
        x = torch.randn(B, H, H, cin, device=DEV).bfloat16()
        w = gu.seeded((cout, cin, 3, 3), 5).to(DEV)
        z = torch.randn(B, 2 * H, 2 * H, cout, device=DEV).bfloat16()
        if not F.conv_blur_ok(x, cout, "U", False):
            print(f"B{B} {cin}->{cout} {H}^2: no fused kernel"); continue
        with torch.no_grad():
            t_c = timeit(lambda: F.ConvFn.apply(x, w, None, "U", 0.1, cin, False, 0))
            y = F.ConvFn.apply(x, w, None, "U", 0.1, cin, False, 0)
            t_b = timeit(lambda: F.BlurFn.apply(y))
            t_bm = timeit(lambda: F.BlurMaskFn.apply(y, z))
            t_f = timeit(lambda: F.ConvBlurFn.apply(x, w, "U", 0.1, cin, False, None))
            t_fm = timeit(lambda: F.ConvBlurFn.apply(x, w, "U", 0.1, cin, False, z))
            t_fb = t_bb = float("nan")
            if cout % 8 == 0:
                bits = (torch.rand(B, 2 * H, 2 * H, cout // 8, device=DEV) * 256).to(torch.uint8)
                t_bb = timeit(lambda: F.BlurMaskFn.apply(y, None, bits))
                t_fb = timeit(lambda: F.ConvBlurFn.apply(x, w, "U", 0.1, cin, False, None, bits))
        print(f"B{B} {cin}->{cout} {H}^2->{2 * H}^2: conv {t_c:7.1f} + blur {t_b:7.1f} = {t_c + t_b:7.1f} | fused {t_f:7.1f} ({t_c + t_b - t_f:+7.1f}) || "
              f"+ blur*mask {t_bm:7.1f} = {t_c + t_bm:7.1f} | fused {t_fm:7.1f} ({t_c + t_bm - t_fm:+7.1f}) || + blur*bits {t_bb:7.1f} = {t_c + t_bb:7.1f} | "
              f"fused {t_fb:7.1f} ({t_c + t_bb - t_fb:+7.1f}) us   [32->16: 'fused' without a mask tensor = the round-5 composite kernel]", flush=True)
