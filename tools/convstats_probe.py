"""Statistics out of the producing kernel vs the separate statistics pass, alone on the GPU.
rows: conv = plain 3x3, conv+s = the same kernel with the statistics epilogue, epi = sgx_gepi_fwd (statistics + apply),
epi(pre) = sgx_gepi_fwd fed with the producer's partials (apply only); blur / blur+s the same for the blur after conv0_up.
usage: python tools/convstats_probe.py [B ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import golden_util as gu  # noqa: E402
from stylegan.pytorch_amd import functional as F, native as N  # noqa: E402

DEV = "cuda:0"
L = N.lib()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for B in [int(v) for v in sys.argv[1:]] or [4, 32]:
    for C, H in [(16, 1024), (32, 512), (64, 256), (128, 128), (256, 64)]:
        x = torch.randn(B, H, H, C, device=DEV).bfloat16()
        w = gu.seeded((C, C, 3, 3), 5).to(DEV)
        bias = torch.zeros(C, device=DEV); nw = 0.1 * torch.ones(C, device=DEV)
        noise = torch.randn(B, 1, H, H, device=DEV); style = torch.zeros(B, 2 * C, device=DEV)
        with torch.no_grad():
            t_conv = timeit(lambda: F.ConvFn.apply(x, w, None, "S", 0.1, C, False, 0))
            np_ = F.conv_stats_nparts(x, C)
            t_convs = timeit(lambda: F.ConvFn.apply(x, w, None, "S", 0.1, C, False, 0, None, False, False, (bias, noise, nw))) if np_ else float("nan")
            y, part = F.ConvFn.apply(x, w, None, "S", 0.1, C, False, 0, None, False, False, (bias, noise, nw)) if np_ else (x, None)
            t_epi = timeit(lambda: F.GEpilogueFn.apply(y, bias, noise, nw, style))
            t_epip = timeit(lambda: F.GEpilogueFn.apply(y, bias, noise, nw, style, 3, part)) if np_ else float("nan")
            t_blur = timeit(lambda: F.BlurFn.apply(x))
            t_blurs = timeit(lambda: F.BlurStatsFn.apply(x, bias, noise, nw))
        print(f"B{B} C{C} {H}^2: conv {t_conv:7.1f}  conv+s {t_convs:7.1f}  epi {t_epi:7.1f}  epi(pre) {t_epip:7.1f}  | conv path gain "
              f"{t_conv + t_epi - t_convs - t_epip:+7.1f} us | blur {t_blur:7.1f}  blur+s {t_blurs:7.1f}  | blur path gain {t_blur + t_epi - t_blurs - t_epip:+7.1f} us (nparts {np_})", flush=True)
