"""3x3 weight gradient of the 16 x 16-channel layers at 1024^2, alone on the GPU: first generation vs wgrad16_s_kernel
(kernel + finishing passes).  usage: [SGX_WGRAD2=3|7] python tools/wgrad16_probe.py [B ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import golden_util as gu  # noqa: E402
from stylegan.pytorch_amd import functional as F, native as N  # noqa: E402

DEV = "cuda:0"
for B in [int(v) for v in sys.argv[1:]] or [4, 32]:
    for C, H in [(16, 1024), (16, 512)]:
        x = torch.randn(B, H, H, C, device=DEV).bfloat16(); gy = torch.randn(B, H, H, C, device=DEV).bfloat16()
        w = gu.seeded((C, C, 3, 3), 5).to(DEV)

        def run():
            F._wgrad_param("S", False, x, gy, w, 0.1, True)
        for _ in range(3):
            run()
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 20
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t) / n * 1e6
        print(f"SGX_WGRAD2={os.environ.get('SGX_WGRAD2', '7')} B{B} {C}x{C} {H}^2: {us:7.1f} us  {4.0 * B * H * H * C / us / 1e6:5.2f} TB/s (both activations once)", flush=True)
