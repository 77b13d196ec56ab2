"""The 16-channel layers of the 1024x1024 level, alone on the GPU: first generation vs conv2 (4 / 8 waves), per batch size.
usage: python tools/conv16_probe.py [B ...]"""
import math
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
CASES = [("S", 16, 16, 1024), ("D", 16, 32, 1024), ("U", 32, 16, 512), ("S", 32, 32, 512)]
for B in [int(v) for v in sys.argv[1:]] or [4, 32]:
    for geo, cin, cout, H in CASES:
        w = gu.seeded((cout, cin, 3, 3), 5).to(DEV)
        wq, _ = F.packs(w, geo, 0.1, cin, torch.bfloat16)
        x = torch.randn(B, H, H, cin, device=DEV).bfloat16()
        OH = H // 2 if geo == "D" else (2 * H if geo == "U" else H)
        y = torch.empty(B, OH, OH, cout, dtype=torch.bfloat16, device=DEV)
        bias = None if geo == "U" else torch.zeros(cout, device=DEV)
        nbytes = 2.0 * B * (H * H * cin + OH * OH * cout)
        row = [f"{geo} {cin}->{cout} {H}^2 B{B}"]
        for v in (0, 4, 8):
            def run():
                N.check(L.sgx_conv_variant({"S": 0, "D": 1, "U": 2}[geo], N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), B, H, H, cin, cout,
                                           0, N.BF16, v, N.stream()), "variant")
            try:
                for _ in range(3):
                    run()
                torch.cuda.synchronize(); t = time.perf_counter()
                n = 20
                for _ in range(n):
                    run()
                torch.cuda.synchronize()
                us = (time.perf_counter() - t) / n * 1e6
                row.append(f"v{v}: {us:7.1f} us {nbytes / us / 1e6:5.2f} TB/s")
            except Exception as e:                                     # noqa: BLE001
                row.append(f"v{v}: {type(e).__name__}")
        print("   ".join(row), flush=True)
