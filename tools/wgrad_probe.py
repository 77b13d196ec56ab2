#!/usr/bin/env python3
"""Time the weight-gradient path (kernel + finishing pass) per layer shape of the 1024x1024 step.  The kernel generation is
a process-wide switch, so run it twice:   SGX_WGRAD2=0 python tools/wgrad_probe.py ;  SGX_WGRAD2=3 python tools/wgrad_probe.py

    python tools/wgrad_probe.py [--reps 10] [--batch 4 32]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylegan.pytorch_amd import functional as F  # noqa: E402
from stylegan.pytorch_amd import native as N  # noqa: E402

# (mode, H of x, Cin, Cout): the discriminator's conv0 / conv1_down and the generator's conv1 / conv0_up at depth 8
SHAPES = [("S", 1024, 16, 16), ("S", 512, 32, 32), ("S", 256, 64, 64), ("S", 128, 128, 128), ("S", 64, 256, 256), ("S", 32, 512, 512),
          ("D", 1024, 16, 32), ("D", 512, 32, 64), ("D", 256, 64, 128), ("D", 128, 128, 256), ("D", 64, 256, 512), ("D", 32, 512, 512),
          ("U", 512, 32, 16), ("U", 256, 64, 32), ("U", 128, 128, 64), ("U", 64, 256, 128)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, nargs="+", default=[4, 32])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print("SGX_WGRAD2 =", os.environ.get("SGX_WGRAD2", "(default)"))
    for B in a.batch:
        for mode, H, ci, co in SHAPES:
            torch.manual_seed(H + ci)
            w = torch.randn(co, ci, 3, 3, device=dev)
            x = torch.randn(B, H, H, ci, device=dev).bfloat16()
            oh = H // 2 if mode == "D" else (2 * H if mode == "U" else H)
            gy = torch.randn(B, oh, oh, co, device=dev).bfloat16()

            def go():
                return F._wgrad_param(mode, False, x, gy, w, 0.05, want_bias=(mode != "U"))
            go(); go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                go()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            taps, npix = (9, B * H * H) if mode == "S" else (16, B * min(H, oh) ** 2)
            fl = 2.0 * taps * ci * co * npix
            by = 2.0 * (x.numel() + gy.numel())
            print(f"wgrad{mode} B{B} {H}x{H} {ci}->{co}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  {by / us / 1e3:7.0f} GB/s (algorithmic)")


if __name__ == "__main__":
    main()
