#!/usr/bin/env python3
"""Micro-probe for the convolution kernels: runs a few layer shapes of the 1024x1024 step repeatedly so that
rocprofv3 (kernel trace or --pmc counters) sees them in isolation.

    python tools/conv_probe.py [--reps 20] [--dtype bf16] [--batch 4]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylegan.pytorch_amd import functional as F  # noqa: E402

SHAPES = [  # (mode, H, Cin, Cout)
    ("S", 256, 64, 64), ("S", 128, 128, 128), ("S", 64, 256, 256), ("S", 32, 512, 512), ("S", 1024, 16, 16), ("S", 512, 32, 32),
    ("U", 128, 128, 64), ("U", 512, 32, 16), ("D", 256, 64, 128), ("D", 1024, 16, 32),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--wgrad", action="store_true")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = torch.device("cuda:0")
    for mode, H, ci, co in SHAPES:
        w = torch.nn.Parameter(torch.randn(co, ci, 3, 3, device=dev))
        x = torch.randn(a.batch, H, H, ci, device=dev).to(dt).requires_grad_(a.wgrad)
        y = None
        for _ in range(a.reps):
            y = F.conv(x, w, None, mode, 0.05)
            if a.wgrad:
                y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            for _ in range(a.reps):
                y = F.conv(x, w, None, mode, 0.05)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        taps = 9 if mode == "S" else 16
        npix = a.batch * H * H * (0.25 if mode == "D" else 1.0)
        fl = 2.0 * taps * ci * co * npix
        es = 2 if a.dtype == "bf16" else 4
        oh = H // 2 if mode == "D" else (2 * H if mode == "U" else H)
        by = es * a.batch * (H * H * ci + oh * oh * co)
        print(f"conv{mode} B{a.batch} {H}x{H} {ci}->{co}: {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s  {by / us / 1e3:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
