#!/usr/bin/env python3
"""Split-K (sgx_conv_splitk, round 6) against the unsplit first-generation launch, alone, per low-resolution layer shape of the 1024 model.
    python tools/splitk_probe.py [--batch 4 32] [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylegan.pytorch_amd import functional as F  # noqa: E402
from stylegan.pytorch_amd import native as N  # noqa: E402

SHAPES = [("S", 4, 512, 512), ("S", 8, 512, 512), ("S", 16, 512, 512), ("S", 32, 512, 512),
          ("D", 64, 256, 512), ("D", 32, 512, 512), ("D", 16, 512, 512), ("D", 8, 512, 512),
          ("U", 4, 512, 512), ("U", 8, 512, 512), ("U", 16, 512, 512), ("U", 32, 512, 256)]


def timed(fn, reps):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[4, 32])
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = N.lib()
    for B in a.batch:
        for geo, H, ci, co in SHAPES:
            gi = "SDU".index(geo)
            k = 3 if geo == "S" else 4
            w = torch.randn(co, ci, k, k, device=dev)
            x = torch.randn(B, H, H, ci, device=dev).bfloat16()
            wq, _ = F.packs(w, geo, 0.05, ci, torch.bfloat16)
            bias = None if geo == "U" else torch.randn(co, device=dev)
            act = 0 if geo == "U" else 1
            oh = H if geo == "S" else (H // 2 if geo == "D" else 2 * H)
            y = torch.empty((B, oh, oh, co), dtype=torch.bfloat16, device=dev)

            def plain():
                if geo == "S":
                    N.check(L.sgx_conv3x3(N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), B, H, H, ci, co, act, None, N.BF16, N.stream()), "conv")
                elif geo == "D":
                    N.check(L.sgx_conv4x4s2_down(N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), B, H, H, ci, co, act, N.BF16, N.stream()), "conv")
                else:
                    N.check(L.sgx_conv4x4s2_up(N.ptr(x), N.ptr(wq), N.ptr(y), B, H, H, ci, co, N.BF16, N.stream()), "conv")
            t0 = timed(plain, a.reps)
            wsb = L.sgx_conv_splitk_ws_bytes(gi, B, H, H, ci, co, N.BF16)
            if not wsb:
                print(f"conv{geo} B{B} {H}x{H} {ci}->{co}: unsplit {t0:6.1f} us   (no split planned)", flush=True)
                continue
            ws = N.workspace(wsb, x.device)
            opix = B * oh * oh
            ks = wsb // (opix * co * 4)

            def split():
                N.check(L.sgx_conv_splitk(gi, N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), None, B, H, H, ci, co, act, N.BF16, N.ptr(ws), wsb, N.stream()), "splitk")
            t1 = timed(split, a.reps)
            print(f"conv{geo} B{B} {H}x{H} {ci}->{co}: unsplit {t0:6.1f} us   split-K x{ks} {t1:6.1f} us (both launches)", flush=True)


if __name__ == "__main__":
    main()
