#!/usr/bin/env python3
"""A/B probe of the two convolution kernel generations (sgx_conv3x3_variant): per layer shape of the 1024x1024 step,
the result of every variant against an fp32 torch convolution of the same bf16 operands, and its time.

    python tools/conv2_probe.py [--reps 10] [--batch 4 32]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylegan.pytorch_amd import functional as F  # noqa: E402
from stylegan.pytorch_amd import native as N  # noqa: E402

SHAPES = [(256, 64, 64), (128, 128, 128), (64, 256, 256), (32, 512, 512), (512, 32, 64), (128, 64, 128), (64, 128, 256)]


def run(variant, x, wq, bias, act, reps):
    B, H, W, Cin = x.shape
    Cout = wq.shape[1]
    y = torch.empty((B, H, W, Cout), dtype=x.dtype, device=x.device)
    L = N.lib()

    def go():
        N.check(L.sgx_conv3x3_variant(N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), B, H, W, Cin, Cout, act, N.BF16, variant, N.stream()),
                "sgx_conv3x3_variant")
    go(); go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record(); torch.cuda.synchronize()
    return y, e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, nargs="+", default=[4, 32])
    ap.add_argument("--variants", type=int, nargs="+", default=[0, 4, 8])
    ap.add_argument("--check", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for B in a.batch:
        for H, ci, co in SHAPES:
            torch.manual_seed(H + ci)
            w = torch.randn(co, ci, 3, 3, device=dev)
            bias = torch.randn(co, device=dev)
            x = torch.randn(B, H, H, ci, device=dev).bfloat16()
            wq, _ = F.packs(w, "S", 0.05, ci, torch.bfloat16)
            ref = None
            if a.check:
                wr = wq.float().view(3, 3, co, ci).permute(2, 3, 0, 1).contiguous()          # the bf16-rounded operands
                nb = min(B, 2)
                ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x[:nb].float().permute(0, 3, 1, 2), wr, bias, padding=1), 0.2)
                ref = ref.permute(0, 2, 3, 1)
            fl = 2.0 * 9 * ci * co * B * H * H
            line = f"convS B{B} {H}x{H} {ci}->{co}:"
            for v in a.variants:
                try:
                    y, us = run(v, x, wq, bias, 1, a.reps)
                except N.SgxError as e:
                    line += f"  v{v}: n/a ({str(e)[:40]})"
                    continue
                err = ""
                if ref is not None:
                    d = (y[:ref.shape[0]].float() - ref)
                    rel = (d.norm() / ref.norm()).item()
                    mx = (d.abs().max() / ref.abs().max()).item()
                    err = f" rel {rel:.1e} max {mx:.1e}"
                line += f"  v{v}: {us:7.1f} us {fl / us / 1e6:7.1f} TF{err}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
