#!/usr/bin/env python3
"""Times the depthwise-blur passes alone (csrc/pointwise.hip: blur3x3_kernel with three loads per input row, blur3x3s_kernel with one
load + lane exchange at prefetch depth SGX_BLUR_SHFL = 1..4; the switch is read at every launch) at the benchmark's batch-32 shapes, and
checks that every variant writes the same bits.   python tools/blur_probe.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylegan.pytorch_amd import native as N  # noqa: E402


def timeit(fn, n=12):
    for _ in range(2):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = N.lib()
    print(f"batch {B}; us (TB/s of algorithmic bytes) per variant: 0 = three loads, 1..4 = one load + lane exchange at that prefetch depth")
    for dt, C, R in ((torch.bfloat16, 16, 1024), (torch.bfloat16, 32, 512), (torch.bfloat16, 64, 256), (torch.bfloat16, 128, 128), (torch.float32, 16, 512)):
        x = torch.randn(B, R, R, C, device=dev).to(dt)
        z = torch.randn(B, R, R, C, device=dev).to(dt)
        bits = torch.randint(0, 256, (B, R, R, C // 8), device=dev, dtype=torch.uint8)
        y = torch.empty_like(x)
        esz = x.element_size()
        modes = [(0, None), (1, None), (2, z)] + ([(4, bits), (5, bits)] if dt == torch.bfloat16 else [])
        for mode, aux in modes:
            def run():
                if mode >= 4:
                    N.check(L.sgx_blur3x3_bits(N.ptr(x), N.ptr(aux), N.ptr(y), B, R, R, C, mode - 2, N.dt(x), N.stream()), "blur_bits")
                else:
                    N.check(L.sgx_blur3x3_act(N.ptr(x), N.ptr(aux), N.ptr(y), B, R, R, C, mode, N.dt(x), N.stream()), "blur_act")
            nbytes = x.numel() * esz * (3.0 if mode == 2 else 2.0) + (bits.numel() if mode >= 4 else 0)
            out, ref = [], None
            for v in (0, 1, 2, 3, 4):
                os.environ["SGX_BLUR_SHFL"] = str(v)
                us = timeit(run)
                got = y.clone()
                if ref is None:
                    ref = got
                same = bool(torch.equal(got.view(torch.int16 if esz == 2 else torch.int32), ref.view(torch.int16 if esz == 2 else torch.int32)))
                out.append(f"{us:7.1f} ({nbytes / us / 1e6:4.2f}){'' if same else ' DIFFERENT'}")
            print(f"  {str(dt)[6:]:9s} {R:4d}^2 C{C:<4d} mode {mode}: " + "  ".join(out), flush=True)
    os.environ.pop("SGX_BLUR_SHFL")


if __name__ == "__main__":
    main()
