#!/usr/bin/env python3
"""The blur passes alone under the library's own launch policy (one process per environment: SGX_BLUR_ROWS / SGX_GRID_ALL_CAP are read once).
   python tools/blur_rows_probe.py [B]"""
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
    tag = f"rows {os.environ.get('SGX_BLUR_ROWS', '8')} cap {os.environ.get('SGX_GRID_ALL_CAP', 'none')}"
    for C, R in ((16, 1024), (32, 512), (64, 256)):
        x = torch.randn(B, R, R, C, device=dev).bfloat16()
        bits = torch.randint(0, 256, (B, R, R, C // 8), device=dev, dtype=torch.uint8)
        y = torch.empty_like(x)
        out = []
        for mode, aux in ((0, None), (1, None), (4, bits), (5, bits)):
            def run():
                if mode >= 4:
                    N.check(L.sgx_blur3x3_bits(N.ptr(x), N.ptr(aux), N.ptr(y), B, R, R, C, mode - 2, N.dt(x), N.stream()), "blur_bits")
                else:
                    N.check(L.sgx_blur3x3_act(N.ptr(x), N.ptr(aux), N.ptr(y), B, R, R, C, mode, N.dt(x), N.stream()), "blur_act")
            nbytes = x.numel() * 2 * 2.0 + (bits.numel() if mode >= 4 else 0)
            us = timeit(run)
            out.append(f"mode {mode} {us:7.1f} us {nbytes / us / 1e6:4.2f} TB/s")
        print(f"[{tag}] B{B} {R}^2 C{C}: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
