#!/usr/bin/env python3
"""Generator layer epilogue, forward and backward, alone: microseconds per call for the layer shapes of the 1024x1024 step.
    SGX_GEPI_SMALL=0|1 python tools/gepi_probe.py [--batch 4 32] [--reps 20]      (the switch is read once per process)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylegan.pytorch_amd import functional as F  # noqa: E402
from stylegan.pytorch_amd import native as N  # noqa: E402

SHAPES = [(4, 512), (8, 512), (16, 512), (32, 512), (64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)]


def timed(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
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
    ap.add_argument("--min-h", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print("SGX_GEPI_SMALL =", os.environ.get("SGX_GEPI_SMALL", "(unset: 1)"), " SGX_GEPI_RPT =", os.environ.get("SGX_GEPI_RPT", "(unset: 64)"),
          " SGX_GEPI_FOLD =", os.environ.get("SGX_GEPI_FOLD", "(unset: 1)"))
    for B in a.batch:
        for H, C in SHAPES:
            if H < a.min_h:
                continue
            x = torch.randn(B, H, H, C, device=dev).bfloat16().requires_grad_(True)
            noise = torch.randn(B, 1, H, H, device=dev); nw = torch.randn(C, device=dev, requires_grad=True)
            bias = torch.randn(C, device=dev, requires_grad=True); style = torch.randn(B, 2 * C, device=dev, requires_grad=True)
            g = torch.randn(B, H, H, C, device=dev).bfloat16()
            def kernels(fn):
                """(launches, sum of the kernels' own durations in us) of one call: HIP events around every launch, inside the library"""
                fn(); fn(); torch.cuda.synchronize()
                best = None
                for _ in range(5):
                    N.prof_start(1); fn(); torch.cuda.synchronize(); N.prof_start(0)
                    recs = N.prof_records()
                    tot = sum(r[1] for r in recs) * 1e3
                    best = (len(recs), tot) if best is None or tot < best[1] else best
                return best
            with torch.no_grad():
                fwd = lambda: F.call(F.GEpilogueFn, x, bias, noise, nw, style, 3)
                t_f = timed(fwd, a.reps); k_f = kernels(fwd)
            y = F.GEpilogueFn.apply(x, bias, noise, nw, style, 3)
            bwd = lambda: torch.autograd.grad(y, [x, bias, nw, style], g, retain_graph=True)
            t_b = timed(bwd, a.reps); k_b = kernels(bwd)
            print(f"epilogue B{B} {H}x{H} C{C}: forward {k_f[0]} launches {k_f[1]:6.1f} us (back to back {t_f:6.1f})   "
                  f"backward {k_b[0]} launches {k_b[1]:6.1f} us (back to back {t_b:6.1f})", flush=True)


if __name__ == "__main__":
    main()
