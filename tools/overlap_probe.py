#!/usr/bin/env python3
"""Do an MFMA-bound convolution and a bandwidth-bound pass share the chip when they sit on two HIP streams?  (round 6)

Per pair (convolution shape, streaming pass): N launches of each, (a) all on one stream, (b) the convolutions on one stream and the passes on
another, (c) each alone.  If (b) ~ max of the two alone times the two kernel classes overlap and the step's stream assignment can pair them;
if (b) ~ (a) the hardware serialises them whatever the streams say.

    python tools/overlap_probe.py [--reps 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylegan.pytorch_amd import functional as F  # noqa: E402
from stylegan.pytorch_amd import native as N  # noqa: E402

GEO = {"S": 0, "D": 1, "U": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--priority", type=int, default=0, help="1: the convolutions' stream gets HIP's high priority")
    ap.add_argument("--stream-kind", default="blur", help="blur | add (an aten elementwise kernel, huge grid)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = N.lib()
    B = a.batch
    ap_pri = a.priority
    s1, s2 = torch.cuda.Stream(priority=-1 if ap_pri else 0), torch.cuda.Stream()

    def conv_launcher(H, ci, co):
        w = torch.randn(co, ci, 3, 3, device=dev)
        bias = torch.randn(co, device=dev)
        x = torch.randn(B, H, H, ci, device=dev).bfloat16()
        wq, _ = F.packs(w, "S", 0.05, ci, torch.bfloat16)
        y = torch.empty((B, H, H, co), dtype=x.dtype, device=dev)

        def go():
            N.check(L.sgx_conv_variant(0, N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), B, H, H, ci, co, 1, N.BF16, 8, N.stream()), "sgx_conv_variant")
        return go, (x, wq, bias, y)

    def stream_launcher(H, C):
        # the 3x3 blur of the step (a read + a write of the tensor, no LDS tile larger than a few rows)
        x = torch.randn(B, H, H, C, device=dev).bfloat16()
        y = torch.empty_like(x)

        def go():
            if a.stream_kind == "add":
                torch.add(x, 1.0, out=y)
            else:
                N.check(L.sgx_blur3x3(N.ptr(x), N.ptr(y), B, H, H, C, N.BF16, N.stream()), "sgx_blur3x3")
        return go, (x, y)

    def timed(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3

    sig = None
    for (H, ci, co), (SH, SC) in [((128, 128, 128), (512, 32)), ((64, 256, 256), (512, 32)), ((32, 512, 512), (256, 64)), ((256, 64, 64), (512, 32)),
                                  ((128, 128, 128), (1024, 16))]:
        cgo, keep1 = conv_launcher(H, ci, co)
        try:
            sgo, keep2 = stream_launcher(SH, SC)
            sgo()
        except Exception as e:                              # signature drift: report and stop
            print("blur launcher failed:", e)
            return
        cgo(); torch.cuda.synchronize()
        n = a.reps

        def alone_c():
            for _ in range(n): cgo()

        def alone_s():
            for _ in range(n): sgo()

        def serial():
            for _ in range(n): cgo(); sgo()

        def two():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                for _ in range(n): cgo()
            with torch.cuda.stream(s2):
                for _ in range(n): sgo()
            cur.wait_stream(s1); cur.wait_stream(s2)

        for f in (alone_c, alone_s, serial, two):
            f()
        tc, ts, tser, ttwo = (min(timed(f) for _ in range(3)) / n for f in (alone_c, alone_s, serial, two))
        print(f"[priority {ap_pri} {a.stream_kind}] B{B} convS {H}^2 {ci}->{co} {tc:7.1f} us alone | blur {SH}^2 C{SC} {ts:7.1f} us alone | one stream {tser:7.1f} | two streams {ttwo:7.1f} "
              f"(sum {tc + ts:.1f}, max {max(tc, ts):.1f}; overlap gain {100 * (1 - ttwo / tser):.0f} %)", flush=True)


if __name__ == "__main__":
    main()
