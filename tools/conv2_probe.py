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

SHAPES = [  # (geo, H (input), Cin, Cout)
    ("S", 512, 32, 32), ("S", 256, 64, 64), ("S", 128, 128, 128), ("S", 64, 256, 256), ("S", 32, 512, 512), ("S", 512, 32, 64),
    ("D", 512, 32, 64), ("D", 256, 64, 128), ("D", 128, 128, 256), ("D", 64, 256, 512), ("D", 512, 32, 32),
    ("U", 256, 64, 32), ("U", 128, 128, 64), ("U", 64, 256, 128), ("U", 32, 512, 256), ("U", 256, 32, 32),
]
LOWRES = [("S", 16, 512, 512), ("S", 8, 512, 512), ("S", 4, 512, 512), ("D", 64, 256, 512), ("D", 32, 512, 512), ("D", 16, 512, 512), ("D", 8, 512, 512),
          ("U", 4, 512, 512), ("U", 8, 512, 512), ("U", 16, 512, 512)]          # the latency-bound layers (first generation: --variants 0)
GEO = {"S": 0, "D": 1, "U": 2}


def run(geo, variant, x, wq, bias, act, reps, cold=0):
    B, H, W, Cin = x.shape
    Cout = wq.shape[1]
    oh = H // 2 if geo == "D" else (2 * H if geo == "U" else H)
    y = torch.empty((B, oh, oh, Cout), dtype=x.dtype, device=x.device)
    L = N.lib()
    # --cold N: rotate through N copies of every operand, so that a launch finds none of them in L2 / the memory-side cache (as in
    # the training step, where the operands were last touched many launches ago); 0 = the same buffers every time (all cache hits)
    xs = [x] + [x.clone() for _ in range(cold)]
    ws = [wq] + [wq.clone() for _ in range(cold)]
    ys = [y] + [torch.empty_like(y) for _ in range(cold)]

    def go(i=0):
        k = i % len(xs)
        N.check(L.sgx_conv_variant(GEO[geo], N.ptr(xs[k]), N.ptr(ws[k]), N.ptr(bias), N.ptr(ys[k]), B, H, W, Cin, Cout, act, N.BF16, variant, N.stream()),
                "sgx_conv_variant")
    go(); go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        go(i)
    e1.record(); torch.cuda.synchronize()
    return y, e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, nargs="+", default=[4, 32])
    ap.add_argument("--variants", type=int, nargs="+", default=[0, 4, 8])
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--geo", nargs="+", default=["S", "D", "U"])
    ap.add_argument("--lowres", action="store_true", help="the 512-channel layers at 4^2..32^2 instead of the default shapes")
    ap.add_argument("--cold", type=int, default=0, help="rotate through this many extra copies of the operands (cache-cold launches)")
    ap.add_argument("--only-h", type=int, default=0, help="only the shapes of this input height (counter passes: one shape per kernel name)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    only = set(a.geo)
    for B in a.batch:
        for geo, H, ci, co in (LOWRES if a.lowres else SHAPES):
            if geo not in only or (a.only_h and H != a.only_h):
                continue
            torch.manual_seed(H + ci)
            w = torch.randn(co, ci, 3, 3, device=dev)
            bias = None if geo == "U" else torch.randn(co, device=dev)
            x = torch.randn(B, H, H, ci, device=dev).bfloat16()
            wq, _ = F.packs(w, geo, 0.05, ci, torch.bfloat16)
            taps = 9 if geo == "S" else 16
            ref = None
            if a.check:
                k = 3 if geo == "S" else 4
                wr = wq.float().view(k, k, co, ci).permute(2, 3, 0, 1).contiguous()          # the bf16-rounded operands
                nb = min(B, 2)
                xi = x[:nb].float().permute(0, 3, 1, 2)
                if geo == "S":
                    ref = torch.nn.functional.conv2d(xi, wr, bias, padding=1)
                elif geo == "D":
                    ref = torch.nn.functional.conv2d(xi, wr, bias, stride=2, padding=1)
                else:
                    ref = torch.nn.functional.conv_transpose2d(xi, wr.permute(1, 0, 2, 3), stride=2, padding=1)
                if geo != "U":
                    ref = torch.nn.functional.leaky_relu(ref, 0.2)
                ref = ref.permute(0, 2, 3, 1)
            npix = B * H * H * (0.25 if geo == "D" else 1.0)
            fl = 2.0 * taps * ci * co * npix
            line = f"conv{geo} B{B} {H}x{H} {ci}->{co}:"
            first = None
            for v in a.variants:
                try:
                    y, us = run(geo, v, x, wq, bias, 0 if geo == "U" else 1, a.reps, a.cold)
                except N.SgxError as e:
                    line += f"  v{v}: n/a"
                    continue
                err = ""
                if v >= 4:                      # second / third generation kernels accumulate in the same order: bit-identical outputs
                    if first is None:
                        first = (v, y.clone())
                    else:
                        err = f" {'==' if torch.equal(y.view(torch.int16), first[1].view(torch.int16)) else '!='}v{first[0]}"
                if ref is not None:
                    d = (y[:ref.shape[0]].float() - ref)
                    rel = (d.norm() / ref.norm()).item()
                    err = f" rel {rel:.1e}" + err
                line += f"  v{v}: {us:7.1f} us {fl / us / 1e6:7.1f} TF{err}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
