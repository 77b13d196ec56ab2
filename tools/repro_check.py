#!/usr/bin/env python3
"""Which fusion makes the full step non-reproducible?  Runs tests/test_gpu_fullsize.one_step twice from identical state and prints the
losses; the environment selects the kernels (SGX_CONV_UPBLUR, SGX_FUSE_FADE_RGB).  Then: kernel-level repeats of sgx_conv_upblur with
sign bits on one input (bitwise)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import test_gpu_fullsize as T  # noqa: E402
from stylegan.pytorch_amd import functional as F  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "kernel":
    dev = "cuda:0"
    torch.manual_seed(1)
    for B, H in ((4, 512), (32, 512)):
        x = torch.randn(B, H, H, 32, device=dev).bfloat16()
        w = torch.randn(16, 32, 3, 3, device=dev)
        bits = (torch.rand(B, 2 * H, 2 * H, 2, device=dev) * 256).to(torch.uint8)
        with torch.no_grad():
            ref_b = F.ConvBlurFn.apply(x, w, "U", 0.1, 32, False, None, bits).clone()
            ref_p = F.ConvBlurFn.apply(x, w, "U", 0.1, 32, False, None).clone()
            bad_b = bad_p = 0
            for i in range(30):
                # other work in between, so that timing varies
                _ = torch.randn(1 << (18 + i % 6), device=dev).sum()
                yb = F.ConvBlurFn.apply(x, w, "U", 0.1, 32, False, None, bits)
                yp = F.ConvBlurFn.apply(x, w, "U", 0.1, 32, False, None)
                bad_b += int(not torch.equal(yb, ref_b)); bad_p += int(not torch.equal(yp, ref_p))
        print(f"upblur B{B} {H}^2: runs that differ from the first: with bits {bad_b}/30, plain {bad_p}/30", flush=True)
    sys.exit(0)

outs = []
for _ in range(3):
    sg = T.build(torch.bfloat16, seed=3)
    d, g = T.one_step(sg, 11)
    outs.append((float(d), float(g)))
    del sg
    torch.cuda.empty_cache()
print({k: os.environ.get(k) for k in ("SGX_CONV_UPBLUR", "SGX_FUSE_FADE_RGB")}, outs, "REPRODUCIBLE" if len(set(outs)) == 1 else "DIFFERS", flush=True)
