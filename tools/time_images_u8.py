#!/usr/bin/env python3
"""Per-launch time and algorithmic GB/s of sgx_images_u8_to_nhwc at the headline batch (library profiler, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylegan.pytorch_amd import functional as F, native
dev = torch.device("cuda:0")
u8 = torch.randint(0, 256, (4, 1024, 1024, 3), dtype=torch.uint8, device=dev)
flip = [True, False, True, False]
for layout, src in (("hwc", u8), ("chw", u8.permute(0, 3, 1, 2).contiguous())):
    for dt in (torch.float32, torch.bfloat16):
        for _ in range(3):
            F.images_from_uint8(src, flip=flip, out_dtype=dt, layout=layout)
        torch.cuda.synchronize()
        native.prof_start(1)
        for _ in range(20):
            F.images_from_uint8(src, flip=flip, out_dtype=dt, layout=layout)
        torch.cuda.synchronize()
        native.prof_start(0)
        recs = native.prof_records()
        ms = sorted(r[1] for r in recs)[len(recs) // 2]
        print(f"{layout} -> {str(dt).split('.')[-1]}: median {ms * 1e3:.1f} us, {recs[0][3] / ms / 1e6:.0f} GB/s algorithmic ({recs[0][3] / 1e6:.1f} MB)")
