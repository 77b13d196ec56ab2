#!/usr/bin/env python3
"""Fixture for the uint8 input pipeline: PNG bytes decoded by PIL, then the reference's transform chain
data/transforms.py:27-32 (``RandomHorizontalFlip(), ToTensor(), Normalize((.5,.5,.5), (.5,.5,.5))``, no Resize).

torchvision is NOT installed in this image (and cannot be: no network), so the chain is restated here operation by
operation as torchvision publishes it -- independent of oracle/stylegan_oracle.py, which the tests compare AGAINST this
file's output:
  * ``F.hflip(PIL)``            = ``img.transpose(Image.FLIP_LEFT_RIGHT)``              (transforms/_functional_pil.py hflip)
  * ``F.to_tensor(PIL 'RGB')``  = ``torch.from_numpy(np.array(pic, np.uint8, copy=True)).view(H, W, 3).permute(2, 0, 1)
                                     .contiguous().to(torch.float32).div(255)``          (transforms/functional.py to_tensor)
  * ``F.normalize(t, m, s)``    = ``t.sub_(mean[:, None, None]).div_(std[:, None, None])`` with fp32 mean/std tensors
                                                                                         (transforms/functional.py normalize)
Writes images_u8.npz: the decoded uint8 batch [B,H,W,3], the flip decisions, and the fp32 [B,3,H,W] result.

    python tests/golden/make_golden_images.py
"""
import io
import os

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(7)
H, W, B = 24, 40, 4
raw = rng.randint(0, 256, size=(B, H, W, 3), dtype=np.uint8)
raw[0, :, :, :] = (np.arange(H * W * 3).reshape(H, W, 3) % 256).astype(np.uint8)          # every level 0..255
flips = [False, True, True, False]
u8, out = [], []
for i in range(B):
    buf = io.BytesIO()
    Image.fromarray(raw[i], "RGB").save(buf, format="PNG")                                 # lossless: decode == raw
    pic = Image.open(io.BytesIO(buf.getvalue())).convert("RGB")
    u8.append(np.array(pic, np.uint8, copy=True))
    if flips[i]:
        pic = pic.transpose(Image.FLIP_LEFT_RIGHT)
    t = torch.from_numpy(np.array(pic, np.uint8, copy=True)).view(pic.size[1], pic.size[0], 3).permute(2, 0, 1).contiguous()
    t = t.to(dtype=torch.float32).div(255)
    mean = torch.as_tensor((0.5, 0.5, 0.5), dtype=torch.float32); std = torch.as_tensor((0.5, 0.5, 0.5), dtype=torch.float32)
    t = t.sub_(mean[:, None, None]).div_(std[:, None, None])
    out.append(t.numpy())
np.savez_compressed(os.path.join(HERE, "images_u8.npz"), u8=np.stack(u8), flips=np.array(flips), out=np.stack(out))
print("wrote images_u8.npz", np.stack(out).shape, float(np.stack(out).min()), float(np.stack(out).max()))
