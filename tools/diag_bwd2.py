"""diagnostic (round 6): parameter gradients of the D loss + R1 with functional.FadeRgbBwdFn on / off / on with its backward composed of the old passes"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import golden_util as gu
from gpu_util import DEV, rel_err
from test_gpu_rgbconv import build_dis, d_loss_grads
from stylegan.pytorch_amd import functional as F
from torch.autograd.function import once_differentiable
depth, B, alpha = 5, 16, 0.3
R = 4 << depth
real = gu.seeded((B, 3, R, R), 61); fake = gu.seeded((B, 3, R, R), 62)
dis, dp = build_dis()
orig_bwd = F.FadeRgbBwdFn.backward

def composed(ctx, ggy, ggp):
    g, bits, wr, alpha_dev = ctx.saved_tensors
    ws, al, be, need_img = ctx.cfg
    t1 = None if ggy is None else F.LReluBwdBitsFn.forward(F._NoGradCtx(), ggy, bits, 0.2, al)
    t2 = None if ggp is None else F.RgbInFn.forward(F._NoGradCtx(), ggp, wr, None, ws * be, g.dtype)
    out = t1 if t2 is None else (t2 if t1 is None else t1 + t2)
    gwr = None
    if ggp is not None and ctx.needs_input_grad[3]:
        gwr = F.RgbWgradFn.forward(F._NoGradCtx(), ggp.contiguous(), g, wr, ws * be)
    print("   [composed backward] ggy", None if ggy is None else tuple(ggy.shape), "ggp", None if ggp is None else tuple(ggp.shape), "gwr", gwr is not None)
    return out, None, None, gwr, None, None, None, None, None

res = {}
for tag in ("off", "on", "on-composed", "off2"):
    F.FUSE_FADE_BWD2 = tag.startswith("on")
    F.FadeRgbBwdFn.backward = staticmethod(once_differentiable(composed)) if tag == "on-composed" else orig_bwd
    res[tag] = d_loss_grads(dis, real.to(DEV), fake.to(DEV), depth, alpha)
for a, b in (("off", "off2"), ("on", "off"), ("on-composed", "off"), ("on", "on-composed")):
    worst = max(rel_err(res[a][1][k], res[b][1][k]) for k in res[a][1])
    k0 = "from_rgb.1.weight"
    print(f"{a:12s} vs {b:12s}: loss {res[a][0]:.8f} {res[b][0]:.8f}  worst param rel {worst:.3e}  img grad rel {rel_err(res[a][2], res[b][2]):.3e}  {k0} {rel_err(res[a][1][k0], res[b][1][k0]):.3e}")

# ---- checksums of what reaches the tail's second-order backward in both structures
def cs(t):
    return None if t is None else (tuple(t.shape), str(t.dtype).replace("torch.", ""), float(t.double().abs().sum()), float(t.double().sum()))
ob1, ob2 = F.LReluBwdBitsFn.backward, F.RgbOutFn.backward
def b1(ctx, gg):
    out = ob1(ctx, gg)
    if gg is not None and gg.dim() == 4 and not torch.is_grad_enabled(): print("   [off] LReluBwdBits.backward in", cs(gg), "out", cs(out[0]))
    return out
def b2(ctx, gg):
    out = ob2(ctx, gg)
    if not torch.is_grad_enabled(): print("   [off] RgbOut.backward in", cs(gg), "gx", cs(out[0]), "gw", cs(out[1]))
    return out
def b3(ctx, ggy, ggp):
    out = orig_bwd(ctx, ggy, ggp)
    print("   [on] FadeRgbBwd.backward ggy", cs(ggy), "ggp", cs(ggp), "out", cs(out[0]), "gwr", cs(out[3]))
    return out
F.LReluBwdBitsFn.backward = staticmethod(b1); F.RgbOutFn.backward = staticmethod(b2); F.FadeRgbBwdFn.backward = staticmethod(b3)
for tag in ("off", "on"):
    F.FUSE_FADE_BWD2 = tag == "on"
    print(tag); d_loss_grads(dis, real.to(DEV), fake.to(DEV), depth, alpha)
