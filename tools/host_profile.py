#!/usr/bin/env python3
"""Host-side profile of the training iteration (cProfile over a few steps, GPU running asynchronously): shows where the
Python/ctypes/autograd launch path spends its time when the step becomes launch-bound.
    python tools/host_profile.py [steps] [single] [depth=D] [batch=B]
("single": one stream, deferred losses -- the mode bench.py times at batch 4; depth / batch: another point of the progressive
schedule, e.g. depth=0 batch=128 -- real batches stay at the full resolution, as bench.py --sweep feeds them)"""
import cProfile
import os
import pstats
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stylegan.pytorch_amd.GAN import StyleGAN  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    torch.manual_seed(0); random.seed(0)
    opt = dict(learning_rate=0.003, beta_1=0, beta_2=0.99, eps=1e-8)
    sg = StyleGAN("linear", 1024, 3, 512, g_args=dict(latent_size=512, mapping_layers=8, blur_filter=[1, 2, 1], truncation_psi=-1.0, truncation_cutoff=8),
                  d_args=dict(use_wscale=True, blur_filter=[1, 2, 1]), g_opt_args=opt, d_opt_args=opt, loss="logistic", use_ema=True,
                  device=dev, act_dtype=torch.bfloat16)
    if "single" in sys.argv[2:]:
        sg.aux_stream = sg.param_stream = False
        sg.deferred_losses = True
    kv = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
    depth, batch = int(kv.get("depth", 8)), int(kv.get("batch", 4))
    if "depth" in kv or "batch" in kv:
        sg.deferred_losses = True
    z = torch.randn(batch, 512, device=dev)
    x = torch.randn(batch, 1024, 1024, 3, device=dev).permute(0, 3, 1, 2)

    def step():
        sg.optimize_discriminator(z, x, depth, 0.5)
        sg.optimize_generator(z, x, depth, 0.5)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    # the backward passes run on autograd's device thread, which the main thread's profiler does not see: the first
    # backward of a convolution on that thread switches a second profiler on there
    import threading
    from stylegan.pytorch_amd import functional as F
    tl, bw_profs = threading.local(), []
    orig_bw = F.ConvFn.backward

    def conv_backward(ctx, *g):
        if not getattr(tl, "on", False) and threading.current_thread() is not threading.main_thread():
            tl.on = True
            p2 = cProfile.Profile(); p2.enable(); bw_profs.append(p2)
        return orig_bw(ctx, *g)
    F.ConvFn.backward = staticmethod(conv_backward)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(22)
    for p2 in bw_profs:
        print("==== autograd device thread")
        st2 = pstats.Stats(p2)
        st2.sort_stats("tottime").print_stats(40)
        st2.sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
