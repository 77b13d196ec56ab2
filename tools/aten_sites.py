#!/usr/bin/env python3
"""Which lines of the package launch torch's own (ATen) kernels inside one training step: every such launch is one more node
of the step's graphs.  One eager step under torch.profiler with Python stacks; ATen ops that reach a GPU kernel are grouped by the
innermost frame inside stylegan/pytorch_amd (or 'autograd engine' when there is none: gradient accumulation, materialised zeros).

    python tools/aten_sites.py [--batch 4]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    a0 = ap.parse_args()
    dev = torch.device("cuda:0")
    a = argparse.Namespace(dtype="bf16", graphs="off", alpha=0.5)
    cfg = bench.CONFIGS["ffhq1024"]
    sg = bench.make_stylegan(a, cfg, dev, None)
    sg.use_graphs, sg.aux_stream, sg.param_stream = False, False, False
    res, depth, B = cfg["resolution"], cfg["depth"], a0.batch
    real = torch.randn(B, res, res, 3, device=dev).permute(0, 3, 1, 2)
    z = torch.randn(B, 512, device=dev)

    def step():
        sg.optimize_discriminator(z, real, depth, 0.5)
        sg.optimize_generator(z, real, depth, 0.5)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.utils._python_dispatch import TorchDispatchMode
    import traceback
    sites = collections.Counter()
    SKIP = {"aten::view", "aten::_unsafe_view", "aten::reshape", "aten::permute", "aten::transpose", "aten::t", "aten::detach", "aten::alias",
            "aten::slice", "aten::select", "aten::as_strided", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::empty", "aten::empty_like",
            "aten::empty_strided", "aten::unbind", "aten::split", "aten::chunk", "aten::narrow", "aten::view_as", "aten::unflatten",
            "aten::flatten", "aten::_reshape_alias", "aten::lift_fresh", "aten::is_same_size", "aten::stride", "aten::size", "aten::record_stream",
            "aten::new_empty", "aten::new_empty_strided", "aten::contiguous", "aten::resize_", "aten::set_", "aten::is_pinned", "aten::_local_scalar_dense"}

    # engine-level adds (a tensor with two gradient contributions) have no Python frame: tag them with the custom Function whose backward
    # finished last -- the producer of the SECOND contribution -- by wrapping every Function.backward of the package
    from stylegan.pytorch_amd import functional as Fm
    last = {"fn": "?"}
    for nm in dir(Fm):
        cls = getattr(Fm, nm)
        if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and "backward" in cls.__dict__:
            orig = cls.__dict__["backward"].__func__

            def wrapped(ctx, *g, _orig=orig, _nm=nm):
                out = _orig(ctx, *g)
                last["fn"] = _nm
                return out
            cls.backward = staticmethod(wrapped)

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = func.name().split(".")[0]
            if name not in SKIP:
                site = "(autograd engine: gradient accumulation / materialised zeros) after " + last["fn"] + ".backward"
                for fr in reversed(traceback.extract_stack()):
                    if "stylegan/pytorch_amd" in fr.filename:
                        site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                        break
                shape = ""
                for x in list(args) + list((kwargs or {}).values()):
                    if isinstance(x, torch.Tensor):
                        shape = f"{tuple(x.shape)} {str(x.dtype).replace('torch.', '')} {x.device.type}"
                        break
                sites[(name, site, shape)] += 1
            return func(*args, **(kwargs or {}))

    with Log():
        step()
        torch.cuda.synchronize()
    print("ATen ops (views / allocations excluded) in one step:", sum(sites.values()))
    for (name, site, shape), n in sorted(sites.items(), key=lambda kv: (kv[0][1], -kv[1])):
        print(f"{n:4d}  {name:26s} {shape:44s} {site}")


if __name__ == "__main__":
    main()
