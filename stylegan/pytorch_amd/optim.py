"""Optimizer step on device: multi-tensor Adam, global-norm clip without a host sync, generator EMA.

Replaces ``torch.optim.Adam`` / ``clip_grad_norm_`` / ``update_average`` of the reference step
(models/GAN.py:529-533, 616-618, 648-656; models/__init__.py:13-40).  ``FusedAdam`` keeps torch.optim.Adam's
``state_dict`` layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter), so optimizer checkpoints interchange.
"""
import math
from collections import OrderedDict

import torch

from . import native as N
from .functional import bump_weight_generation


# Host -> device tables of the multi-tensor kernels.  Pinned staging + non_blocking so that the copy is stream ordered
# and the host does not wait for the GPU queue to drain; pointer tables are cached (parameter / state / gradient
# addresses are stable from step to step).
_TABLES = OrderedDict()             # LRU: key -> device table
_TABLES_MAX = 64
_GRAPH_TABLES = []                  # tables a captured step graph baked the DEVICE ADDRESS of into its kernel arguments: kept for good
_capturing = N.capturing
_to_dev = N.upload


def _dev_i64(vals, device):
    """Cached device copy of an int64 table.  Keys contain gradient addresses, which change in eager mode whenever the
    allocator hands ``zero_grad(set_to_none)``'s successors other blocks, so the cache is a small LRU; an evicted table
    that a kernel in flight still reads stays alive through ``record_stream`` (callers do that), and every table used while
    a step graph is being captured is pinned in ``_GRAPH_TABLES`` (the graph replays with its address)."""
    key = (tuple(vals), str(device))
    got = _TABLES.get(key)
    if got is None:
        got = _TABLES[key] = _to_dev(torch.tensor(vals, dtype=torch.int64), device)[0]
        while len(_TABLES) > _TABLES_MAX:
            _TABLES.popitem(last=False)
    else:
        _TABLES.move_to_end(key)
    if _capturing():
        _GRAPH_TABLES.append(got)
    elif got.is_cuda:
        got.record_stream(torch.cuda.current_stream())       # a consumer on another stream than the one that uploaded it
    return got


def _dev_f32(vals, device):
    """-> (device tensor, pinned staging tensor)."""
    return _to_dev(torch.tensor(vals, dtype=torch.float32), device)


class FusedAdam(torch.optim.Optimizer):
    """Adam (no weight decay, no amsgrad) with one kernel launch per step for all parameters that have a gradient."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        defaults = dict(lr=lr, betas=(float(betas[0]), float(betas[1])), eps=eps, weight_decay=0, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)

    _capture_log = None              # list collecting (pinned scalars, group, active params) while a step graph is captured

    def ensure_state(self):
        """Create the moment buffers of every parameter now (a graph capture must not allocate-and-zero them)."""
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    @staticmethod
    def _bias_scalars(act, lr, b1, b2):
        steps, bc2s = [], []
        for _, st in act:
            t = float(st["step"])
            steps.append(lr / (1.0 - b1 ** t))
            bc2s.append(math.sqrt(1.0 - b2 ** t))
        return steps + bc2s

    @staticmethod
    def graph_advance(entries):
        """Before replaying a captured step: advance the step counts and refresh the bias-correction scalars that the
        graph's H2D copy node reads from pinned memory."""
        for pinned, act, group in entries:
            for _, st in act:
                st["step"] += 1
            b1, b2 = group["betas"]
            pinned.copy_(torch.tensor(FusedAdam._bias_scalars(act, group["lr"], b1, b2), dtype=torch.float32))

    def _active(self, group):
        out = []
        for p in group["params"]:
            if p.grad is None:
                continue
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            out.append((p, st))
        return out

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """``grad_scale``: optional device fp32 scalar multiplied into every gradient (clip coefficient)."""
        assert closure is None
        L = N.lib()
        changed = []
        for group in self.param_groups:
            act = self._active(group)
            if not act:
                continue
            changed.extend(p for p, _ in act)
            b1, b2 = group["betas"]
            lr, eps = group["lr"], group["eps"]
            dev = act[0][0].device
            ptrs, sizes = [[], [], [], []], []
            for p, st in act:
                g = p.grad
                if not (p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32):
                    raise N.SgxError("FusedAdam: parameters and gradients must be contiguous fp32")
                st["step"] += 1
                ptrs[0].append(p.data_ptr()); ptrs[1].append(g.data_ptr())
                ptrs[2].append(st["exp_avg"].data_ptr()); ptrs[3].append(st["exp_avg_sq"].data_ptr())
                sizes.append(p.numel())
            n = len(act)
            table = _dev_i64(ptrs[0] + ptrs[1] + ptrs[2] + ptrs[3] + sizes, dev)
            scal, pinned = _dev_f32(self._bias_scalars(act, lr, b1, b2), dev)
            if self._capture_log is not None and _capturing():
                self._capture_log.append((pinned, act, group))
            base, sb = table.data_ptr(), scal.data_ptr()
            N.check(L.sgx_adam_multi(base, base + 8 * n, base + 16 * n, base + 24 * n, base + 32 * n, n, b1, b2, eps,
                                     sb, sb + 4 * n, None if grad_scale is None else N.ptr(grad_scale), N.stream()),
                    "sgx_adam_multi")
            if not _capturing():
                scal.record_stream(torch.cuda.current_stream())
        bump_weight_generation(changed)              # parameters changed behind torch's version counters
        return None


@torch.no_grad()
def clip_and_step(optim: FusedAdam, max_norm: float):
    """clip_grad_norm_(params, max_norm) followed by optim.step(), with the coefficient kept on the device."""
    L = N.lib()
    grads = [p.grad for g in optim.param_groups for p in g["params"] if p.grad is not None]
    if not grads:
        return
    dev = grads[0].device
    n = len(grads)
    table = _dev_i64([g.data_ptr() for g in grads] + [g.numel() for g in grads], dev)
    partial = torch.empty(n * 32, dtype=torch.float64, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    N.check(L.sgx_gradnorm_clip_coef(table.data_ptr(), table.data_ptr() + 8 * n, n, float(max_norm), N.ptr(partial),
                                     N.ptr(out), N.stream()), "sgx_gradnorm_clip_coef")
    optim.step(grad_scale=out[1:])
    return out


@torch.no_grad()
def ema_update(model_tgt, model_src, beta):
    """tgt = beta*tgt + (1-beta)*src over named_parameters (buffers excluded) -- reference models/__init__.py:31-36."""
    pairs = model_tgt.__dict__.get("_sgx_ema_pairs")              # (source module id, [(target, source)]): walk the trees once
    if pairs is None or pairs[0] != id(model_src):
        src = dict(model_src.named_parameters())
        pairs = model_tgt.__dict__["_sgx_ema_pairs"] = (id(model_src), [(p, src[name]) for name, p in model_tgt.named_parameters()])
    tg, sr, sizes = [], [], []
    dev = None
    changed = []
    for p, q in pairs[1]:
        changed.append(p)
        assert q is not p
        if not p.is_cuda:
            raise N.SgxError("ema_update needs GPU parameters")
        tg.append(p.data_ptr()); sr.append(q.data_ptr()); sizes.append(p.numel()); dev = p.device
        assert p.is_contiguous() and q.is_contiguous()
    n = len(tg)
    table = _dev_i64(tg + sr + sizes, dev)
    base = table.data_ptr()
    N.check(N.lib().sgx_ema_multi(base, base + 8 * n, base + 16 * n, n, float(beta), N.stream()), "sgx_ema_multi")
    bump_weight_generation(changed)
