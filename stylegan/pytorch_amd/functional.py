"""torch.autograd wrappers around the libsgx_hip.so kernels.

Activations inside this module are contiguous NHWC tensors ``[B, H, W, C]`` (fp32 or bf16); the nn.Modules in
CustomLayers.py / Blocks.py / GAN.py convert at their boundary (a logical-NCHW view with channels_last strides
is the same memory).  Parameters, statistics, RGB images and parameter gradients are fp32 and stay in the
reference's layouts: the kernels read/write them in place (weight scale, 3x3->4x4 synthesis, operand packing and
their adjoints all run in HIP -- no torch arithmetic on parameters).

Differentiation structure: every discriminator op is closed under differentiation -- the
backward of each Function is built from other Functions of this file -- so ``create_graph=True`` (the R1 penalty,
reference models/Losses.py:197-211) works through autograd composition.  Generator-only ops (the fused layer
epilogue, PixelNorm) and parameter gradients are first order.
"""
import contextlib
import functools
import os
import weakref

import numpy as np
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import native as N

# ---------------------------------------------------------------------------------------------------
# "data gradient only" mode: inside torch.autograd.grad(logit, image, create_graph=True) (the R1 penalty) only
# the chain to the image is needed, but a Python Function cannot see which of its input gradients the engine
# wants.  The flag is process-global (backward runs on autograd worker threads; one process drives one GPU).
_DATA_GRAD_ONLY = 0


@contextlib.contextmanager
def data_grad_only():
    global _DATA_GRAD_ONLY
    _DATA_GRAD_ONLY += 1
    try:
        yield
    finally:
        _DATA_GRAD_ONLY -= 1


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _NoGradCtx:
    """Stand-in for the autograd context when nothing needs a gradient (``call``)."""
    needs_input_grad = (False,) * 16
    saved_tensors = ()

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


def call(fn, *args):
    """``fn.apply(*args)`` -- or, when no gradient can flow (``torch.no_grad()``: the D-step generator forward; or no
    input requires one), ``fn.forward`` directly: the kernels are launched without the ~10 us of autograd bookkeeping
    per op that makes the eager step host-bound."""
    if torch.is_grad_enabled():
        for a in args:
            if isinstance(a, torch.Tensor) and a.requires_grad:
                return fn.apply(*args)
    return fn.forward(_NoGradCtx(), *args)


# the Functions' own backward passes go through the same bypass: a first-order backward (grad mode off) launches its
# kernels without building a second autograd graph node per op; under create_graph (the R1 penalty) the inputs require
# grad and ``call`` falls through to ``apply``.  SGX_BWD_BYPASS=0: always ``apply`` (A/B).
_bcall = call if os.environ.get("SGX_BWD_BYPASS", "1") != "0" else (lambda fn, *args: fn.apply(*args))


# ---------------------------------------------------------------------------------------------------
# packed-weight cache.  mode: 'S' plain 3x3 | 'D' fused down | 'U' fused up | 'UF' non-fused-up semantics.
# One sgx_pack_weight launch per (parameter, version) produces both MFMA operand packs in the activation dtype;
# every forward / data-gradient / R1 pass of the step reuses them.  Parameters updated by the HIP optimizer do not
# bump torch's version counter, so a process-wide generation number is bumped instead (optim.py).
MODES = {"S": N.PACK_S, "D": N.PACK_D, "U": N.PACK_U, "UF": N.PACK_UF}
FWD_GEO = {"S": "S", "D": "D", "U": "U", "UF": "U"}
ADJ_GEO = {"S": "S", "D": "U", "U": "D", "UF": "D"}
_PACKS = {}
GRAD_NOTE = None                    # callable(parameter) told of every in-kernel accumulation into .grad (set by dist.py users)
_WEIGHT_GEN = 0
_ACCUM_PARAM_GRADS = False
_PARAM_GRAD_STREAM = None


@contextlib.contextmanager
def param_grad_stream(stream):
    """Inside: convolution weight/bias gradient kernels are issued on ``stream`` (a side stream) instead of the stream the
    backward chain runs on.  Nothing in the backward chain depends on them, so they overlap with the data-gradient
    convolutions of the earlier layers -- which matters exactly where kernels are launch/latency bound (the 4x4..64x64
    layers at batch 4).  The caller joins ``stream`` before it reads the gradients.  Process-global, like the others."""
    global _PARAM_GRAD_STREAM
    prev, _PARAM_GRAD_STREAM = _PARAM_GRAD_STREAM, stream
    try:
        yield
    finally:
        _PARAM_GRAD_STREAM = prev


_FAST_FORK = os.environ.get("SGX_FAST_FORK", "1") != "0"
_STREAMS = {}                     # raw hipStream_t -> torch.cuda.Stream (torch.cuda.current_stream() costs ~10 us of Python)
_set_stream = torch._C._cuda_setStream if hasattr(torch._C, "_cuda_setStream") else None


def _stream_of(raw):
    st = _STREAMS.get(raw)
    if st is None:
        st = _STREAMS[raw] = torch.cuda.current_stream()
        assert st.cuda_stream == raw
    return st


def _param_grads(mode, adjoint, x, gy, weight, scale, want_bias, into):
    """_wgrad_param on the side stream of ``param_grad_stream`` if one is set (with the allocator told about every tensor
    the side stream touches), else inline.  ~70 forks per iteration: the ordering is one library call on raw stream handles
    and torch's current stream is switched by id, not through the ``torch.cuda.stream`` context manager."""
    side = _PARAM_GRAD_STREAM
    if side is None:
        return _wgrad_param(mode, adjoint, x, gy, weight, scale, want_bias, into)
    if not _FAST_FORK:                                      # A/B: the torch-level fork
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            dW, db = _wgrad_param(mode, adjoint, x, gy, weight, scale, want_bias, into)
        for t in (x, gy):
            t.record_stream(side)
        for t in (dW, db):
            if t is not None:
                t.record_stream(cur)
        return dW, db
    cur_raw = N.stream()
    if side.cuda_stream == cur_raw:                         # the "side" stream is the one this backward branch runs on
        return _wgrad_param(mode, adjoint, x, gy, weight, scale, want_bias, into)
    cur = _stream_of(cur_raw)
    N.check(N.lib().sgx_stream_wait_stream(side.cuda_stream, cur_raw), "sgx_stream_wait_stream")   # gy (and x) are complete
    _set_stream(stream_id=side.stream_id, device_index=side.device_index, device_type=side.device_type)
    try:
        dW, db = _wgrad_param(mode, adjoint, x, gy, weight, scale, want_bias, into)
    finally:
        _set_stream(stream_id=cur.stream_id, device_index=cur.device_index, device_type=cur.device_type)
    x.record_stream(side); gy.record_stream(side)
    dW.record_stream(cur)
    if db is not None:
        db.record_stream(cur)
    return dW, db


def _fork_to(side, cur_raw, launch, reads=(), fresh=()):
    """Run ``launch()`` on the stream ``side`` ordered behind everything enqueued on the current stream (raw handle ``cur_raw``), as
    ``_param_grads`` does: ``reads`` = tensors of the current stream that the side stream touches, ``fresh`` = a list ``launch`` fills with
    tensors it allocated (under the side stream) that the current stream will use."""
    if not _FAST_FORK or _set_stream is None:
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            launch()
    else:
        cur = _stream_of(cur_raw)
        N.check(N.lib().sgx_stream_wait_stream(side.cuda_stream, cur_raw), "sgx_stream_wait_stream")
        _set_stream(stream_id=side.stream_id, device_index=side.device_index, device_type=side.device_type)
        try:
            launch()
        finally:
            _set_stream(stream_id=cur.stream_id, device_index=cur.device_index, device_type=cur.device_type)
    for t in reads:
        t.record_stream(side)
    for t in fresh:
        t.record_stream(cur)


@contextlib.contextmanager
def accumulate_param_grads():
    """Inside: the backward of a convolution writes / accumulates the gradients of its LEAF weight and bias straight into
    ``.grad`` (sgx_wgrad*_param(accumulate=...)) and hands autograd nothing for them -- for ``loss.backward()`` of the
    training step, where a discriminator parameter collects up to three contributions.  Process-global: backward runs on
    autograd's worker thread."""
    global _ACCUM_PARAM_GRADS
    prev, _ACCUM_PARAM_GRADS = _ACCUM_PARAM_GRADS, True
    try:
        yield
    finally:
        _ACCUM_PARAM_GRADS = prev


def bump_weight_generation(params=None):
    """Mark parameters as changed behind torch's version counters (the HIP optimizer / EMA kernels write them in
    place).  ``params``: the tensors that changed; None: everything."""
    global _WEIGHT_GEN
    if params is None:
        _WEIGHT_GEN += 1
        return
    for p in params:
        p._sgx_gen = getattr(p, "_sgx_gen", 0) + 1


def _pack_tag(weight):
    return (weight._version, _WEIGHT_GEN, getattr(weight, "_sgx_gen", 0), weight.data_ptr())


def _pack_alloc(weight, sub):
    mode, _, ipad, dtype = sub
    taps = 9 if mode == "S" else 16
    O = weight.shape[0]
    return (torch.empty((taps, O, ipad), dtype=dtype, device=weight.device), torch.empty((taps, ipad, O), dtype=dtype, device=weight.device))


def _pack_mark():
    """(event recorded after the pack kernel, raw handle of the stream it ran on): a consumer on ANOTHER stream (the step
    runs its independent branches on several) waits for the event before it reads the packs."""
    ev = torch.cuda.Event()
    ev.record()
    raw = N.stream()
    return (ev, raw, {raw})                                 # + the streams already ordered after it (one wait each is enough)


def prepack(weights):
    """Re-pack every STALE weight of ``weights`` (for all the (mode, scale, ipad, dtype) combinations it has been used
    with) in one sgx_pack_weight_multi launch -- instead of one small launch per layer on first use after the optimizer
    step.  Weights never used yet are left to the lazy path of ``packs``."""
    rows, blk, dtype, touched = [], 0, None, []
    for w in weights:
        ent = _PACKS.get(id(w))
        if ent is None or ent[0]() is not w or not ent[3]:
            continue
        tag = _pack_tag(w)
        if ent[1] == tag:
            continue
        if w.dtype != torch.float32 or not w.is_contiguous():
            raise N.SgxError("parameters must be contiguous fp32")
        ent[1], ent[2] = tag, {}
        touched.append(ent)
        for sub in ent[3]:
            if dtype is None:
                dtype = sub[3]
            if sub[3] != dtype:                                   # a second activation dtype: leave it to the lazy path
                continue
            fwd, adj = _pack_alloc(w, sub)
            ent[2][sub] = (fwd, adj)
            O, I = w.shape[0], w.shape[1]
            nb = ((O + 31) // 32) * ((sub[2] + 31) // 32)        # == sgx_pack_weight_blocks(O, Ipad)
            rows.append([w.data_ptr(), fwd.data_ptr(), adj.data_ptr(), O, I, sub[2], MODES[sub[0]],
                         int(np.float32(sub[1]).view(np.uint32)), blk, nb])
            blk += nb
    if rows:
        table, _ = N.upload(torch.tensor(rows, dtype=torch.int64), weights[0].device)
        N.check(N.lib().sgx_pack_weight_multi(N.ptr(table), len(rows), blk, N.F32 if dtype == torch.float32 else N.BF16, N.stream()),
                "sgx_pack_weight_multi")
        if not N.capturing():
            table.record_stream(torch.cuda.current_stream())
        mark = _pack_mark()
        for ent in touched:
            ent[4] = mark


def packs(weight, mode, scale, ipad, dtype):
    """(fwd, adj) operand packs of a [O][I][3][3] fp32 parameter; cached per (version, generation).  Entry layout:
    [weakref, tag, {sub: (fwd, adj)} valid for the tag, {sub} ever requested (what ``prepack`` rebuilds)]."""
    key = id(weight)
    ent = _PACKS.get(key)
    tag = _pack_tag(weight)
    if ent is None or ent[0]() is not weight:
        ent = [weakref.ref(weight, lambda _r, k=key: _PACKS.pop(k, None)), tag, {}, set(), None]
        _PACKS[key] = ent
    elif ent[1] != tag:
        ent[1], ent[2] = tag, {}
    sub = (mode, float(scale), int(ipad), dtype)
    ent[3].add(sub)
    got = ent[2].get(sub)
    if got is not None and ent[4] is not None:
        raw = N.stream()
        if raw not in ent[4][2] or (not _FAST_FORK and raw != ent[4][1]):   # packed on another stream: wait once per consumer stream
            _stream_of(raw).wait_event(ent[4][0])
            ent[4][2].add(raw)
    if got is None:
        w = _c(weight.detach())
        if w.dtype != torch.float32:
            raise N.SgxError("parameters must be fp32")
        O, I = w.shape[0], w.shape[1]
        fwd, adj = _pack_alloc(w, sub)
        N.check(N.lib().sgx_pack_weight(N.ptr(w), N.ptr(fwd), N.ptr(adj), O, I, ipad, MODES[mode], float(scale),
                                        N.F32 if dtype == torch.float32 else N.BF16, N.stream()), "sgx_pack_weight")
        got = (fwd, adj)
        ent[2][sub] = got
        ent[4] = _pack_mark()
    return got


def clear_pack_cache():
    """Forget every pack (the usage records that ``prepack`` batches by are kept)."""
    for ent in _PACKS.values():
        ent[1], ent[2] = None, {}
    _UPBLUR_PACKS.clear()


# Composite packs of "transposed convolution, then blur" (sgx_conv_upblur): the 9 composite taps to the 4 output parity classes + the
# 22 border-correction tiles, bf16 [9][4N][K] + [22][2N][K], composed by sgx_pack_upblur straight from the fp32 parameter (the 16 transposed-convolution
# taps synthesised as sgx_pack_weight does, the blur's 1/16 in the scale -- one rounding to bf16, of the composite).  Cached per (parameter, use) under the parameter's pack tag, with the event other consumer streams wait for.
_UPBLUR_PACKS = {}


def upblur_pack(weight, mode, scale, ipad, adjoint):
    key = (id(weight), mode, float(scale), int(ipad), bool(adjoint))
    tag = _pack_tag(weight)
    ent = _UPBLUR_PACKS.get(key)
    if ent is not None and ent[0]() is weight and ent[1] == tag:
        raw = N.stream()
        if raw not in ent[3][2] or (not _FAST_FORK and raw != ent[3][1]):
            _stream_of(raw).wait_event(ent[3][0])
            ent[3][2].add(raw)
        return ent[2]
    w = _c(weight.detach())
    if w.dtype != torch.float32 or int(ipad) != w.shape[1]:
        raise N.SgxError("conv+blur composite pack: contiguous fp32 parameter without channel padding expected")
    O, I = w.shape[0], w.shape[1]
    Nn, K = (I, O) if adjoint else (O, I)
    wc = torch.empty(((9 * 4 + 22 * 2) * Nn * K,), dtype=torch.bfloat16, device=w.device)      # [9][4N][K] composite taps + [22][2N][K] correction tiles
    N.check(N.lib().sgx_pack_upblur(N.ptr(w), N.ptr(wc), O, I, MODES[mode], int(bool(adjoint)), float(scale) / 16.0, N.stream()), "sgx_pack_upblur")
    _UPBLUR_PACKS[key] = [weakref.ref(weight, lambda _r, k=key: _UPBLUR_PACKS.pop(k, None)), tag, wc, _pack_mark()]
    return wc


# ---------------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def _splitk_ws_bytes(gi, B, H, W, Cin, Cout, dt):
    """``sgx_conv_splitk_ws_bytes`` per shape, asked of the library once (the plan is a pure function of the shape and process-wide switches;
    a ctypes call per convolution launch would be host time on the host-bound low-resolution depths)."""
    return int(N.lib().sgx_conv_splitk_ws_bytes(gi, B, H, W, Cin, Cout, dt))


def _conv_launch(geo, x, wq, bias, act, mask=None):
    B, H, W, Cin = x.shape
    taps, Cout, K = wq.shape
    if K != Cin:
        raise N.SgxError(f"conv: weight pack expects {K} input channels, activation has {Cin}")
    L = N.lib()
    gi = "SDU".index(geo)
    wsb = _splitk_ws_bytes(gi, B, H, W, Cin, Cout, N.dt(x))
    if wsb:
        # round 6: the launches that would leave most of the chip idle split their reduction over blocks (fp32 partials + a finishing launch)
        oh, ow = (H, W) if geo == "S" else ((H // 2, W // 2) if geo == "D" else (2 * H, 2 * W))
        y = torch.empty((B, oh, ow, Cout), dtype=x.dtype, device=x.device)
        if mask is not None and (geo != "S" or mask.shape != y.shape or mask.dtype != y.dtype):
            raise N.SgxError("conv: the output mask must have the output's shape and dtype (3x3 geometry)")
        ws = N.workspace(wsb, x.device)
        N.check(L.sgx_conv_splitk(gi, N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), N.ptr(mask), B, H, W, Cin, Cout, act, N.dt(x), N.ptr(ws), wsb, N.stream()),
                "sgx_conv_splitk")
        return y
    if geo == "S":
        y = torch.empty((B, H, W, Cout), dtype=x.dtype, device=x.device)
        if mask is not None and (mask.shape != y.shape or mask.dtype != y.dtype):
            raise N.SgxError("conv: the output mask must have the output's shape and dtype")
        N.check(L.sgx_conv3x3(N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), B, H, W, Cin, Cout, act, N.ptr(mask), N.dt(x), N.stream()), "sgx_conv3x3")
    elif geo == "D":
        assert mask is None
        y = torch.empty((B, H // 2, W // 2, Cout), dtype=x.dtype, device=x.device)
        N.check(L.sgx_conv4x4s2_down(N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), B, H, W, Cin, Cout, act, N.dt(x), N.stream()), "sgx_conv4x4s2_down")
    else:
        assert bias is None and act == 0 and mask is None
        y = torch.empty((B, 2 * H, 2 * W, Cout), dtype=x.dtype, device=x.device)
        N.check(L.sgx_conv4x4s2_up(N.ptr(x), N.ptr(wq), N.ptr(y), B, H, W, Cin, Cout, N.dt(x), N.stream()), "sgx_conv4x4s2_up")
    return y


def _wgrad_param(mode, adjoint, x, gy, weight, scale, want_bias=False, into=None):
    """Gradient w.r.t. the [O][I][3][3] parameter of y = conv(x) (or of the layer's data-gradient conv if adjoint);
    with ``want_bias`` also the bias gradient sum(gy) over batch and pixels, out of the same pass -> (dW, db|None).
    ``into`` = (dW tensor or None, db tensor or None): accumulate into these existing gradients instead."""
    L = N.lib()
    O, I = weight.shape[0], weight.shape[1]
    acc_w = into is not None and into[0] is not None
    acc_b = want_bias and into is not None and into[1] is not None
    dW = into[0] if acc_w else torch.empty((O, I, 3, 3), dtype=torch.float32, device=x.device)
    db = (into[1] if acc_b else torch.empty((O,), dtype=torch.float32, device=x.device)) if want_bias else None
    acc = (1 if acc_w else 0) | (2 if acc_b else 0)
    B = x.shape[0]
    if mode == "S":
        _, H, W, Cx = x.shape
        Cdy = gy.shape[3]
        wsb = L.sgx_wgrad_ws_bytes(9, B, H, W, Cx, Cdy)
        ws = N.workspace(wsb, x.device)
        N.check(L.sgx_wgrad3x3_param(N.ptr(x), N.ptr(gy), N.ptr(dW), N.ptr(db), N.ptr(ws), wsb, B, H, W, Cx, Cdy, int(adjoint),
                                     float(scale), O, I, acc, N.dt(x), N.stream()), "sgx_wgrad3x3_param")
        return dW, db
    launched = ADJ_GEO[mode] if adjoint else FWD_GEO[mode]          # geometry of the convolution that ran
    fine, coarse = (x, gy) if launched == "D" else (gy, x)
    _, H, W, Cf = fine.shape
    Cc = coarse.shape[3]
    wsb = L.sgx_wgrad_ws_bytes(16, B, H, W, Cf, Cc)
    ws = N.workspace(wsb, x.device)
    N.check(L.sgx_wgrad4x4s2_param(N.ptr(fine), N.ptr(coarse), N.ptr(dW), N.ptr(db), N.ptr(ws), wsb, B, H, W, Cf, Cc, MODES[mode],
                                   float(scale), O, I, acc, N.dt(x), N.stream()), "sgx_wgrad4x4s2_param")
    return dW, db


class ConvFn(Function):
    """y = act(conv(x, pack(weight)) + bias) [* slope(mask)] for one EqualizedConv2d parameter.

    ``adjoint=False``: the layer's own convolution (geometry by ``mode``).  ``adjoint=True``: its data-gradient
    convolution (transposed/flipped pack).  The backward of either is the other, so the op is closed under
    differentiation; the parameter gradient comes back in the parameter's own layout.

    Activation backward without a pass of its own (discriminator chain conv1_down+LeakyReLU -> next block's conv0):
    ``defer_act``: this layer's LeakyReLU backward is applied by whoever consumes its output -- the incoming gradient is
    already masked; ``x_masked``: x is such an output, so the data gradient leaves the kernel multiplied by slope(x)
    (``mask`` argument of the adjoint launch, fused in its store; bit-identical to the separate pass)."""

    @staticmethod
    def forward(ctx, x, weight, bias, mode, scale, ipad, adjoint, act, mask=None, defer_act=False, x_masked=False, stats=None,
                x_pre=None, bits_out=False, x_pre_bits=None):
        """``stats`` = (epilogue bias or None, noise [B,1,H,W], noise weight) of the generator LayerEpilogue that consumes y: the
        store of the convolution also emits the epilogue's partial instance-norm statistics -> returns (y, partials); only where
        ``conv_stats_nparts`` says the shape has such a kernel (plain 3x3, bf16).
        ``x_pre`` (stride-2 layers of the discriminator): x = blur(lrelu(x_pre)), made by ``ActBlurPassFn`` whose own backward is
        the identity -- THIS op's backward then returns the gradient w.r.t. x_pre, blur(conv_adjoint(gy)) * slope(x_pre), from
        one kernel where the shape has it (``ConvBlurFn``), else from the adjoint convolution and the blur-and-mask pass
        (``x_pre_bits``: the sign bits of x_pre, which that pass then reads instead of x_pre).
        ``bits_out`` (plain 3x3, bf16, where ``conv_signbits_ok``): also return the sign bits of y, [B,H,W,C/8] uint8."""
        x = _c(x)
        fwd, adj = packs(weight, mode, scale, ipad, x.dtype)
        geo = ADJ_GEO[mode] if adjoint else FWD_GEO[mode]
        ctx.cfg = (mode, scale, ipad, adjoint, act, bias is not None, bool(defer_act), bool(x_masked))
        ctx.bias_ref = weakref.ref(bias) if bias is not None else (lambda: None)
        # (two outputs: without this the engine hands backward a freshly zero-filled "gradient" of the non-differentiable one --
        # a fill kernel per call, 20 a step)
        ctx.set_materialize_grads(False)
        # saved-tensor layout shared with ConvBlurFn / ConvDownFadeFn (their backward ends in ConvFn.backward on the same ctx):
        # (x, weight, y | None, mask | None, x_pre | None, x_pre_bits | None, ...).  x_pre / x_pre_bits are SAVED, not plain ctx
        # attributes: an in-place write to them between forward and backward is then caught by autograd's version check
        if bits_out:
            assert geo == "S" and not adjoint and stats is None
            y, bits = _conv_bits_launch(x, fwd, None if bias is None else _c(bias.detach()), act, mask)
            ctx.mark_non_differentiable(bits)
            ctx.save_for_backward(x, weight, y if (act and not defer_act) else None, mask, x_pre, x_pre_bits)
            return y, bits
        if stats is not None:
            assert geo == "S" and not adjoint and bias is None and act == 0 and mask is None
            y, part = _conv_stats_launch(x, fwd, *stats)
            ctx.mark_non_differentiable(part)
            ctx.save_for_backward(x, weight, None, None, x_pre, x_pre_bits)
            return y, part
        y = _conv_launch(geo, x, adj if adjoint else fwd, None if bias is None else _c(bias.detach()), act, mask)
        ctx.save_for_backward(x, weight, y if (act and not defer_act) else None, mask, x_pre, x_pre_bits)
        return y

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        if gy is None:                                        # (only a non-differentiable output was "used")
            return (None,) * 15
        x, weight, y, mask, x_pre, xb = ctx.saved_tensors[:6]
        mode, scale, ipad, adjoint, act, has_bias, defer_act, x_masked = ctx.cfg
        gy = _c(gy)
        if mask is not None:                                  # adjoint of the output mask (reached by the R1 double backward only)
            gy = _bcall(LReluBwdFn, gy, mask, 0.2, 1.0)
        if act and not defer_act:
            gy = _bcall(LReluBwdFn, gy, y, 0.2, 1.0)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if x_pre is not None or xb is not None:
                assert not x_masked
                # (x_pre None: the producer never materialised the pre-activation -- RgbConvBlurFn -- and only its sign bits exist)
                # (the fused kernel reads the mask TENSOR: with the mask as sign bits its blur epilogue, already the limit of that
                # kernel, gets another ~16 VALU operations per element -- measured 990 us against 342 + 553 for the two passes at
                # batch 32, 1024^2, profiles/r04_rgbconv_probe_history.txt -- so layers that only have the bits take the two passes)
                cout_b = weight.shape[1] if not adjoint else weight.shape[0]
                if xb is not None and conv_upblur_ok(gy, cout_b, mode, not adjoint):
                    gx = _bcall(ConvBlurFn, gy, weight, mode, scale, ipad, not adjoint, None, xb)       # one kernel, the mask from its sign bits
                elif x_pre is not None and conv_blur_ok(gy, cout_b, mode, not adjoint, has_mask_tensor=True):
                    gx = _bcall(ConvBlurFn, gy, weight, mode, scale, ipad, not adjoint, x_pre)      # one kernel
                else:
                    gx = _bcall(BlurMaskFn, _bcall(ConvFn, gy, weight, None, mode, scale, ipad, not adjoint, 0), x_pre, xb)
            else:
                gx = _bcall(ConvFn, gy, weight, None, mode, scale, ipad, not adjoint, 0, x if x_masked else None, False, False)
        if not _DATA_GRAD_ONLY:
            want_b = has_bias and ctx.needs_input_grad[2]
            # the bias gradient rides along in the weight-gradient pass when gy is its O-channel side
            fuse_b = want_b and ctx.needs_input_grad[1] and not adjoint and mode in ("S", "D")
            if ctx.needs_input_grad[1] and _ACCUM_PARAM_GRADS and weight.is_leaf and not torch.is_grad_enabled():
                # training step: accumulate straight into .grad in the finishing kernel (no add kernel per contribution)
                bias = ctx.bias_ref() if fuse_b else None
                if fuse_b and (bias is None or not bias.is_leaf):
                    fuse_b, bias = False, None
                dW, db = _param_grads(mode, adjoint, _c(x), gy, weight, scale, fuse_b, (weight.grad, bias.grad if fuse_b else None))
                if weight.grad is None:
                    weight.grad = dW
                if fuse_b:
                    if bias.grad is None:
                        bias.grad = db
                    want_b = False
                if GRAD_NOTE is not None:                      # data parallel: bucket-level all-reduce overlap (dist.BucketScheduler)
                    GRAD_NOTE(weight)
                    if fuse_b:
                        GRAD_NOTE(bias)
            elif ctx.needs_input_grad[1]:
                gw, gb = _bcall(WgradFn, x, gy, weight, mode, scale, adjoint, fuse_b)
            if want_b and not fuse_b:
                gb = _bcall(ColSumFn, gy, 1.0)
        return gx, gw, gb, None, None, None, None, None, None, None, None, None, None, None, None


# Where the transposed convolution and the blur after it run as ONE kernel.  "auto": where it was measured to win alone on the
# GPU (tools/upblur_probe.py, profiles/r03_upblur_probe.txt) -- the 16-channel 1024x1024 output at a large batch: 872 -> 740 us
# (plain blur) and 1026 -> 846 us (blur * activation mask) at batch 32; at 512x512 (64 -> 32) it is a wash (430 vs 450 / 507 vs
# 482 us), below that and at batch 4 a loss: the blur epilogue is ~1000 VALU instructions per wave and tile, executed by all
# eight waves of the CU's single block at once, where the separate blur pass streams at 4.6-5.2 TB/s.  "all": every shape that
# has the kernel (the parity tests); "off": never (process-wide A/B without a rebuild: SGX_CONV_UP_BLUR=0).
CONV_BLUR_POLICY = os.environ.get("SGX_CONV_UP_BLUR_POLICY", "auto")
CONV_BLUR_MIN_PIXELS = 1 << 22          # coarse input pixels B*H*W from which "auto" fuses the 16-channel layer


def conv_upblur_ok(x, cout, mode, adjoint):
    """True if the round-5 kernel (``sgx_conv_upblur``: blur o transposed convolution as ONE 3x3 convolution to the four output parity
    classes, depth-to-space store, mask from sign bits) takes the convolution (mode, adjoint) applied to NHWC ``x``: 32 -> 16 channels,
    bf16 -- the 1024x1024 level of both networks, at any batch."""
    geo = ADJ_GEO[mode] if adjoint else FWD_GEO[mode]
    if CONV_BLUR_POLICY == "off" or not CONV_UPBLUR or geo != "U" or x.dtype != torch.bfloat16:
        return False
    B, H, W, Cin = x.shape
    return bool(N.lib().sgx_conv_upblur_ok(B, H, W, Cin, int(cout), N.BF16))


CONV_UPBLUR = os.environ.get("SGX_CONV_UPBLUR", "1") != "0"     # A/B: 0 = the round-3 kernel (blur in the store epilogue) / the two passes


def conv_blur_ok(x, cout, mode, adjoint, has_mask_tensor=False):
    """True if ``ConvBlurFn`` has a kernel for the convolution (mode, adjoint) applied to NHWC ``x`` with ``cout`` outputs AND the
    policy wants it used.  ``has_mask_tensor``: the caller will pass the mask as a TENSOR (``z``): the round-5 composite kernel takes
    no mask or sign bits only, so that call is served by the round-3 kernel under its own policy and shape check."""
    geo = ADJ_GEO[mode] if adjoint else FWD_GEO[mode]
    if CONV_BLUR_POLICY == "off" or geo != "U" or x.dtype != torch.bfloat16:
        return False
    if not has_mask_tensor and conv_upblur_ok(x, cout, mode, adjoint):
        return True
    B, H, W, Cin = x.shape
    if CONV_BLUR_POLICY == "auto" and not (int(cout) == 16 and B * H * W >= CONV_BLUR_MIN_PIXELS):
        return False
    return bool(N.lib().sgx_conv4x4s2_up_blur_ok(B, H, W, Cin, int(cout), N.BF16))


class ConvBlurFn(Function):
    """blur3x3(conv(x)) [* slope(z)] for a transposed (4x4 stride-2 up) convolution in ONE kernel: the blur is applied to the
    accumulators in the store epilogue (sgx_conv4x4s2_up_blur).  (mode, adjoint) as in ``ConvFn``: the layer's own up-convolution
    (generator conv0_up -> blur) or the data gradient of a stride-2 layer (discriminator: ``z`` = the pre-activation whose
    LeakyReLU and blur precede that layer; the result is the gradient w.r.t. z).  Backward: the blur (and mask) adjoint as their
    own pass, then exactly ``ConvFn``'s backward -- so the op composes under ``create_graph`` like the separate ops did."""

    @staticmethod
    def forward(ctx, x, weight, mode, scale, ipad, adjoint, z=None, zbits=None):
        """``zbits``: the sign bits of z ([B,2H,2W,C/8] uint8) -- read instead of z where the producer wrote them (1 bit per
        element instead of 16; z itself may then be None: functional.RgbConvBlurFn never materialises it)."""
        x = _c(x)
        geo = ADJ_GEO[mode] if adjoint else FWD_GEO[mode]
        assert geo == "U"
        B, H, W, Cin = x.shape
        cout_ = weight.shape[1] if adjoint else weight.shape[0]
        if z is None and conv_upblur_ok(x, cout_, mode, adjoint):
            # round 5: ONE 3x3 convolution to the four output parity classes (composite weights), depth-to-space store, mask from bits
            if zbits is not None and (tuple(zbits.shape) != (B, 2 * H, 2 * W, cout_ // 8) or zbits.dtype != torch.uint8):
                raise N.SgxError("conv+blur: sign bits [B, 2H, 2W, Cout/8] uint8 expected")
            wc = upblur_pack(weight, mode, scale, ipad, adjoint)
            y = torch.empty((B, 2 * H, 2 * W, cout_), dtype=x.dtype, device=x.device)
            N.check(N.lib().sgx_conv_upblur(N.ptr(x), N.ptr(wc), N.ptr(y), N.ptr(None if zbits is None else _c(zbits)), B, H, W, Cin, cout_, N.dt(x),
                                            N.stream()), "sgx_conv_upblur")
            ctx.cfg = (mode, scale, ipad, adjoint, 0, False, False, False)
            ctx.bias_ref = lambda: None
            ctx.save_for_backward(x, weight, None, None, None, None, None, zbits)
            return y
        fwd, adj = packs(weight, mode, scale / 16.0, ipad, x.dtype)    # the blur's 1/16 rides in the weight scale (a power of two: exact)
        wq = adj if adjoint else fwd
        taps, Cout, K = wq.shape
        if K != Cin:
            raise N.SgxError(f"conv+blur: weight pack expects {K} input channels, activation has {Cin}")
        y = torch.empty((B, 2 * H, 2 * W, Cout), dtype=x.dtype, device=x.device)
        if z is not None and (z.shape != y.shape or z.dtype != y.dtype):
            raise N.SgxError("conv+blur: the mask must have the output's shape and dtype")
        if zbits is not None:
            if tuple(zbits.shape) != (B, 2 * H, 2 * W, Cout // 8) or zbits.dtype != torch.uint8:
                raise N.SgxError("conv+blur: sign bits [B, 2H, 2W, Cout/8] uint8 expected")
            N.check(N.lib().sgx_conv4x4s2_up_blur_bits(N.ptr(x), N.ptr(wq), N.ptr(y), N.ptr(_c(zbits)), B, H, W, Cin, Cout, N.dt(x), N.stream()),
                    "sgx_conv4x4s2_up_blur_bits")
        else:
            N.check(N.lib().sgx_conv4x4s2_up_blur(N.ptr(x), N.ptr(wq), N.ptr(y), N.ptr(None if z is None else _c(z)), B, H, W, Cin, Cout,
                                                  N.dt(x), N.stream()), "sgx_conv4x4s2_up_blur")
        ctx.cfg = (mode, scale, ipad, adjoint, 0, False, False, False)
        ctx.bias_ref = lambda: None
        ctx.save_for_backward(x, weight, None, None, None, None, z, zbits)       # (ConvFn's layout + the mask and its sign bits)
        return y

    @staticmethod
    def backward(ctx, gg):
        gg = _c(gg)
        z, zbits = ctx.saved_tensors[6:8]
        masked = z is not None or zbits is not None
        m = _bcall(MaskBlurFn, gg, z, zbits) if masked else _bcall(BlurFn, gg)
        out = ConvFn.backward(ctx, m)                      # (gx, gw, gb, ...): same saved tensors / cfg layout
        return out[0], out[1], None, None, None, None, None, None


class ActBlurPassFn(Function):
    """blur(lrelu(z)) whose backward is the IDENTITY: its only consumer is a ``ConvFn`` called with ``x_pre=z``, whose backward
    already returns the gradient w.r.t. z (the blur and the activation's mask folded into the data-gradient kernel)."""

    @staticmethod
    def forward(ctx, z):
        return _blur_act(_c(z), None, 1)

    @staticmethod
    def backward(ctx, g):
        return g


def act_blur_pass(z):
    """blur(lrelu(z)) for a consumer that is called with ``x_pre=z`` (``ActBlurPassFn``: identity backward); the output is tagged with
    its source so that ``conv`` / ``ConvDownFadeFn`` callers cannot pair it with another tensor."""
    x = call(ActBlurPassFn, z)
    x._sgx_pre_of = z
    return x


class WgradFn(Function):
    """Parameter gradient(s) of ConvFn (first order only: nothing in the training step differentiates through it)."""

    @staticmethod
    def forward(ctx, x, gy, weight, mode, scale, adjoint, want_bias):
        return _wgrad_param(mode, adjoint, _c(x), _c(gy), weight, scale, want_bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, g, gb):
        raise NotImplementedError("second derivative through a weight gradient is not part of the training path")


def conv(x, weight, bias, mode, scale, act=N.ACT_NONE, ipad=None, defer_act=False, x_masked=False, stats=None, x_pre=None,
         bits_out=False, x_pre_bits=None):
    if x_pre is not None and getattr(x, "_sgx_pre_of", None) is not x_pre:
        # x_pre makes this op's backward return the gradient w.r.t. x_pre (blur and mask folded in): only right if x really is the
        # pass-through blur of x_pre, whose own backward is the identity
        raise N.SgxError("conv: x_pre given, but x is not the ActBlurPassFn output of that tensor (functional.act_blur_pass)")
    return call(ConvFn, x, weight, bias, mode, float(scale), int(ipad if ipad is not None else weight.shape[1]), False, act, None,
                bool(defer_act), bool(x_masked), stats, x_pre, bool(bits_out), x_pre_bits)


SIGNBITS_ON = True                  # tests flip this to compare against masks read from the pre-activation itself


def conv_signbits_ok(x, cout):
    """True if the plain 3x3 convolution of NHWC ``x`` to ``cout`` channels can also write the sign bits of its output."""
    if not SIGNBITS_ON or x.dtype != torch.bfloat16:
        return False
    B, H, W, Cin = x.shape
    return bool(N.lib().sgx_conv3x3_signbits_ok(B, H, W, Cin, int(cout), N.BF16))


def _conv_bits_launch(x, wq, bias, act, mask):
    B, H, W, Cin = x.shape
    taps, Cout, K = wq.shape
    if K != Cin or taps != 9:
        raise N.SgxError("conv+bits: 3x3 pack with the activation's channel count expected")
    y = torch.empty((B, H, W, Cout), dtype=x.dtype, device=x.device)
    bits = torch.empty((B, H, W, Cout // 8), dtype=torch.uint8, device=x.device)
    N.check(N.lib().sgx_conv3x3_signbits(N.ptr(x), N.ptr(wq), N.ptr(bias), N.ptr(y), N.ptr(bits), B, H, W, Cin, Cout, act, N.ptr(mask),
                                         N.dt(x), N.stream()), "sgx_conv3x3_signbits")
    return y, bits


def conv_stats_nparts(x, cout):
    """Tiles per image of the 3x3 convolution kernel that also emits the following epilogue's statistics for NHWC input ``x``
    and ``cout`` output channels; 0 = no such kernel for this shape / dtype."""
    if x.dtype != torch.bfloat16:
        return 0
    B, H, W, Cin = x.shape
    return N.lib().sgx_conv3x3_stats_nparts(B, H, W, Cin, int(cout), N.BF16)


def _conv_stats_launch(x, wq, ebias, noise, nw):
    B, H, W, Cin = x.shape
    taps, Cout, K = wq.shape
    if K != Cin or taps != 9:
        raise N.SgxError("conv+stats: 3x3 pack with the activation's channel count expected")
    L = N.lib()
    npart = L.sgx_conv3x3_stats_nparts(B, H, W, Cin, Cout, N.dt(x))
    if npart <= 0:
        raise N.SgxError("conv+stats: no fused kernel for this shape (ask conv_stats_nparts first)")
    y = torch.empty((B, H, W, Cout), dtype=x.dtype, device=x.device)
    part = torch.empty((B, npart, Cout, 2), dtype=torch.float64, device=x.device)
    noise_c = _c(noise.detach().reshape(B, H * W))
    if noise_c.dtype != torch.float32:
        noise_c = noise_c.float()
    N.check(L.sgx_conv3x3_stats(N.ptr(x), N.ptr(wq), N.ptr(y), N.ptr(None if ebias is None else _c(ebias.detach())), N.ptr(noise_c),
                                N.ptr(_c(nw.detach())), N.ptr(part), part.numel() * 8, B, H, W, Cin, Cout, N.dt(x), N.stream()),
            "sgx_conv3x3_stats")
    return y, part


# ---------------------------------------------------------------------------------------------------
# The discriminator's first layer pair at the current resolution -- from_rgb (1x1, no activation) -> conv0 (3x3) -> LeakyReLU ->
# blur, reference models/GAN.py:353,413-427 + models/Blocks.py:137-142 -- as ONE 3-channel convolution of the RGB image
# (csrc/rgbconv.hip; the algebra is in its header).  Three Functions, closed under differentiation like the ConvFn family:
#   RgbConvBlurFn : img -> (xb = blur(lrelu(conv(img) + b0)), sign bits of the pre-activation); its "gradient" input is the gradient
#                   w.r.t. the PRE-ACTIVATION z (its only consumer is the stride-2 ConvFn called with x_pre_bits, whose backward
#                   applies the blur and the activation mask -- the ActBlurPassFn convention)
#   RgbConvAdjFn  : gz -> gradient w.r.t. the image;  RgbConvPlainFn: img -> conv(img), the adjoint's own adjoint (R1 double backward)
# Parameters stay the reference's four tensors (conv0.weight/bias, from_rgb.weight/bias): their gradients come from the composed
# weight gradient by the chain rule inside sgx_rgbconv_wgrad, in the parameters' own layouts.
_RGB_PACKS = {}


RGBCONV = os.environ.get("SGX_RGBCONV", "1") != "0"        # A/B (tests flip the attribute; the library reads the same variable)


def rgbconv_ok(B, H, W, C, dtype):
    return RGBCONV and dtype == torch.bfloat16 and bool(N.lib().sgx_rgbconv_ok(int(B), int(H), int(W), int(C), N.BF16))


def rgb_packs(w0, s0, wr, sr, br):
    """(wf, wd): operand packs of the composed convolution, cached per version of the three parameters they are made of."""
    key = (id(w0), id(wr))
    tag = (_pack_tag(w0), _pack_tag(wr), None if br is None else _pack_tag(br), float(s0), float(sr))
    ent = _RGB_PACKS.get(key)
    if ent is None or ent[0]() is not w0 or ent[1]() is not wr:
        ent = [weakref.ref(w0, lambda _r, k=key: _RGB_PACKS.pop(k, None)), weakref.ref(wr), None, None, None]
        _RGB_PACKS[key] = ent
    if ent[2] != tag:
        C = w0.shape[0]
        if w0.dtype != torch.float32 or wr.dtype != torch.float32 or tuple(w0.shape) != (C, C, 3, 3) or tuple(wr.shape) != (C, 3, 1, 1):
            raise N.SgxError("rgb_packs: conv0.weight [C,C,3,3] and from_rgb.weight [C,3,1,1] (fp32) expected")
        wf = torch.empty((3, C, 16), dtype=torch.bfloat16, device=w0.device)
        wd = torch.empty((3, 16, C), dtype=torch.bfloat16, device=w0.device)
        N.check(N.lib().sgx_rgbconv_pack(N.ptr(_c(w0.detach())), float(s0), N.ptr(_c(wr.detach())), float(sr),
                                         N.ptr(None if br is None else _c(br.detach())), 1.0, N.ptr(wf), N.ptr(wd), C, N.stream()), "sgx_rgbconv_pack")
        ent[2], ent[3], ent[4] = tag, (wf, wd), _pack_mark()
    else:
        raw = N.stream()
        if raw not in ent[4][2]:                               # packed on another stream: wait once per consumer stream
            _stream_of(raw).wait_event(ent[4][0])
            ent[4][2].add(raw)
    return ent[3]


def _rgb_wgrad(img, gz, ones, w0, b0, wr, br, s0, sr, want):
    """Gradients of (conv0.weight, conv0.bias, from_rgb.weight, from_rgb.bias) -- ``want`` says which -- from the image (or, for
    the adjoint op, the gradient that flowed into it) and the pre-activation gradient gz.  Training step: accumulated straight into
    ``.grad`` on the weight-gradient side stream, nothing handed to autograd (-> four Nones); otherwise -> the four tensors."""
    params = (w0, b0, wr, br)
    want = [bool(w) and p is not None for w, p in zip(want, params)]
    if not ones:
        want[1] = want[3] = False                              # the adjoint / plain ops do not involve the biases
    if not any(want):
        return None, None, None, None
    accum = _ACCUM_PARAM_GRADS and not torch.is_grad_enabled() and all(p.is_leaf for p, w in zip(params, want) if w)
    outs, acc, fresh = [None] * 4, 0, []

    def launch():
        # new gradient tensors are allocated HERE, i.e. under the stream that writes them (the side stream when forked, as in
        # _param_grads), and the consumer stream is recorded afterwards -- not allocated under the current stream and recorded on it
        nonlocal acc
        for k, (p, w) in enumerate(zip(params, want)):
            if not w:
                continue
            if accum and p.grad is not None:
                outs[k] = p.grad; acc |= 1 << k
            else:
                outs[k] = torch.empty(p.shape, dtype=torch.float32, device=gz.device); fresh.append(outs[k])
        L = N.lib()
        B, H, W, C = gz.shape
        wsb = L.sgx_rgbconv_wgrad_ws_bytes(B, H, W, C)
        ws = N.workspace(wsb, gz.device)
        N.check(L.sgx_rgbconv_wgrad(N.ptr(img), N.ptr(gz), int(bool(ones)), N.ptr(_c(w0.detach())), float(s0), N.ptr(_c(wr.detach())), float(sr),
                                    N.ptr(_c(br.detach())) if (ones and br is not None) else None, 1.0, N.ptr(outs[0]), N.ptr(outs[1]), N.ptr(outs[2]),
                                    N.ptr(outs[3]), acc, N.ptr(ws), wsb, B, H, W, C, N.BF16, N.stream()), "sgx_rgbconv_wgrad")
    side = _PARAM_GRAD_STREAM if accum else None
    cur_raw = N.stream()
    if side is None or side.cuda_stream == cur_raw:
        launch()
    elif not _FAST_FORK or _set_stream is None:                # A/B (SGX_FAST_FORK=0) / a torch without the raw setter: the torch-level fork
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            launch()
        img.record_stream(side); gz.record_stream(side)
        for o in fresh:
            o.record_stream(cur)
    else:                                                      # as _param_grads: fork to the side stream on raw handles
        cur = _stream_of(cur_raw)
        N.check(N.lib().sgx_stream_wait_stream(side.cuda_stream, cur_raw), "sgx_stream_wait_stream")
        _set_stream(stream_id=side.stream_id, device_index=side.device_index, device_type=side.device_type)
        try:
            launch()
        finally:
            _set_stream(stream_id=cur.stream_id, device_index=cur.device_index, device_type=cur.device_type)
        img.record_stream(side); gz.record_stream(side)
        for o in fresh:
            o.record_stream(cur)
    if not accum:
        return tuple(outs)
    for p, w, o in zip(params, want, outs):
        if w:
            if p.grad is None:
                p.grad = o
            if GRAD_NOTE is not None:
                GRAD_NOTE(p)
    return None, None, None, None


class RgbConvBlurFn(Function):
    @staticmethod
    def forward(ctx, img, w0, b0, wr, br, s0, sr):
        img = _c(img)
        if img.dtype != torch.float32 or img.dim() != 4 or img.shape[3] != 3:
            raise N.SgxError("RgbConvBlurFn: fp32 NHWC RGB image expected")
        B, H, W, _ = img.shape
        C = w0.shape[0]
        wf, wd = rgb_packs(w0, s0, wr, sr, br)
        xb = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=img.device)
        bits = torch.empty((B, H, W, C // 8), dtype=torch.uint8, device=img.device)
        N.check(N.lib().sgx_rgbconv_fwd(N.ptr(img), N.ptr(wf), N.ptr(None if b0 is None else _c(b0.detach())), N.ptr(xb), N.ptr(bits), B, H, W, C,
                                        1, 1, N.BF16, N.stream()), "sgx_rgbconv_fwd")
        ctx.cfg = (float(s0), float(sr))
        ctx.save_for_backward(img, w0, b0, wr, br)
        ctx.mark_non_differentiable(bits)
        ctx.set_materialize_grads(False)
        return xb, bits

    @staticmethod
    def backward(ctx, gz, _gbits=None):
        if gz is None:
            return (None,) * 7
        img, w0, b0, wr, br = ctx.saved_tensors
        s0, sr = ctx.cfg
        gz = _c(gz)
        gi = _bcall(RgbConvAdjFn, gz, w0, wr, br, s0, sr) if ctx.needs_input_grad[0] else None
        gw0 = gb0 = gwr = gbr = None
        if not _DATA_GRAD_ONLY:
            gw0, gb0, gwr, gbr = _bcall_wgrad(img, gz, True, w0, b0, wr, br, s0, sr, ctx.needs_input_grad[1:5])
        return gi, gw0, gb0, gwr, gbr, None, None


class RgbConvAdjFn(Function):
    @staticmethod
    def forward(ctx, gz, w0, wr, br, s0, sr):
        gz = _c(gz)
        B, H, W, C = gz.shape
        wf, wd = rgb_packs(w0, s0, wr, sr, br)
        gi = torch.empty((B, H, W, 3), dtype=torch.float32, device=gz.device)
        N.check(N.lib().sgx_rgbconv_dgrad(N.ptr(gz), N.ptr(wd), N.ptr(gi), B, H, W, C, N.dt(gz), N.stream()), "sgx_rgbconv_dgrad")
        ctx.cfg = (float(s0), float(sr))
        ctx.save_for_backward(gz, w0, wr, br)
        return gi

    @staticmethod
    def backward(ctx, gg):
        gz, w0, wr, br = ctx.saved_tensors
        s0, sr = ctx.cfg
        gg = _c(gg.float())
        ggz = _bcall(RgbConvPlainFn, gg, w0, wr, br, s0, sr) if ctx.needs_input_grad[0] else None
        gw0 = gwr = None
        if not _DATA_GRAD_ONLY:
            gw0, _, gwr, _ = _bcall_wgrad(gg, gz, False, w0, None, wr, br, s0, sr, (ctx.needs_input_grad[1], False, ctx.needs_input_grad[2], False))
        return ggz, gw0, gwr, None, None, None


class RgbConvPlainFn(Function):
    @staticmethod
    def forward(ctx, img, w0, wr, br, s0, sr):
        img = _c(img)
        B, H, W, _ = img.shape
        C = w0.shape[0]
        wf, wd = rgb_packs(w0, s0, wr, sr, br)
        z = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=img.device)
        N.check(N.lib().sgx_rgbconv_fwd(N.ptr(img), N.ptr(wf), None, N.ptr(z), None, B, H, W, C, 0, 0, N.BF16, N.stream()), "sgx_rgbconv_fwd")
        ctx.cfg = (float(s0), float(sr))
        ctx.save_for_backward(img, w0, wr, br)
        return z

    @staticmethod
    def backward(ctx, gz):
        img, w0, wr, br = ctx.saved_tensors
        s0, sr = ctx.cfg
        gz = _c(gz)
        gi = _bcall(RgbConvAdjFn, gz, w0, wr, br, s0, sr) if ctx.needs_input_grad[0] else None
        gw0 = gwr = None
        if not _DATA_GRAD_ONLY:
            gw0, _, gwr, _ = _bcall_wgrad(img, gz, False, w0, None, wr, br, s0, sr, (ctx.needs_input_grad[1], False, ctx.needs_input_grad[2], False))
        return gi, gw0, gwr, None, None, None


class RgbConvWgradFn(Function):
    """The parameter gradients as autograd outputs (first order only; the training step accumulates in-kernel instead)."""

    @staticmethod
    def forward(ctx, img, gz, ones, w0, b0, wr, br, s0, sr, want):
        outs = _rgb_wgrad(img, gz, ones, w0, b0, wr, br, s0, sr, want)
        ctx.set_materialize_grads(False)
        return tuple(o if o is not None else torch.zeros((), device=gz.device) for o in outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *g):
        raise NotImplementedError("second derivative through a weight gradient is not part of the training path")


def _bcall_wgrad(img, gz, ones, w0, b0, wr, br, s0, sr, want):
    """Parameter gradients of the composed convolution: in-kernel accumulation in the training step, else tensors for autograd
    (None where not wanted)."""
    want = tuple(bool(w) for w in want)
    if not any(want):
        return None, None, None, None
    if not torch.is_grad_enabled():
        return _rgb_wgrad(_c(img), gz, ones, w0, b0, wr, br, s0, sr, want)
    params = (w0, b0, wr, br)
    outs = RgbConvWgradFn.apply(_c(img), gz, ones, w0, b0, wr, br, s0, sr, want)
    real = [w and p is not None and (ones or k in (0, 2)) for k, (w, p) in enumerate(zip(want, params))]
    return tuple(o if r else None for o, r in zip(outs, real))


def rgbconv_blur(img, w0, b0, wr, br, s0, sr):
    """-> (blur(lrelu(conv0(from_rgb(img)))), sign bits of conv0's pre-activation)."""
    return call(RgbConvBlurFn, img, w0, b0, wr, br, float(s0), float(sr))


# ---------------------------------------------------------------------------------------------------
class LReluBwdFn(Function):
    """scale * g * (y > 0 ? 1 : slope), y = the activation's output (slope 0.2: LeakyReLU, 0: ReLU).  Linear in g; no second
    derivative w.r.t. y.  ``scale``: 1.0, a python float, or a one-element device fp32 tensor (the fade-in coefficient of the
    branch the activation sits on: lerp backward and activation backward in one pass)."""

    @staticmethod
    def forward(ctx, g, y, slope, scale=1.0):
        g = _c(g)
        out = torch.empty_like(g)
        dev = isinstance(scale, torch.Tensor)
        N.check(N.lib().sgx_lrelu_bwd(N.ptr(g), N.ptr(y), N.ptr(out), g.numel(), float(slope), 1.0 if dev else float(scale),
                                      scale.data_ptr() if dev else None, N.dt(g), N.stream()), "sgx_lrelu_bwd")
        ctx.slope, ctx.scale = float(slope), scale
        ctx.save_for_backward(y)
        return out

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        return _bcall(LReluBwdFn, gg, y, ctx.slope, ctx.scale), None, None, None


class LReluBwdBitsFn(Function):
    """scale * g * (bit ? 1 : slope): ``LReluBwdFn`` with the activation's output given as its SIGN BITS ([..., C/8] uint8, one
    byte per 8 channels) -- for activations that were never stored (``ConvDownFadeFn``).  Linear in g: its own backward."""

    @staticmethod
    def forward(ctx, g, bits, slope, scale=1.0):
        g = _c(g)
        if g.dtype != torch.bfloat16 or bits.dtype != torch.uint8 or bits.numel() * 8 != g.numel():
            raise N.SgxError("LReluBwdBitsFn: bf16 gradient and one sign byte per 8 channels expected")
        out = torch.empty_like(g)
        dev = isinstance(scale, torch.Tensor)                # a one-element device fp32 tensor (graph replay), as in LReluBwdFn
        N.check(N.lib().sgx_lrelu_bwd_bits(N.ptr(g), N.ptr(bits), N.ptr(out), g.numel(), float(slope), 1.0 if dev else float(scale),
                                           scale.data_ptr() if dev else None, N.dt(g), N.stream()), "sgx_lrelu_bwd_bits")
        ctx.slope, ctx.scale = float(slope), scale
        ctx.save_for_backward(bits)
        return out

    @staticmethod
    def backward(ctx, gg):
        (bits,) = ctx.saved_tensors
        return _bcall(LReluBwdBitsFn, gg, bits, ctx.slope, ctx.scale), None, None, None


FUSE_FADE = os.environ.get("SGX_FUSE_FADE", "1") != "0"        # A/B (tests flip the attribute; the library reads the same variable)


def conv_down_fade_ok(x, cout):
    """True if ``ConvDownFadeFn`` has a kernel for the stride-2 convolution of NHWC ``x`` to ``cout`` channels."""
    if not FUSE_FADE or x.dtype != torch.bfloat16:
        return False
    B, H, W, Cin = x.shape
    return bool(N.lib().sgx_conv4x4s2_down_fade_ok(B, H, W, Cin, int(cout), N.BF16))


class ConvDownFadeFn(Function):
    """alpha * lrelu(conv_down(x) + bias) + beta * resid in ONE kernel: the tail of the discriminator's newest block with the
    fade-in lerp (reference models/Blocks.py:143-146 + models/GAN.py:425-427) in the convolution's store; the activation itself is
    never written, its sign bits are (the mask of its backward).  Backward: d resid = beta * g; the convolution's upstream gradient
    alpha * g * slope(bits) is one pass (``LReluBwdBitsFn``), then exactly ``ConvFn``'s backward (incl. ``x_pre_bits``: the blur and
    activation mask in front of this layer)."""

    @staticmethod
    def forward(ctx, x, weight, bias, resid, scale, ipad, alpha, beta, x_pre=None, x_pre_bits=None):
        x, resid = _c(x), _c(resid)
        fwd, adj = packs(weight, "D", scale, ipad, x.dtype)
        B, H, W, Cin = x.shape
        taps, Cout, K = fwd.shape
        if K != Cin or tuple(resid.shape) != (B, H // 2, W // 2, Cout) or resid.dtype != x.dtype:
            raise N.SgxError("conv+fade: residual branch must have the output's shape and dtype")
        y = torch.empty((B, H // 2, W // 2, Cout), dtype=x.dtype, device=x.device)
        bits = torch.empty((B, H // 2, W // 2, Cout // 8), dtype=torch.uint8, device=x.device)
        dev = isinstance(alpha, torch.Tensor)                # [alpha, 1 - alpha] in device memory (graph replay); beta is then ignored
        if dev and not (alpha.dtype == torch.float32 and alpha.numel() == 2 and alpha.is_contiguous()):
            raise N.SgxError("conv+fade: device coefficients must be a contiguous fp32 [alpha, beta] pair")
        N.check(N.lib().sgx_conv4x4s2_down_fade(N.ptr(x), N.ptr(fwd), N.ptr(None if bias is None else _c(bias.detach())), N.ptr(resid),
                                                0.0 if dev else float(alpha), 0.0 if dev else float(beta), alpha.data_ptr() if dev else None,
                                                N.ptr(y), N.ptr(bits), B, H, W, Cin, Cout, N.dt(x), N.stream()), "sgx_conv4x4s2_down_fade")
        # ConvFn.backward's view of this op: the stride-2 layer with its activation already undone (see backward)
        ctx.cfg = ("D", scale, ipad, False, 0, bias is not None, False, False)
        ctx.bias_ref = weakref.ref(bias) if bias is not None else (lambda: None)
        ctx.fade = (None, None) if dev else (float(alpha), float(beta))
        ctx.save_for_backward(x, weight, None, None, x_pre, x_pre_bits, bits, alpha if dev else None)   # (ConvFn's layout + bits, device coefficients)
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        bits, alpha_dev = ctx.saved_tensors[6:8]
        alpha, beta = ctx.fade
        if alpha_dev is not None:
            alpha = alpha_dev
        g_res = None
        if isinstance(alpha, torch.Tensor):
            if ctx.needs_input_grad[3]:
                g_res = _bcall(ScaleDevFn, g, alpha[1:2])
            gy = _bcall(LReluBwdBitsFn, g, bits, 0.2, alpha[0:1])
        else:
            if ctx.needs_input_grad[3]:
                g_res = g if beta == 1.0 else _bcall(ScaleFn, g, beta)
            gy = _bcall(LReluBwdBitsFn, g, bits, 0.2, alpha)
        out = ConvFn.backward(ctx, gy)                     # (x, weight, bias are inputs 0..2 of both Functions)
        return out[0], out[1], out[2], g_res, None, None, None, None, None, None


FUSE_FADE_RGB = os.environ.get("SGX_FUSE_FADE_RGB", "1") != "0"        # A/B: the residual branch from_rgb(pool(img)) computed in the lerp's store


class RgbResidual:
    """The discriminator's residual branch ``from_rgb(pimg)`` (reference models/GAN.py:423-427) as a RECIPE instead of a tensor: the
    pooled image, the 1x1 layer's parameters and scales (``out_scale``: the (1 - alpha) prescale when alpha is a host number).
    ``ConvDownFadeRgbFn`` evaluates it inside the store of the newest block's stride-2 convolution; ``materialize`` runs the layer
    (``RgbInFn``, the unfused path -- where the lerp is a pass of its own)."""

    def __init__(self, pimg, layer, out_scale, dtype):
        self.pimg, self.layer, self.out_scale, self.dtype = pimg, layer, float(out_scale), dtype

    def materialize(self):
        return self.layer.forward_nhwc(self.pimg, out_dtype=self.dtype, out_scale=self.out_scale)


def fade_rgb_ok(layer, cout, dtype):
    """True if ``ConvDownFadeRgbFn`` takes the residual layer: a 3 -> cout 1x1 convolution feeding bf16 activations, cout in {32, 64, 128}."""
    return (FUSE_FADE and FUSE_FADE_RGB and dtype == torch.bfloat16 and int(cout) in (32, 64, 128) and layer.kernel_size == 1
            and tuple(layer.weight.shape) == (int(cout), 3, 1, 1))


FUSE_FADE_BWD2 = os.environ.get("SGX_FUSE_FADE_BWD2", "1") != "0"      # A/B: 0 = the differentiable composition under create_graph


class FadeRgbBwdFn(Function):
    """g -> (gy, gpimg): the data half of the backward of the newest discriminator block's tail (``ConvDownFadeRgbFn``) as ONE autograd node
    with a one-pass adjoint -- for the R1 penalty's inner gradient (``create_graph``), which until round 6 ran the differentiable composition
    ``LReluBwdBitsFn`` + [``ScaleDevFn``] + ``RgbOutFn`` (2-3 passes over g) and whose own backward then ran their three adjoints plus
    autograd's add over the [B, H, W, C] tensor.  Forward = the composition's own kernels (bit-identical inner gradient); backward =
    ``sgx_fade_rgb_bwd2`` (same roundings as the passes it replaces: bit-identical) + from_rgb's weight gradient of the R1 term."""

    @staticmethod
    def forward(ctx, g, bits, pimg, wr, ws, alpha, beta, alpha_dev, need_img):
        # The SAME kernels, hence the same bits, as the composition this Function replaces (mask pass, [scaling pass,] to_rgb-shaped pass):
        # the one-pass sgx_fade_rgb_bwd sums the image gradient in another order, and the R1 double backward amplifies that last-bit
        # difference to 1e-3 in the parameter gradients (tools/diag_bwd2.py) -- harmless against the 5e-2 bf16 bars, but the
        # fused-vs-unfused tests hold this path to 1e-5
        g = _c(g)
        dev = alpha_dev is not None
        with torch.no_grad():
            gy = LReluBwdBitsFn.forward(_NoGradCtx(), g, bits, 0.2, alpha_dev[0:1] if dev else alpha)
            gpimg = None
            if need_img:
                g_res = ScaleDevFn.forward(_NoGradCtx(), g, alpha_dev[1:2]) if dev else (g if beta == 1.0 else ScaleFn.forward(_NoGradCtx(), g, beta))
                gpimg = RgbOutFn.forward(_NoGradCtx(), g_res, wr, None, ws)
        ctx.cfg = (float(ws), None if dev else float(alpha), None if dev else float(beta), need_img)
        ctx.save_for_backward(g, bits, wr, alpha_dev)
        ctx.set_materialize_grads(False)
        return gy, gpimg

    @staticmethod
    @once_differentiable
    def backward(ctx, ggy, ggp):
        g, bits, wr, alpha_dev = ctx.saved_tensors
        ws, alpha, beta, need_img = ctx.cfg
        if ggy is None and ggp is None:
            return (None,) * 9
        dev = alpha_dev is not None
        L = N.lib()
        C = g.shape[-1]
        npix = g.numel() // C
        out = torch.empty_like(g)
        ggy_c = None if ggy is None else _c(ggy)
        ggp_c = None if ggp is None else _c(ggp)
        N.check(L.sgx_fade_rgb_bwd2(N.ptr(ggy_c), N.ptr(ggp_c), N.ptr(bits), N.ptr(_c(wr.detach())), ws * (1.0 if dev else beta), 0.0 if dev else alpha, 1.0,
                                    alpha_dev.data_ptr() if dev else None, N.ptr(out), npix, C, N.dt(g), N.stream()), "sgx_fade_rgb_bwd2")
        gwr = None
        if ggp_c is not None and ctx.needs_input_grad[3] and not _DATA_GRAD_ONLY:
            # d gpimg / d wr: the R1 term reaches the residual from_rgb's weight through the image gradient
            gwr = RgbWgradFn.forward(_NoGradCtx(), ggp_c, g, wr, ws * (1.0 if dev else beta))
            if dev:
                gwr = gwr * alpha_dev[1]
        return out, None, None, gwr, None, None, None, None, None


class ConvDownFadeRgbFn(Function):
    """``ConvDownFadeFn`` with the residual branch evaluated in the store: alpha * lrelu(conv_down(x) + bias) + beta * from_rgb(pimg),
    from_rgb(pimg)[c] = bf16(rb[c] * bs1 * bs2 + ws * sum_j pimg[j] * wr[c][j]) -- the arithmetic of ``RgbInFn`` on a bf16 output, bit for
    bit -- so the [B, H/2, W/2, C] residual tensor is neither written nor read (12 bytes of image per pixel instead of 2 x 2C).
    Backward, first order (the training step): ONE pass over the incoming gradient (``sgx_fade_rgb_bwd``) gives the convolution's
    upstream gradient alpha * g * slope(bits) (bit for bit ``LReluBwdBitsFn``), from_rgb's weight and bias gradients (accumulated
    straight into ``.grad`` under ``accumulate_param_grads``) and, where the image needs one, its gradient -- instead of the mask
    pass + ``RgbWgradFn`` + ``ColSumFn`` + ``RgbOutFn``.  Under ``create_graph`` (the R1 penalty's inner gradient) the backward is
    the composition of those twice-differentiable ops, as for the unfused branch."""

    @staticmethod
    def forward(ctx, x, weight, bias, pimg, wr, br, scale, ipad, alpha, beta, ws, bs1, bs2, x_pre=None, x_pre_bits=None):
        x, pimg = _c(x), _c(pimg)
        fwd, adj = packs(weight, "D", scale, ipad, x.dtype)
        B, H, W, Cin = x.shape
        taps, Cout, K = fwd.shape
        if K != Cin or tuple(pimg.shape) != (B, H // 2, W // 2, 3) or pimg.dtype != torch.float32 or tuple(wr.shape) != (Cout, 3, 1, 1):
            raise N.SgxError("conv+fade(rgb): pooled fp32 image [B, H/2, W/2, 3] and a [Cout, 3, 1, 1] weight expected")
        y = torch.empty((B, H // 2, W // 2, Cout), dtype=x.dtype, device=x.device)
        bits = torch.empty((B, H // 2, W // 2, Cout // 8), dtype=torch.uint8, device=x.device)
        dev = isinstance(alpha, torch.Tensor)                # [alpha, 1 - alpha] in device memory (graph replay); beta is then ignored
        if dev and not (alpha.dtype == torch.float32 and alpha.numel() == 2 and alpha.is_contiguous()):
            raise N.SgxError("conv+fade(rgb): device coefficients must be a contiguous fp32 [alpha, beta] pair")
        N.check(N.lib().sgx_conv4x4s2_down_fade_rgb(N.ptr(x), N.ptr(fwd), N.ptr(None if bias is None else _c(bias.detach())), N.ptr(pimg),
                                                    N.ptr(_c(wr.detach())), float(ws), N.ptr(None if br is None else _c(br.detach())), float(bs1), float(bs2),
                                                    0.0 if dev else float(alpha), 0.0 if dev else float(beta), alpha.data_ptr() if dev else None,
                                                    N.ptr(y), N.ptr(bits), B, H, W, Cin, Cout, N.dt(x), N.stream()), "sgx_conv4x4s2_down_fade_rgb")
        ctx.cfg = ("D", scale, ipad, False, 0, bias is not None, False, False)
        ctx.bias_ref = weakref.ref(bias) if bias is not None else (lambda: None)
        ctx.br_ref = weakref.ref(br) if br is not None else (lambda: None)
        ctx.fade = ((None, None) if dev else (float(alpha), float(beta))) + (float(ws), float(bs1) * float(bs2))
        ctx.save_for_backward(x, weight, None, None, x_pre, x_pre_bits, bits, alpha if dev else None, pimg, wr)
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        bits, alpha_dev, pimg, wr = ctx.saved_tensors[6:10]
        alpha, beta, ws, bs = ctx.fade
        br = ctx.br_ref()
        need_img = ctx.needs_input_grad[3]
        want_w = ctx.needs_input_grad[4] and not _DATA_GRAD_ONLY
        want_b = br is not None and ctx.needs_input_grad[5] and not _DATA_GRAD_ONLY
        gpimg = gwr = gbr = None
        if torch.is_grad_enabled() and FUSE_FADE_BWD2 and not want_w and not want_b:
            # R1's inner gradient (create_graph; data gradients only): one pass now, one pass when it is differentiated (FadeRgbBwdFn)
            gy, gpimg = FadeRgbBwdFn.apply(g, bits, pimg, wr, ws, alpha, beta, alpha_dev, bool(need_img))
        elif torch.is_grad_enabled():
            # the differentiable composition (R1: this backward is itself differentiated)
            if alpha_dev is not None:
                g_res = _bcall(ScaleDevFn, g, alpha_dev[1:2])
                gy = _bcall(LReluBwdBitsFn, g, bits, 0.2, alpha_dev[0:1])
            else:
                g_res = g if beta == 1.0 else _bcall(ScaleFn, g, beta)
                gy = _bcall(LReluBwdBitsFn, g, bits, 0.2, alpha)
            if need_img:
                gpimg = _bcall(RgbOutFn, g_res, wr, None, ws)
            if want_w:
                gwr = _bcall(RgbWgradFn, pimg, g_res, wr, ws)
            if want_b:
                gbr = _bcall(ColSumFn, g_res, bs)
        else:
            L = N.lib()
            C = g.shape[-1]
            npix = g.numel() // C
            accum = _ACCUM_PARAM_GRADS and wr.is_leaf and (br is None or br.is_leaf)
            acc = 0
            dw = db = None
            if want_w and accum and wr.grad is not None:
                dw, acc = wr.grad, acc | 1
            if want_b and accum and br.grad is not None:
                db, acc = br.grad, acc | 2
            gy = torch.empty_like(g)                         # (fresh dw / db are allocated under the stream that writes them, below)
            if need_img:
                gpimg = torch.empty_like(pimg)
            wsb = L.sgx_fade_rgb_bwd_ws_bytes(npix, C)
            wsp = N.workspace(wsb, g.device)
            dev = alpha_dev is not None
            side = _PARAM_GRAD_STREAM if (accum and (want_w or want_b)) else None
            cur_raw = N.stream()
            if side is not None and side.cuda_stream == cur_raw:
                side = None
            if side is None:
                if want_w and dw is None:
                    dw = torch.empty(wr.shape, dtype=torch.float32, device=g.device)
                if want_b and db is None:
                    db = torch.empty(br.shape, dtype=torch.float32, device=g.device)
            # with a side stream: this pass leaves the block partials in wsp, and the .grad write / accumulate runs where every other
            # accumulation into .grad of the step runs (D(fake)'s backward is on the auxiliary stream, D(real)'s on the main one, and both
            # reach this from_rgb: unordered read-modify-writes of one gradient otherwise)
            N.check(L.sgx_fade_rgb_bwd(N.ptr(g), N.ptr(bits), N.ptr(pimg), N.ptr(_c(wr.detach())), ws, bs, 0.0 if dev else alpha, 0.0 if dev else beta,
                                       alpha_dev.data_ptr() if dev else None, N.ptr(gy), None if side is not None else N.ptr(dw),
                                       None if side is not None else N.ptr(db), acc, N.ptr(gpimg), N.ptr(wsp), wsb, npix, C,
                                       N.dt(g), cur_raw), "sgx_fade_rgb_bwd")
            if side is not None:
                fresh = []

                def finish():
                    nonlocal dw, db
                    if want_w and dw is None:
                        dw = torch.empty(wr.shape, dtype=torch.float32, device=g.device); fresh.append(dw)
                    if want_b and db is None:
                        db = torch.empty(br.shape, dtype=torch.float32, device=g.device); fresh.append(db)
                    N.check(L.sgx_fade_rgb_bwd_finish(N.ptr(wsp), wsb, npix, C, ws, bs, 0.0 if dev else beta, alpha_dev.data_ptr() if dev else None,
                                                      N.ptr(dw), N.ptr(db), acc, N.stream()), "sgx_fade_rgb_bwd_finish")
                _fork_to(side, cur_raw, finish, reads=(wsp,) + ((alpha_dev,) if dev else ()), fresh=fresh)
            if accum:
                for p_, d_ in ((wr, dw), (br, db)):
                    if d_ is not None:
                        if p_.grad is None:
                            p_.grad = d_
                        if GRAD_NOTE is not None:
                            GRAD_NOTE(p_)
            else:
                gwr, gbr = dw, db
        out = ConvFn.backward(ctx, gy)                     # (x, weight, bias are inputs 0..2 of both Functions)
        return (out[0], out[1], out[2], gpimg, gwr, gbr) + (None,) * 9


class ColSumFn(Function):
    """[..., C] -> fp32 [C], times ``scale`` (bias gradient)."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _c(x)
        C = x.shape[-1]
        npix = x.numel() // C
        out = torch.empty((C,), dtype=torch.float32, device=x.device)
        L = N.lib()
        ws = N.workspace(L.sgx_colsum_ws_bytes(npix, C), x.device)
        N.check(L.sgx_colsum(N.ptr(x), N.ptr(out), float(scale), N.ptr(ws), ws.numel(), npix, C, N.dt(x), N.stream()), "sgx_colsum")
        ctx.shape, ctx.dtype, ctx.scale = x.shape, x.dtype, float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        return (g * ctx.scale).to(ctx.dtype).expand(ctx.shape), None


class BiasActFn(Function):
    """y = act(x + bscale*bias[c]) on [..., C]; act: ACT_NONE | ACT_LRELU | ACT_RELU."""

    @staticmethod
    def forward(ctx, x, bias, bscale, act):
        x = _c(x)
        C = x.shape[-1]
        y = torch.empty_like(x)
        N.check(N.lib().sgx_bias_act(N.ptr(x), N.ptr(None if bias is None else _c(bias.detach())), float(bscale), N.ptr(y),
                                     x.numel() // C, C, act, N.dt(x), N.stream()), "sgx_bias_act")
        ctx.act, ctx.has_bias, ctx.bscale = act, bias is not None, float(bscale)
        ctx.save_for_backward(y if act else None)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _c(g)
        if ctx.act:
            g = _bcall(LReluBwdFn, g, y, 0.0 if ctx.act == N.ACT_RELU else 0.2, 1.0)
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[1] and not _DATA_GRAD_ONLY:
            gb = _bcall(ColSumFn, g, ctx.bscale)
        return g, gb, None, None


class ScaleFn(Function):
    @staticmethod
    def forward(ctx, x, s):
        x = _c(x)
        out = torch.empty_like(x)
        N.check(N.lib().sgx_axpby(N.ptr(x), None, N.ptr(out), float(s), 0.0, x.numel(), N.dt(x), N.stream()), "sgx_axpby")
        ctx.s = float(s)
        return out

    @staticmethod
    def backward(ctx, g):
        return _bcall(ScaleFn, g, ctx.s), None


class AxpbyFn(Function):
    """alpha*a + beta*b (fade-in lerp).  ``a_act``: ``a`` is a LeakyReLU output whose activation backward was deferred to its
    consumer (ConvFn ``defer_act``): the gradient of ``a`` is alpha * g * slope(a), one pass."""

    @staticmethod
    def forward(ctx, a, b, alpha, beta, a_act=False):
        a, b = _c(a), _c(b)
        assert a.shape == b.shape and a.dtype == b.dtype
        out = torch.empty_like(a)
        N.check(N.lib().sgx_axpby(N.ptr(a), N.ptr(b), N.ptr(out), float(alpha), float(beta), a.numel(), N.dt(a), N.stream()), "sgx_axpby")
        ctx.alpha, ctx.beta, ctx.a_act = float(alpha), float(beta), bool(a_act)
        ctx.save_for_backward(a if a_act else None)
        return out

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = _bcall(LReluBwdFn, g, a, 0.2, ctx.alpha) if ctx.a_act else _bcall(ScaleFn, g, ctx.alpha)
        if ctx.needs_input_grad[1]:
            gb = g if ctx.beta == 1.0 else _bcall(ScaleFn, g, ctx.beta)
        return ga, gb, None, None, None


class ScaleDevFn(Function):
    """x * s[0], s a device fp32 tensor (no gradient w.r.t. s)."""

    @staticmethod
    def forward(ctx, x, s):
        x = _c(x)
        out = torch.empty_like(x)
        N.check(N.lib().sgx_axpby_dev(N.ptr(x), None, N.ptr(out), s.data_ptr(), None, x.numel(), N.dt(x), N.stream()), "sgx_axpby_dev")
        ctx.s = s
        return out

    @staticmethod
    def backward(ctx, g):
        return _bcall(ScaleDevFn, g, ctx.s), None


class FadeFn(Function):
    """ab[0]*a + ab[1]*b with the two coefficients in a device fp32 tensor (fade-in under graph replay).  ``a_act`` as in
    ``AxpbyFn``."""

    @staticmethod
    def forward(ctx, a, b, ab, a_act=False):
        a, b = _c(a), _c(b)
        assert a.shape == b.shape and a.dtype == b.dtype and ab.dtype == torch.float32 and ab.numel() == 2 and ab.is_contiguous()
        out = torch.empty_like(a)
        N.check(N.lib().sgx_axpby_dev(N.ptr(a), N.ptr(b), N.ptr(out), ab.data_ptr(), ab.data_ptr() + 4, a.numel(), N.dt(a), N.stream()),
                "sgx_axpby_dev")
        ctx.ab, ctx.a_act = ab, bool(a_act)
        ctx.save_for_backward(a if a_act else None)
        return out

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = _bcall(LReluBwdFn, g, a, 0.2, ctx.ab[0:1]) if ctx.a_act else _bcall(ScaleDevFn, g, ctx.ab[0:1])
        if ctx.needs_input_grad[1]:
            gb = _bcall(ScaleDevFn, g, ctx.ab[1:2])
        return ga, gb, None, None


def fade(a, b, alpha, a_act=False, b_prescaled=False):
    """alpha*a + (1-alpha)*b.  ``alpha``: python float, or a device fp32 tensor [alpha, 1-alpha] (graph replay).  ``a_act``:
    ``a`` is a LeakyReLU output with its activation backward deferred to this op.  ``b_prescaled``: ``b`` already carries its
    (1-alpha) (folded into the layer that produced it; python-float alpha only), so its gradient is g itself."""
    if isinstance(alpha, torch.Tensor):
        assert not b_prescaled
        return call(FadeFn, a, b, alpha, bool(a_act))
    return call(AxpbyFn, a, b, float(alpha), 1.0 if b_prescaled else float(1 - alpha), bool(a_act))


def downsample_fade_rgb(x, alpha):
    """alpha * x + (1 - alpha) * nearest_up2(avgpool2(x)) for an fp32 NHWC RGB batch that needs no gradient (the real images
    of a training step, reference models/GAN.py:575-586), one pass.  ``alpha`` as in ``fade``."""
    x = _c(x)
    B, H, W, C = x.shape
    assert C == 3 and x.dtype == torch.float32 and not x.requires_grad
    out = torch.empty_like(x)
    dev = isinstance(alpha, torch.Tensor)
    N.check(N.lib().sgx_downsample_fade_rgb(N.ptr(x), N.ptr(out), B, H, W, 0.0 if dev else float(alpha), 0.0 if dev else float(1 - alpha),
                                            alpha.data_ptr() if dev else None, N.stream()), "sgx_downsample_fade_rgb")
    return out


class BlurFn(Function):
    """Depthwise [1,2,1]x[1,2,1]/16 blur with zero padding; self-adjoint."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, H, W, C = x.shape
        y = torch.empty_like(x)
        N.check(N.lib().sgx_blur3x3(N.ptr(x), N.ptr(y), B, H, W, C, N.dt(x), N.stream()), "sgx_blur3x3")
        return y

    @staticmethod
    def backward(ctx, g):
        return _bcall(BlurFn, g)


class BlurStatsFn(Function):
    """y = blur(x) plus the statistics pass of the LayerEpilogue that consumes y, out of the same kernel: ``part`` holds the
    per-block partial (sum a, sum a^2) of a = lrelu(y + bias + nw*noise) per (image, channel) -- handed to ``GEpilogueFn``
    (``pre``), which then reads y once instead of twice.  ``part`` is not differentiable: the epilogue's own backward carries
    the dependence of the statistics on y."""

    @staticmethod
    def forward(ctx, x, bias, noise, nw):
        x = _c(x)
        B, H, W, C = x.shape
        y = torch.empty_like(x)
        L = N.lib()
        npart = L.sgx_blur3x3_stats_nparts(B, H, W, C, N.dt(x))
        part = torch.empty((B, npart, C, 2), dtype=torch.float64, device=x.device)
        noise_c = _c(noise.detach().reshape(B, H * W))
        if noise_c.dtype != torch.float32:
            noise_c = noise_c.float()
        N.check(L.sgx_blur3x3_stats(N.ptr(x), N.ptr(y), N.ptr(None if bias is None else _c(bias.detach())), N.ptr(noise_c),
                                    N.ptr(_c(nw.detach())), N.ptr(part), part.numel() * 8, B, H, W, C, N.ACT_LRELU, N.dt(x), N.stream()),
                "sgx_blur3x3_stats")
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)
        return y, part

    @staticmethod
    def backward(ctx, g, _gpart):
        return (_bcall(BlurFn, g) if g is not None else None), None, None, None


class BlurGenFn(Function):
    """Depthwise K x K correlation with zero padding for blur filters other than [1,2,1] (reference BlurLayer,
    models/CustomLayers.py:251-276): ``taps`` = the K*K kernel as a tuple of floats (row major), ``pad`` the zero padding,
    output [B, OH, OW, C].  Its adjoint is the same op with the taps flipped, pad' = K-1-pad and the sizes swapped, so it is
    closed under differentiation."""

    @staticmethod
    def forward(ctx, x, taps, K, pad, OH, OW):
        import ctypes
        x = _c(x)
        B, IH, IW, C = x.shape
        y = torch.empty((B, OH, OW, C), dtype=x.dtype, device=x.device)
        arr = (ctypes.c_float * (K * K))(*taps)
        N.check(N.lib().sgx_blur_kxk(N.ptr(x), N.ptr(y), ctypes.addressof(arr), K, pad, B, IH, IW, OH, OW, C, N.dt(x), N.stream()), "sgx_blur_kxk")
        ctx.cfg = (taps, K, pad, IH, IW)
        return y

    @staticmethod
    def backward(ctx, g):
        taps, K, pad, IH, IW = ctx.cfg
        return _bcall(BlurGenFn, g, tuple(reversed(taps)), K, K - 1 - pad, IH, IW), None, None, None, None, None


def _blur_act(x, z, mode, bits=None):
    """``bits`` (modes 2, 3): the sign bits of z ([B,H,W,C/8] uint8, written by the convolution that produced z) instead of z."""
    x = _c(x)
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    if bits is not None:
        N.check(N.lib().sgx_blur3x3_bits(N.ptr(x), N.ptr(bits), N.ptr(y), B, H, W, C, mode, N.dt(x), N.stream()), "sgx_blur3x3_bits")
        return y
    N.check(N.lib().sgx_blur3x3_act(N.ptr(x), None if z is None else N.ptr(z), N.ptr(y), B, H, W, C, mode, N.dt(x), N.stream()), "sgx_blur3x3_act")
    return y


class ActBlurFn(Function):
    """blur(lrelu(z)): the discriminator block's activation folded into its blur pass (one kernel forward, one backward)."""

    @staticmethod
    def forward(ctx, z):
        z = _c(z)
        ctx.save_for_backward(z)
        return _blur_act(z, None, 1)

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        return _bcall(BlurMaskFn, g, z)


class BlurMaskFn(Function):
    """blur(g) * slope(z): backward of ActBlurFn.  Linear in g; z only selects the slope (``bits``: its sign bits, read instead
    of z where the producing convolution wrote them)."""

    @staticmethod
    def forward(ctx, g, z, bits=None):
        ctx.save_for_backward(z, bits)
        return _blur_act(g, z, 2, bits)

    @staticmethod
    def backward(ctx, gg):
        z, bits = ctx.saved_tensors
        return _bcall(MaskBlurFn, gg, z, bits), None, None


class MaskBlurFn(Function):
    """blur(g * slope(z)): adjoint of BlurMaskFn in g (the blur is self-adjoint)."""

    @staticmethod
    def forward(ctx, g, z, bits=None):
        ctx.save_for_backward(z, bits)
        return _blur_act(g, z, 3, bits)

    @staticmethod
    def backward(ctx, gg):
        z, bits = ctx.saved_tensors
        return _bcall(BlurMaskFn, gg, z, bits), None, None


class Pool2Fn(Function):
    """scale * (2x2 block sum); adjoint = scale * nearest-up."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _c(x)
        B, H, W, C = x.shape
        y = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
        N.check(N.lib().sgx_pool2(N.ptr(x), N.ptr(y), B, H, W, C, float(scale), N.dt(x), N.stream()), "sgx_pool2")
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, g):
        return _bcall(Up2Fn, g, ctx.scale), None


class Up2Fn(Function):
    """scale * nearest-neighbour x2; adjoint = scale * (2x2 block sum)."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _c(x)
        B, H, W, C = x.shape
        y = torch.empty((B, 2 * H, 2 * W, C), dtype=x.dtype, device=x.device)
        N.check(N.lib().sgx_up2(N.ptr(x), N.ptr(y), B, H, W, C, float(scale), N.dt(x), N.stream()), "sgx_up2")
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, g):
        return _bcall(Pool2Fn, g, ctx.scale), None


class UpAddFn(Function):
    """a + scale * nearest-up(x) in one pass (``sgx_up2_add``); adjoint = (g, scale * 2x2 block sum of g)."""

    @staticmethod
    def forward(ctx, a, x, scale):
        a, x = _c(a), _c(x)
        B, H, W, C = x.shape
        if tuple(a.shape) != (B, 2 * H, 2 * W, C) or a.dtype != x.dtype:
            raise N.SgxError("UpAddFn: a must be [B, 2H, 2W, C] of x's dtype")
        y = torch.empty_like(a)
        N.check(N.lib().sgx_up2_add(N.ptr(x), N.ptr(a), N.ptr(y), B, H, W, C, float(scale), N.dt(x), N.stream()), "sgx_up2_add")
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, _bcall(Pool2Fn, g, ctx.scale), None


POOL_FORK = os.environ.get("SGX_POOL_FORK", "1") != "0"      # A/B: 0 = Pool2Fn and autograd's own sum of the two image gradients


class PoolForkFn(Function):
    """x -> (x, scale * pool2(x)) for a tensor that feeds BOTH a full-resolution branch and a pooled one -- the discriminator's image under
    fade-in: the newest block on the image, the residual from_rgb on its 2x2 average (reference models/GAN.py:423-427).  Forward = ``Pool2Fn``;
    the point is the backward: autograd would run the pool's adjoint (nearest-up: one pass writing a full-resolution fp32 image) and then sum
    it with the other branch's gradient in a pass of its own (read 2, write 1); here the two meet in ONE pass, ``g_x + scale * up(g_y)``
    (``UpAddFn``: twice differentiable, so the R1 double backward goes through it unchanged)."""

    @staticmethod
    def forward(ctx, x, scale):
        y = Pool2Fn.forward(_NoGradCtx(), x, scale)
        ctx.scale = float(scale)
        ctx.set_materialize_grads(False)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, gx, gy):
        if gy is None:
            return gx, None
        if gx is None:
            return _bcall(Up2Fn, gy, ctx.scale), None
        return _bcall(UpAddFn, gx, gy, ctx.scale), None


# ---------------------------------------------------------------------------------------------------
# 1x1 RGB convolutions.  Images are fp32 [B,H,W,3]; the weight is the raw parameter ([C,3,1,1] from_rgb or
# [3,C,1,1] to_rgb), read in place: element (j, c) at w[j*sj + c*sc], times wscale (= w_mul).
def _dtype_code(dtype):
    return N.F32 if dtype == torch.float32 else N.BF16


def rgb_layout(weight):
    """(sj, sc, C) for a from_rgb [C,3,1,1] or to_rgb [3,C,1,1] parameter."""
    if weight.shape[1] == 3 and weight.shape[0] != 3:
        return 1, 3, weight.shape[0]
    if weight.shape[0] == 3:
        return weight.shape[1], 1, weight.shape[1]
    raise N.SgxError("1x1 convolution is built for the RGB layers (3 channels on one side)")


class RgbInFn(Function):
    """f[p][c] = bias[c] + wscale * sum_j img[p][j] * W(j,c)."""

    @staticmethod
    def forward(ctx, img, weight, bias, wscale, out_dtype):
        img, w = _c(img), _c(weight.detach())
        sj, sc, C = rgb_layout(w)
        B, H, W, _ = img.shape
        y = torch.empty((B, H, W, C), dtype=out_dtype, device=img.device)
        N.check(N.lib().sgx_rgb_in(N.ptr(img), N.ptr(w), sj, sc, float(wscale), N.ptr(None if bias is None else _c(bias.detach())), N.ptr(y),
                                   B * H * W, C, _dtype_code(out_dtype), N.stream()), "sgx_rgb_in")
        ctx.has_bias, ctx.wscale = bias is not None, float(wscale)
        ctx.save_for_backward(img, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        img, weight = ctx.saved_tensors
        g = _c(g)
        gi = gw = gb = None
        if ctx.needs_input_grad[0]:
            gi = _bcall(RgbOutFn, g, weight, None, ctx.wscale)
        if not _DATA_GRAD_ONLY:
            if ctx.needs_input_grad[1]:
                gw = _bcall(RgbWgradFn, img, g, weight, ctx.wscale)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = _bcall(ColSumFn, g, 1.0)
        return gi, gw, gb, None, None


class RgbOutFn(Function):
    """img[p][j] = bias[j] + wscale * sum_c x[p][c] * W(j,c)."""

    @staticmethod
    def forward(ctx, x, weight, bias, wscale):
        x, w = _c(x), _c(weight.detach())
        sj, sc, C = rgb_layout(w)
        B, H, W, Cx = x.shape
        assert Cx == C
        img = torch.empty((B, H, W, 3), dtype=torch.float32, device=x.device)
        N.check(N.lib().sgx_rgb_out(N.ptr(x), N.ptr(w), sj, sc, float(wscale), N.ptr(None if bias is None else _c(bias.detach())), N.ptr(img),
                                    B * H * W, C, N.dt(x), N.stream()), "sgx_rgb_out")
        ctx.has_bias, ctx.wscale = bias is not None, float(wscale)
        ctx.save_for_backward(x, weight)
        return img

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _c(g)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _bcall(RgbInFn, g, weight, None, ctx.wscale, x.dtype)
        if not _DATA_GRAD_ONLY:
            if ctx.needs_input_grad[1]:
                gw = _bcall(RgbWgradFn, g, x, weight, ctx.wscale)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = _bcall(ColSumFn, g, 1.0)                   # 3 numbers
        return gx, gw, gb, None


RGB_FORK = os.environ.get("SGX_RGB_FORK", "1") != "0"        # A/B: 0 = RgbOutFn and autograd's own sum of the activation's two gradients


class RgbOutForkFn(Function):
    """x -> (x, to_rgb(x)) for an activation with TWO consumers: the generator under fade-in hands a block's output to the next block AND to
    the previous resolution's to_rgb (reference models/GAN.py:199-202).  Forward = ``RgbOutFn``.  Backward: to_rgb's data gradient is written
    ON TOP of the gradient that came back from the other consumer (``sgx_rgb_in_add``: one pass, one rounding) instead of ``sgx_rgb_in``
    followed by autograd's add pass (read 2, write 1 of the activation).  The generator is never under a double backward; should this
    backward itself be differentiated it falls back to the differentiable composition."""

    @staticmethod
    def forward(ctx, x, weight, bias, wscale):
        img = RgbOutFn.forward(ctx, x, weight, bias, wscale)         # (saves x, weight; sets has_bias, wscale)
        ctx.set_materialize_grads(False)
        return x.view_as(x), img

    @staticmethod
    def backward(ctx, gx, g):
        if g is None:
            return gx, None, None, None
        x, weight = ctx.saved_tensors
        g = _c(g)
        out = gw = gb = None
        if ctx.needs_input_grad[0]:
            if gx is None:
                out = _bcall(RgbInFn, g, weight, None, ctx.wscale, x.dtype)
            elif torch.is_grad_enabled():
                out = gx + _bcall(RgbInFn, g, weight, None, ctx.wscale, x.dtype)
            else:
                gx, w = _c(gx), _c(weight.detach())
                sj, sc, C = rgb_layout(w)
                B, H, W, _ = g.shape
                out = torch.empty_like(gx)
                N.check(N.lib().sgx_rgb_in_add(N.ptr(g), N.ptr(w), sj, sc, ctx.wscale, N.ptr(gx), N.ptr(out), B * H * W, C, N.dt(gx), N.stream()),
                        "sgx_rgb_in_add")
        else:
            out = gx
        if not _DATA_GRAD_ONLY:
            if ctx.needs_input_grad[1]:
                gw = _bcall(RgbWgradFn, g, x, weight, ctx.wscale)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = _bcall(ColSumFn, g, 1.0)
        return out, gw, gb, None


class RgbOutFadeFn(Function):
    """img = alpha * to_rgb(x) + (1 - alpha) * nearest_up2(low): the generator's output (1x1 convolution, upsample of the
    previous resolution's RGB image and fade-in lerp, reference models/GAN.py:199-202) in ONE pass over x.  ``alpha``: python
    float, or a device fp32 tensor [alpha, 1 - alpha] (graph replay).  Backward: the lerp coefficients ride in the scales of the
    1x1 convolution's own gradient kernels (python-float alpha), so no scaling pass touches the image-sized gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, wscale, low, alpha):
        x, w, low = _c(x), _c(weight.detach()), _c(low)
        sj, sc, C = rgb_layout(w)
        B, H, W, Cx = x.shape
        assert Cx == C and low.shape == (B, H // 2, W // 2, 3) and low.dtype == torch.float32
        img = torch.empty((B, H, W, 3), dtype=torch.float32, device=x.device)
        dev = isinstance(alpha, torch.Tensor)
        N.check(N.lib().sgx_rgb_out_fade(N.ptr(x), N.ptr(w), sj, sc, float(wscale), N.ptr(None if bias is None else _c(bias.detach())), N.ptr(low),
                                         0.0 if dev else float(alpha), 0.0 if dev else float(1 - alpha), alpha.data_ptr() if dev else None,
                                         N.ptr(img), B, H, W, C, N.dt(x), N.stream()), "sgx_rgb_out_fade")
        ctx.has_bias, ctx.wscale, ctx.alpha = bias is not None, float(wscale), alpha
        ctx.save_for_backward(x, weight)
        return img

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _c(g)
        al = ctx.alpha
        gx = gw = gb = glow = None
        if isinstance(al, torch.Tensor):                              # coefficients on the device: two scaling passes, then as below
            ga, a_s = _bcall(ScaleDevFn, g, al[0:1]), 1.0
            gl, b_s = (_bcall(ScaleDevFn, g, al[1:2]) if ctx.needs_input_grad[4] else None), 1.0
        else:
            ga, a_s, gl, b_s = g, float(al), g, float(1 - al)
        if ctx.needs_input_grad[0]:
            gx = _bcall(RgbInFn, ga, weight, None, ctx.wscale * a_s, x.dtype)
        if not _DATA_GRAD_ONLY:
            if ctx.needs_input_grad[1]:
                gw = _bcall(RgbWgradFn, ga, x, weight, ctx.wscale * a_s)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = _bcall(ColSumFn, ga, a_s)                         # 3 numbers
        if ctx.needs_input_grad[4]:
            glow = _bcall(Pool2Fn, gl, b_s)                            # adjoint of the nearest upsample, times (1 - alpha)
        return gx, gw, gb, None, glow, None


class RgbWgradFn(Function):
    """dW(j,c) = wscale * sum_p img[p][j] * f[p][c], written in the parameter's layout.  First order."""

    @staticmethod
    def forward(ctx, img, f, weight, wscale):
        img, f = _c(img), _c(f)
        sj, sc, C = rgb_layout(weight)
        npix = img.numel() // 3
        dw = torch.empty_like(weight, dtype=torch.float32, memory_format=torch.contiguous_format)
        L = N.lib()
        ws = N.workspace(L.sgx_rgb_wgrad_ws_bytes(npix, C), f.device)
        N.check(L.sgx_rgb_wgrad(N.ptr(img), N.ptr(f), N.ptr(dw), sj, sc, float(wscale), N.ptr(ws), ws.numel(), npix, C, N.dt(f),
                                N.stream()), "sgx_rgb_wgrad")
        return dw

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        raise NotImplementedError("second derivative through a weight gradient is not part of the training path")


# ---------------------------------------------------------------------------------------------------
class GEpilogueFn(Function):
    """noise + LeakyReLU + InstanceNorm + StyleMod (reference LayerEpilogue), conv bias folded in.  First order.
    ``flags``: EPI_ACT | EPI_NORM select the activation / instance-norm stages (default both); ``noise``/``nw`` None = no
    noise stage, ``style`` None = no style stage (exactly: zero noise weight / zero style)."""

    @staticmethod
    def forward(ctx, x, bias, noise, nw, style, flags=N.EPI_ACT | N.EPI_NORM, pre=None):
        """``pre``: [B, npart, C, 2] float64 partial statistics already produced by the kernel that wrote x (``BlurStatsFn``,
        ``ConvFn`` with ``stats``): the statistics pass over x is skipped."""
        x = _c(x)
        B, H, W, C = x.shape
        ctx.has_noise, ctx.has_style = nw is not None, style is not None
        if nw is None:
            noise = torch.zeros((B, H * W), dtype=torch.float32, device=x.device)
            nw_c = torch.zeros((C,), dtype=torch.float32, device=x.device)
        else:
            noise = _c(noise.detach().reshape(B, H * W))
            if noise.dtype != torch.float32:
                noise = noise.float()
            nw_c = _c(nw.detach())
        style_c = _c(style.detach()) if style is not None else torch.zeros((B, 2 * C), dtype=torch.float32, device=x.device)
        bias_c = None if bias is None else _c(bias.detach())
        y = torch.empty_like(x)
        mean = torch.empty((B, C), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        L = N.lib()
        ws = N.workspace(L.sgx_gepi_ws_bytes(B, H * W, C), x.device)
        if pre is not None and (pre.dtype != torch.float64 or pre.dim() != 4 or pre.shape[0] != B or pre.shape[2] != C):
            raise N.SgxError("GEpilogueFn: producer statistics must be float64 [B, npart, C, 2]")
        N.check(L.sgx_gepi_fwd(N.ptr(x), N.ptr(bias_c), N.ptr(noise), N.ptr(nw_c), N.ptr(style_c), N.ptr(y), N.ptr(mean), N.ptr(rstd),
                               N.ptr(ws), ws.numel(), N.ptr(pre), 0 if pre is None else pre.shape[1], B, H * W, C, int(flags), N.dt(x),
                               N.stream()), "sgx_gepi_fwd")
        ctx.has_bias, ctx.flags = bias is not None, int(flags)
        ctx.save_for_backward(x, bias_c, noise, nw_c, style_c, mean, rstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, bias_c, noise, nw_c, style_c, mean, rstd = ctx.saved_tensors
        gy = _c(gy)
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        dstyle = torch.empty((B, 2 * C), dtype=torch.float32, device=x.device)
        dnw = torch.empty((C,), dtype=torch.float32, device=x.device)
        dbias = torch.empty((C,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
        L = N.lib()
        ws = N.workspace(L.sgx_gepi_ws_bytes(B, H * W, C), x.device)
        N.check(L.sgx_gepi_bwd(N.ptr(gy), N.ptr(x), N.ptr(bias_c), N.ptr(noise), N.ptr(nw_c), N.ptr(style_c), N.ptr(mean), N.ptr(rstd),
                               N.ptr(dx), N.ptr(dstyle), N.ptr(dnw), N.ptr(dbias), N.ptr(ws), ws.numel(), B, H * W, C, ctx.flags,
                               N.dt(x), N.stream()), "sgx_gepi_bwd")
        return dx, dbias, None, dnw if ctx.has_noise else None, dstyle if ctx.has_style else None, None, None


FUSE_EPI_RGB = os.environ.get("SGX_FUSE_EPI_RGB", "1") != "0"    # A/B: the last generator epilogue inside to_rgb (EpiRgbOutFn)


def epi_rgb_out_ok(C, dtype):
    """The widths ``sgx_rgb_out_epi`` / ``sgx_rgb_wgrad_epi`` take (the checks at the top of those entry points): whole 16-byte channel
    vectors, min(vectors per pixel, 16) a power of two, at most 2048 channels.  Other widths (non-default ``fmap`` settings: C = 24,
    48 ...) run the epilogue and to_rgb as separate passes instead of raising."""
    ve = 4 if dtype == torch.float32 else 8
    if dtype not in (torch.float32, torch.bfloat16) or C % ve or C > 2048:
        return False
    lpp = min(C // ve, 16)
    return lpp & (lpp - 1) == 0


class EpiRgbOutFn(Function):
    """img = alpha * to_rgb(epilogue(y)) + (1 - alpha) * nearest_up2(low): the LAST LayerEpilogue of the synthesis network (noise,
    LeakyReLU, InstanceNorm, StyleMod -- reference models/CustomLayers.py:219-248) applied on the fly inside the 1x1 to_rgb convolution
    that is its only consumer (models/GAN.py:199-202), in one pass over the convolution's output y: the epilogue's output is never
    written.  ``low`` None: plain to_rgb(epilogue(y)).  ``pre``: producer statistics as in ``GEpilogueFn``.  First order (generator
    only).  Backward: d x2 = to_rgb^T(g) (``sgx_rgb_in``), the epilogue's own backward on y (``sgx_gepi_bwd``), and to_rgb's weight /
    bias gradient with x2 recomputed from y (``sgx_rgb_wgrad_epi``)."""

    @staticmethod
    def forward(ctx, y, ebias, noise, nw, style, weight, rbias, wscale, low, alpha, pre=None):
        y = _c(y)
        B, H, W, C = y.shape
        L = N.lib()
        noise_c = _c(noise.detach().reshape(B, H * W))
        if noise_c.dtype != torch.float32:
            noise_c = noise_c.float()
        nw_c, style_c = _c(nw.detach()), _c(style.detach())
        ebias_c = None if ebias is None else _c(ebias.detach())
        rbias_c = None if rbias is None else _c(rbias.detach())
        w = _c(weight.detach())
        sj, sc, Cw = rgb_layout(w)
        if Cw != C or w.shape[0] != 3 or tuple(style_c.shape) != (B, 2 * C):
            raise N.SgxError("EpiRgbOutFn: to_rgb weight [3,C,1,1] and style [B,2C] expected")
        mean = torch.empty((B, C), dtype=torch.float32, device=y.device)
        rstd = torch.empty_like(mean)
        wsb = L.sgx_gepi_ws_bytes(B, H * W, C)
        ws = N.workspace(wsb, y.device)
        if pre is not None and (pre.dtype != torch.float64 or pre.dim() != 4 or pre.shape[0] != B or pre.shape[2] != C):
            raise N.SgxError("EpiRgbOutFn: producer statistics must be float64 [B, npart, C, 2]")
        flags = N.EPI_ACT | N.EPI_NORM
        N.check(L.sgx_gepi_stats(N.ptr(y), N.ptr(ebias_c), N.ptr(noise_c), N.ptr(nw_c), N.ptr(mean), N.ptr(rstd), N.ptr(ws), wsb, N.ptr(pre),
                                 0 if pre is None else pre.shape[1], B, H * W, C, flags, N.dt(y), N.stream()), "sgx_gepi_stats")
        ab = None                                            # [alpha, 1 - alpha] in device memory (graph replay)
        if low is not None:
            low = _c(low)
            if tuple(low.shape) != (B, H // 2, W // 2, 3) or low.dtype != torch.float32:
                raise N.SgxError("EpiRgbOutFn: low-resolution image [B,H/2,W/2,3] fp32 expected")
            if isinstance(alpha, torch.Tensor):
                ab, a, b = alpha, 1.0, 1.0
            else:
                a, b = float(alpha), float(1 - alpha)
        else:
            a, b = 1.0, 0.0
        img = torch.empty((B, H, W, 3), dtype=torch.float32, device=y.device)
        N.check(L.sgx_rgb_out_epi(N.ptr(y), N.ptr(ebias_c), N.ptr(noise_c), N.ptr(nw_c), N.ptr(style_c), N.ptr(mean), N.ptr(rstd), N.ptr(w), sj, sc,
                                  float(wscale), N.ptr(rbias_c), N.ptr(low), a, b, None if ab is None else ab.data_ptr(), N.ptr(img), B, H, W, C,
                                  N.dt(y), N.stream()), "sgx_rgb_out_epi")
        ctx.ab = ab
        ctx.cfg = (float(wscale), a, b, ebias is not None, rbias is not None, low is not None, flags)
        ctx.save_for_backward(y, ebias_c, noise_c, nw_c, style_c, mean, rstd, weight)
        return img

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        y, ebias_c, noise_c, nw_c, style_c, mean, rstd, weight = ctx.saved_tensors
        wscale, a, b, has_ebias, has_rbias, has_low, flags = ctx.cfg
        g = _c(g)
        glow_src = g
        if ctx.ab is not None:                               # coefficients on the device: two scaling passes over the image gradient, then as below
            glow_src = _bcall(ScaleDevFn, g, ctx.ab[1:2]) if (has_low and ctx.needs_input_grad[8]) else None
            g = _bcall(ScaleDevFn, g, ctx.ab[0:1])
        B, H, W, C = y.shape
        L = N.lib()
        w = _c(weight.detach())
        sj, sc, _ = rgb_layout(w)
        dy = dbias = dnw = dstyle = None
        if any(ctx.needs_input_grad[k] for k in (0, 1, 3, 4)):
            gx2 = torch.empty_like(y)                                    # d x2 = alpha * wscale * W^T g
            N.check(L.sgx_rgb_in(N.ptr(g), N.ptr(w), sj, sc, wscale * a, None, N.ptr(gx2), B * H * W, C, _dtype_code(y.dtype), N.stream()), "sgx_rgb_in")
            dy = torch.empty_like(y)
            dstyle = torch.empty((B, 2 * C), dtype=torch.float32, device=y.device)
            dnw = torch.empty((C,), dtype=torch.float32, device=y.device)
            dbias = torch.empty((C,), dtype=torch.float32, device=y.device) if has_ebias else None
            wsb = L.sgx_gepi_ws_bytes(B, H * W, C)
            ws = N.workspace(wsb, y.device)
            N.check(L.sgx_gepi_bwd(N.ptr(gx2), N.ptr(y), N.ptr(ebias_c), N.ptr(noise_c), N.ptr(nw_c), N.ptr(style_c), N.ptr(mean), N.ptr(rstd), N.ptr(dy),
                                   N.ptr(dstyle), N.ptr(dnw), N.ptr(dbias), N.ptr(ws), wsb, B, H * W, C, flags, N.dt(y), N.stream()), "sgx_gepi_bwd")
        gw = gb = None
        if ctx.needs_input_grad[5] or (has_rbias and ctx.needs_input_grad[6]):
            gw = torch.empty_like(weight, dtype=torch.float32, memory_format=torch.contiguous_format)
            gb = torch.empty((3,), dtype=torch.float32, device=y.device) if has_rbias else None
            wsb = L.sgx_rgb_wgrad_epi_ws_bytes(B, H * W, C)
            ws = N.workspace(wsb, y.device)
            N.check(L.sgx_rgb_wgrad_epi(N.ptr(y), N.ptr(g), N.ptr(ebias_c), N.ptr(noise_c), N.ptr(nw_c), N.ptr(style_c), N.ptr(mean), N.ptr(rstd), N.ptr(gw),
                                        N.ptr(gb), sj, sc, wscale * a, a, N.ptr(ws), wsb, B, H * W, C, N.dt(y), N.stream()), "sgx_rgb_wgrad_epi")
        glow = None
        if has_low and ctx.needs_input_grad[8]:
            glow = _bcall(Pool2Fn, glow_src, b)                          # adjoint of the nearest upsample, times (1 - alpha)
        return dy, dbias, None, dnw, dstyle, gw, gb, None, glow, None, None


class PixelNormFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x.float())
        y = torch.empty_like(x)
        N.check(N.lib().sgx_pixelnorm_fwd(N.ptr(x), N.ptr(y), x.shape[0], x.shape[1], N.stream()), "sgx_pixelnorm_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _c(g)
        dx = torch.empty_like(x)
        N.check(N.lib().sgx_pixelnorm_bwd(N.ptr(g), N.ptr(x), N.ptr(dx), x.shape[0], x.shape[1], N.stream()), "sgx_pixelnorm_bwd")
        return dx


class MbstdFn(Function):
    """Minibatch stddev: [B,H,W,C] -> [B,H,W,Cpad] (channel C = group statistic, the rest zero padding)."""

    @staticmethod
    def forward(ctx, x, cpad):
        x = _c(x)
        B, H, W, C = x.shape
        y = torch.empty((B, H, W, cpad), dtype=x.dtype, device=x.device)
        N.check(N.lib().sgx_mbstd_fwd(N.ptr(x), N.ptr(y), B, H * W, C, cpad, N.dt(x), N.stream()), "sgx_mbstd_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return _bcall(MbstdBwdFn, gy, x), None


class MbstdBwdFn(Function):
    @staticmethod
    def forward(ctx, gy, x):
        gy = _c(gy)
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        N.check(N.lib().sgx_mbstd_bwd(N.ptr(gy), N.ptr(x), N.ptr(dx), B, H * W, C, gy.shape[3], N.dt(x), N.stream()), "sgx_mbstd_bwd")
        ctx.save_for_backward(gy, x)
        return dx

    @staticmethod
    @once_differentiable
    def backward(ctx, ggx):
        gy, x = ctx.saved_tensors
        ggx = _c(ggx)
        B, H, W, C = x.shape
        ddy = torch.empty_like(gy)
        gx = torch.empty_like(x)
        N.check(N.lib().sgx_mbstd_bwd2(N.ptr(ggx), N.ptr(gy), N.ptr(x), N.ptr(ddy), N.ptr(gx), B, H * W, C, gy.shape[3], N.dt(x),
                                       N.stream()), "sgx_mbstd_bwd2")
        return ddy, gx


def _dense_view(x):
    """A contiguous view of x's storage (NHWC-stored images arrive as permuted NCHW views), plus how to undo it."""
    if x.is_contiguous():
        return x, None
    if x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous():
        return x.permute(0, 2, 3, 1), (0, 3, 1, 2)
    return x.contiguous(), None


class SumSqFn(Function):
    """sum(x*x) -> fp32 scalar tensor (the R1 penalty head).  First order."""

    @staticmethod
    def forward(ctx, x):
        xv, back = _dense_view(x.detach())
        if xv.dtype != torch.float32:
            raise N.SgxError("SumSqFn expects fp32 (image gradients)")
        L = N.lib()
        out = torch.empty((), dtype=torch.float32, device=x.device)
        ws = N.workspace(L.sgx_sumsq_ws_bytes(), x.device)
        N.check(L.sgx_sumsq_f32(N.ptr(xv), xv.numel(), N.ptr(ws), ws.numel(), N.ptr(out), N.stream()), "sgx_sumsq_f32")
        ctx.back = back
        ctx.save_for_backward(xv)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (xv,) = ctx.saved_tensors
        gx = torch.empty_like(xv)
        N.check(N.lib().sgx_scale_dev_f32(N.ptr(xv), N.ptr(_c(g.float())), 2.0, N.ptr(gx), xv.numel(), N.stream()), "sgx_scale_dev_f32")
        return gx if ctx.back is None else gx.permute(*ctx.back)


class LogisticLossFn(Function):
    """The logistic loss heads (reference models/Losses.py:213-229) as ONE launch forward -- the loss scalar and its derivative w.r.t.
    every logit -- and one scaling launch per logit tensor backward, instead of ~9 elementwise / reduction kernels each way.
    ``generator``: mean softplus(-fake) * scale; else (mean softplus(fake) + mean softplus(-real)) * scale.  First order only (the R1
    penalty differentiates the discriminator, not this head)."""

    @staticmethod
    def forward(ctx, fake, real, scale, generator):
        f = _c(fake.detach().reshape(-1))
        r = None if real is None else _c(real.detach().reshape(-1))
        if f.dtype != torch.float32 or (r is not None and r.dtype != torch.float32):
            raise N.SgxError("LogisticLossFn: fp32 logits expected")
        loss = torch.empty((), dtype=torch.float32, device=f.device)
        gf = torch.empty_like(f)
        gr = None if r is None else torch.empty_like(r)
        N.check(N.lib().sgx_logistic_loss(N.ptr(f), f.numel(), N.ptr(r), 0 if r is None else r.numel(), float(scale), int(bool(generator)),
                                          N.ptr(loss), N.ptr(gf), N.ptr(gr), N.stream()), "sgx_logistic_loss")
        ctx.save_for_backward(gf, gr)
        ctx.shapes = (fake.shape, None if real is None else real.shape)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        gf, gr = ctx.saved_tensors
        g = _c(g.float())
        out = []
        for t, shape in zip((gf, gr), ctx.shapes):
            if t is None:
                out.append(None)
                continue
            o = torch.empty_like(t)
            N.check(N.lib().sgx_scale_dev_f32(N.ptr(t), N.ptr(g), 1.0, N.ptr(o), t.numel(), N.stream()), "sgx_scale_dev_f32")
            out.append(o.view(shape))
        return out[0], out[1], None, None


class MatMulFn(Function):
    """C = alpha * op(A) @ op(B), fp32 row-major.  ta/tb as in sgx_gemm_f32.  Closed under differentiation."""

    @staticmethod
    def forward(ctx, A, Bm, ta, tb, alpha):
        A, Bm = _c(A), _c(Bm)
        assert A.dtype == torch.float32 and Bm.dtype == torch.float32
        M, K = (A.shape[1], A.shape[0]) if ta else (A.shape[0], A.shape[1])
        K2, Nn = (Bm.shape[1], Bm.shape[0]) if tb else (Bm.shape[0], Bm.shape[1])
        assert K == K2, (A.shape, Bm.shape, ta, tb)
        C = torch.empty((M, Nn), dtype=torch.float32, device=A.device)
        wsb = N.lib().sgx_gemm_ws_bytes(M, Nn, K)
        ws = N.workspace(wsb, A.device) if wsb else None
        N.check(N.lib().sgx_gemm_f32(N.ptr(A), N.ptr(Bm), N.ptr(C), M, Nn, K, int(ta), int(tb), float(alpha), N.ptr(ws), wsb, N.stream()), "sgx_gemm_f32")
        ctx.ta, ctx.tb, ctx.alpha = int(ta), int(tb), float(alpha)
        ctx.save_for_backward(A, Bm)
        return C

    @staticmethod
    def backward(ctx, gC):
        A, Bm = ctx.saved_tensors
        ta, tb, al = ctx.ta, ctx.tb, ctx.alpha
        gA = gB = None
        if ctx.needs_input_grad[0]:
            gA = _bcall(MatMulFn, gC, Bm, 0, 1 - tb, al) if not ta else _bcall(MatMulFn, Bm, gC, tb, 1, al)
        if ctx.needs_input_grad[1] and not _DATA_GRAD_ONLY:
            gB = _bcall(MatMulFn, A, gC, 1 - ta, 0, al) if not tb else _bcall(MatMulFn, gC, A, 1, ta, al)
        return gA, gB, None, None, None


# ---------------------------------------------------------------------------------------------------
class LinearFn(Function):
    """EqualizedLinear with everything fused: one launch forward (GEMM + bias + LeakyReLU), two backward (data gradient;
    weight gradient + bias gradient), the activation backward folded into their operand loads.  First order only: used
    by the generator's mapping network and style affines, which no double backward reaches (the discriminator's dense
    layers sit under R1 and keep the differentiable composite ``linear``)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_mul, b_mul, act):
        x, weight = _c(x), _c(weight)
        assert x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2
        B, K = x.shape
        Nn = weight.shape[0]
        assert weight.shape[1] == K
        y = torch.empty((B, Nn), dtype=torch.float32, device=x.device)
        N.check(N.lib().sgx_linear_fwd(N.ptr(x), N.ptr(weight), None if bias is None else N.ptr(_c(bias.detach())), N.ptr(y), B, Nn, K,
                                       float(w_mul), float(b_mul), int(act), N.stream()), "sgx_linear_fwd")
        ctx.cfg = (float(w_mul), float(b_mul), int(act), bias is not None)
        ctx.save_for_backward(x, weight, y if act else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        w_mul, b_mul, act, has_bias = ctx.cfg
        gy = _c(gy)
        B, K = x.shape
        Nn = weight.shape[0]
        L = N.lib()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            N.check(L.sgx_linear_bwd_data(N.ptr(gy), N.ptr(y), N.ptr(weight), N.ptr(gx), B, Nn, K, w_mul, N.stream()), "sgx_linear_bwd_data")
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(weight)
            gb = torch.empty((Nn,), dtype=torch.float32, device=x.device) if has_bias else None
            N.check(L.sgx_linear_bwd_param(N.ptr(gy), N.ptr(y), N.ptr(x), N.ptr(gw), N.ptr(gb), B, Nn, K, w_mul, b_mul, N.stream()),
                    "sgx_linear_bwd_param")
        return gx, gw, gb, None, None, None


def linear_fused(x, weight, bias, w_mul, b_mul, act=N.ACT_NONE):
    return call(LinearFn, x, weight, bias, float(w_mul), float(b_mul), int(act))


class SplitLayersFn(Function):
    """dlatents [B, L, D] -> L contiguous [B, D] tensors (one transposing copy); backward: one stack.  Replaces L strided
    slices (a copy kernel each) whose autograd backward is L zero-fills + L scatters + L-1 adds."""

    @staticmethod
    def forward(ctx, dl):
        lm = dl.transpose(0, 1).contiguous()                      # [L, B, D]
        ctx.L = lm.shape[0]
        return tuple(lm.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        ref = next(g for g in grads if g is not None)
        return torch.stack([g if g is not None else torch.zeros_like(ref) for g in grads], dim=1)


class PreStyle:
    """A style vector [B, 2C] already computed (by ``GroupedStyleFn``), handed to a LayerEpilogue in place of its dlatent."""
    __slots__ = ("style",)

    def __init__(self, style):
        self.style = style


_STYLE_TABLES = {}


class GroupedStyleFn(Function):
    """All style affines of a generator forward in one launch (two for the backward): (lm [L,B,D], meta, *weights, *biases)
    -> one [B, N_g] tensor per group.  ``meta`` = tuple of (layer index, w_mul, b_mul) per group.  First order only."""

    @staticmethod
    def forward(ctx, lm, meta, *params):
        G = len(meta)
        ws, bs = params[:G], params[G:]
        lm = _c(lm)
        L_, B, D = lm.shape
        key = (tuple(w.data_ptr() for w in ws), tuple(b.data_ptr() for b in bs), meta, B)
        ent = _STYLE_TABLES.get(key)
        if ent is None:                                  # never evicted: step graphs hold the device address of the table
            rows, yoff, tile = [], 0, 0
            for (layer, w_mul, b_mul), w, b in zip(meta, ws, bs):
                n = w.shape[0]
                assert w.shape[1] == D and w.is_contiguous() and w.dtype == torch.float32
                rows.append([w.data_ptr(), b.data_ptr(), n, yoff, tile, int(np.float32(w_mul).view(np.uint32)),
                             int(np.float32(b_mul).view(np.uint32)), layer])
                yoff += B * n; tile += (n + 15) // 16
            table, _ = N.upload(torch.tensor(rows, dtype=torch.int64), lm.device)
            ent = _STYLE_TABLES[key] = (table, [r[2] for r in rows], [r[3] for r in rows], yoff, tile)
        table, ns, yoffs, ytotal, tiles = ent
        y = torch.empty(ytotal, dtype=torch.float32, device=lm.device)
        N.check(N.lib().sgx_style_fwd(N.ptr(lm), N.ptr(table), N.ptr(y), G, B, D, tiles, N.stream()), "sgx_style_fwd")
        ctx.ent, ctx.G, ctx.shape = ent, G, (L_, B, D)
        ctx.save_for_backward(lm)
        return tuple(y[o:o + B * n].view(B, n) for o, n in zip(yoffs, ns))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gys):
        (lm,) = ctx.saved_tensors
        table, ns, yoffs, ytotal, tiles = ctx.ent
        L_, B, D = ctx.shape
        G = ctx.G
        gy = torch.cat([(g if g is not None else torch.zeros(B, n, device=lm.device)).reshape(-1) for g, n in zip(gys, ns)])
        L = N.lib()
        glm = None
        if ctx.needs_input_grad[0]:
            glm = torch.zeros_like(lm) if G < L_ else torch.empty_like(lm)
            N.check(L.sgx_style_bwd_data(N.ptr(gy), N.ptr(table), N.ptr(glm), G, B, D, max(ns), N.stream()), "sgx_style_bwd_data")
        ntot = sum(ns)
        dw = torch.empty((ntot, D), dtype=torch.float32, device=lm.device)
        db = torch.empty((ntot,), dtype=torch.float32, device=lm.device)
        N.check(L.sgx_style_bwd_param(N.ptr(gy), N.ptr(lm), N.ptr(table), N.ptr(dw), N.ptr(db), G, B, D, tiles, N.stream()),
                "sgx_style_bwd_param")
        offs = [o // B for o in yoffs]
        gws = tuple(dw[o:o + n] for o, n in zip(offs, ns))
        gbs = tuple(db[o:o + n] for o, n in zip(offs, ns))
        return (glm, None) + gws + gbs


class NoiseArena:
    """One device randn per generator forward; the per-layer noise maps [B,1,H,W] are consecutive slices of it."""

    def __init__(self, numel, device):
        self.buf = torch.randn(int(numel), device=device, dtype=torch.float32)
        self.off = 0

    def take(self, b, h, w):
        n = b * h * w
        if self.off + n > self.buf.numel():
            return torch.randn(b, 1, h, w, device=self.buf.device, dtype=torch.float32)
        out = self.buf[self.off:self.off + n].view(b, 1, h, w)
        self.off += n
        return out


NOISE_ARENA = None                  # set by GSynthesis.forward for the duration of one forward


def linear(x, weight, bias, w_mul, b_mul, act=N.ACT_NONE):
    """EqualizedLinear: F.linear(x, W*w_mul, b*b_mul) (+ LeakyReLU).  x fp32 [B, in]; parameters read in place."""
    y = call(MatMulFn, x, weight, 0, 1, w_mul)
    if bias is None and act == N.ACT_NONE:
        return y
    return call(BiasActFn, y, bias, b_mul, act)


def nhwc(x_nchw, dtype=None):
    """Logical NCHW tensor (any strides) -> contiguous NHWC."""
    t = x_nchw.permute(0, 2, 3, 1)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def images_from_uint8(u8, flip=None, out_dtype=torch.float32, layout=None):
    """uint8 image batch on the GPU -> the ``real_batch`` the training step takes: logical [B,3,H,W] view over NHWC storage,
    (v/255 - 0.5)/0.5 -- the reference's ToTensor + Normalize (data/transforms.py:27-32) done on the device, so the batch
    crosses PCIe as bytes.  ``u8``: [B,H,W,3] (decoded images; ``layout='hwc'``) or [B,3,H,W] (``'chw'``; inferred when
    unambiguous).  ``flip``: per-image horizontal-flip decisions (RandomHorizontalFlip, drawn by the caller), or None."""
    if not (u8.is_cuda and u8.dtype == torch.uint8 and u8.dim() == 4):
        raise N.SgxError("images_from_uint8: need a 4-d uint8 GPU tensor")
    if layout is None:
        hwc, chw = u8.shape[3] == 3, u8.shape[1] == 3
        if hwc == chw:
            raise N.SgxError(f"images_from_uint8: cannot infer the layout of {tuple(u8.shape)}; pass layout='hwc' or 'chw'")
        layout = "hwc" if hwc else "chw"
    u8 = _c(u8)
    B, H, W = (u8.shape[0], u8.shape[1], u8.shape[2]) if layout == "hwc" else (u8.shape[0], u8.shape[2], u8.shape[3])
    fl = None
    if flip is not None:
        fl = torch.as_tensor(flip).to(device=u8.device, dtype=torch.int32).contiguous()
        if fl.numel() != B:
            raise N.SgxError("images_from_uint8: one flip decision per image")
    out = torch.empty((B, H, W, 3), dtype=out_dtype, device=u8.device)
    N.check(N.lib().sgx_images_u8_to_nhwc(N.ptr(u8), N.ptr(out), N.ptr(fl), B, H, W, int(layout == "chw"), N.dt(out), N.stream()),
            "sgx_images_u8_to_nhwc")
    return nchw_view(out)


def nchw_view(x_nhwc):
    """Contiguous NHWC -> logical NCHW view (channels_last strides, zero copy)."""
    return x_nhwc.permute(0, 3, 1, 2)
