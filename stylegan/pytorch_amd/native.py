"""ctypes binding of libsgx_hip.so (C ABI: include/sgx.h).

The reference has no FFI -- its seam is the nn.Module surface -- so this file *is* the binding a maintainer
would add.  Tensors are passed as raw device pointers + sizes; the stream is read from
``torch.cuda.current_stream()`` at call time (backward runs on autograd worker threads).

There is NO fallback: if the shared library is missing or a tensor is not on a GPU the call raises.
"""
import ctypes
import os
import subprocess
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGX_HIP_LIB") or os.path.join(_HERE, "libsgx_hip.so")   # env: kernel experiments only
CSRC = os.path.join(_HERE, "csrc")

F32, BF16 = 0, 1
ACT_NONE, ACT_LRELU, ACT_RELU = 0, 1, 2
EPI_ACT, EPI_NORM = 1, 2
PACK_S, PACK_D, PACK_U, PACK_UF = 0, 1, 2, 3

_lib = None
_lock = threading.Lock()

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
P, I, F, Z = c_void_p, c_int, c_float, c_size_t

# name -> (restype, argtypes); mirrors include/sgx.h one to one
SIGNATURES = {
    "sgx_version": (I, []),
    "sgx_last_error": (ctypes.c_char_p, []),
    "sgx_clear_error": (I, []),
    "sgx_stream_wait_stream": (I, [P, P]),
    "sgx_conv3x3": (I, [P, P, P, P, I, I, I, I, I, I, P, I, P]),
    "sgx_conv_variant": (I, [I, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "sgx_conv4x4s2_down": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
    "sgx_conv4x4s2_up": (I, [P, P, P, I, I, I, I, I, I, P]),
    "sgx_wgrad_ws_bytes": (Z, [I, I, I, I, I, I]),
    "sgx_images_u8_to_nhwc": (I, [P, P, P, I, I, I, I, I, P]),
    "sgx_selftest_tr16": (I, [P, P]),
    "sgx_conv_splitk_ws_bytes": (Z, [I, I, I, I, I, I, I]),
    "sgx_conv_splitk": (I, [I, P, P, P, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "sgx_conv_config": (I, [I, I, I, I, I, I, I, P]),
    "sgx_prof_start": (I, [I, I]),
    "sgx_prof_count": (I, []),
    "sgx_prof_get": (I, [I, P, I, P, P, P, P, I]),
    "sgx_pack_weight": (I, [P, P, P, I, I, I, I, F, I, P]),
    "sgx_pack_weight_multi": (I, [P, I, I, I, P]),
    "sgx_pack_weight_blocks": (I, [I, I]),
    "sgx_wgrad3x3_param": (I, [P, P, P, P, P, Z, I, I, I, I, I, I, F, I, I, I, I, P]),
    "sgx_wgrad4x4s2_param": (I, [P, P, P, P, P, Z, I, I, I, I, I, I, F, I, I, I, I, P]),
    "sgx_bias_act": (I, [P, P, F, P, Z, I, I, I, P]),
    "sgx_lrelu_bwd": (I, [P, P, P, Z, F, F, P, I, P]),
    "sgx_lrelu_bwd_bits": (I, [P, P, P, Z, F, F, P, I, P]),
    "sgx_conv4x4s2_down_fade_ok": (I, [I, I, I, I, I, I]),
    "sgx_conv4x4s2_down_fade": (I, [P, P, P, P, F, F, P, P, P, I, I, I, I, I, I, P]),
    "sgx_conv4x4s2_down_fade_rgb": (I, [P, P, P, P, P, F, P, F, F, F, F, P, P, P, I, I, I, I, I, I, P]),
    "sgx_fade_rgb_bwd_ws_bytes": (Z, [Z, I]),
    "sgx_fade_rgb_bwd": (I, [P, P, P, P, F, F, F, F, P, P, P, P, I, P, P, Z, Z, I, I, P]),
    "sgx_fade_rgb_bwd_finish": (I, [P, Z, Z, I, F, F, F, P, P, P, I, P]),
    "sgx_fade_rgb_bwd2": (I, [P, P, P, P, F, F, F, P, P, Z, I, I, P]),
    "sgx_axpby": (I, [P, P, P, F, F, Z, I, P]),
    "sgx_axpby_dev": (I, [P, P, P, P, P, Z, I, P]),
    "sgx_blur3x3": (I, [P, P, I, I, I, I, I, P]),
    "sgx_blur3x3_act": (I, [P, P, P, I, I, I, I, I, I, P]),
    "sgx_blur_kxk": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "sgx_pool2": (I, [P, P, I, I, I, I, F, I, P]),
    "sgx_up2": (I, [P, P, I, I, I, I, F, I, P]),
    "sgx_up2_add": (I, [P, P, P, I, I, I, I, F, I, P]),
    "sgx_colsum_ws_bytes": (Z, [Z, I]),
    "sgx_colsum": (I, [P, P, F, P, Z, Z, I, I, P]),
    "sgx_rgb_in": (I, [P, P, I, I, F, P, P, Z, I, I, P]),
    "sgx_rgb_in_add": (I, [P, P, I, I, F, P, P, Z, I, I, P]),
    "sgx_rgb_out": (I, [P, P, I, I, F, P, P, Z, I, I, P]),
    "sgx_downsample_fade_rgb": (I, [P, P, I, I, I, F, F, P, P]),
    "sgx_rgb_out_fade": (I, [P, P, I, I, F, P, P, F, F, P, P, I, I, I, I, I, P]),
    "sgx_rgb_wgrad_ws_bytes": (Z, [Z, I]),
    "sgx_rgb_wgrad": (I, [P, P, P, I, I, F, P, Z, Z, I, I, P]),
    "sgx_gepi_ws_bytes": (Z, [I, I, I]),
    "sgx_gepi_fwd": (I, [P, P, P, P, P, P, P, P, P, Z, P, I, I, I, I, I, I, P]),
    "sgx_gepi_stats": (I, [P, P, P, P, P, P, P, Z, P, I, I, I, I, I, I, P]),
    "sgx_rgb_out_epi": (I, [P, P, P, P, P, P, P, P, I, I, F, P, P, F, F, P, P, I, I, I, I, I, P]),
    "sgx_rgb_wgrad_epi_ws_bytes": (Z, [I, I, I]),
    "sgx_rgb_wgrad_epi": (I, [P, P, P, P, P, P, P, P, P, P, I, I, F, F, P, Z, I, I, I, I, P]),
    "sgx_blur3x3_stats_nparts": (I, [I, I, I, I, I]),
    "sgx_conv3x3_stats_nparts": (I, [I, I, I, I, I, I]),
    "sgx_conv3x3_signbits_ok": (I, [I, I, I, I, I, I]),
    "sgx_conv3x3_signbits": (I, [P, P, P, P, P, I, I, I, I, I, I, P, I, P]),
    "sgx_blur3x3_bits": (I, [P, P, P, I, I, I, I, I, I, P]),
    "sgx_conv4x4s2_up_blur_ok": (I, [I, I, I, I, I, I]),
    "sgx_conv4x4s2_up_blur": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "sgx_conv4x4s2_up_blur_bits": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "sgx_pack_upblur": (I, [P, P, I, I, I, I, F, P]),
    "sgx_conv_upblur_ok": (I, [I, I, I, I, I, I]),
    "sgx_conv_upblur": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "sgx_rgbconv_ok": (I, [I, I, I, I, I]),
    "sgx_rgbconv_pack": (I, [P, F, P, F, P, F, P, P, I, P]),
    "sgx_rgbconv_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "sgx_rgbconv_tune": (I, [I, I, I]),
    "sgx_rgbconv_dgrad": (I, [P, P, P, I, I, I, I, I, P]),
    "sgx_rgbconv_wgrad_ws_bytes": (Z, [I, I, I, I]),
    "sgx_rgbconv_wgrad": (I, [P, P, I, P, F, P, F, P, F, P, P, P, P, I, P, Z, I, I, I, I, I, P]),
    "sgx_conv3x3_stats": (I, [P, P, P, P, P, P, P, Z, I, I, I, I, I, I, P]),
    "sgx_blur3x3_stats": (I, [P, P, P, P, P, P, Z, I, I, I, I, I, I, P]),
    "sgx_gepi_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, Z, I, I, I, I, I, P]),
    "sgx_pixelnorm_fwd": (I, [P, P, I, I, P]),
    "sgx_pixelnorm_bwd": (I, [P, P, P, I, I, P]),
    "sgx_mbstd_fwd": (I, [P, P, I, I, I, I, I, P]),
    "sgx_mbstd_bwd": (I, [P, P, P, I, I, I, I, I, P]),
    "sgx_mbstd_bwd2": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "sgx_sumsq_ws_bytes": (Z, []),
    "sgx_sumsq_f32": (I, [P, Z, P, Z, P, P]),
    "sgx_scale_dev_f32": (I, [P, P, F, P, Z, P]),
    "sgx_logistic_loss": (I, [P, I, P, I, F, I, P, P, P, P]),
    "sgx_gemm_ws_bytes": (Z, [I, I, I]),
    "sgx_gemm_f32": (I, [P, P, P, I, I, I, I, I, F, P, Z, P]),
    "sgx_linear_fwd": (I, [P, P, P, P, I, I, I, F, F, I, P]),
    "sgx_linear_bwd_data": (I, [P, P, P, P, I, I, I, F, P]),
    "sgx_linear_bwd_param": (I, [P, P, P, P, P, I, I, I, F, F, P]),
    "sgx_style_fwd": (I, [P, P, P, I, I, I, I, P]),
    "sgx_style_bwd_data": (I, [P, P, P, I, I, I, I, P]),
    "sgx_style_bwd_param": (I, [P, P, P, P, P, I, I, I, I, P]),
    "sgx_adam_multi": (I, [P, P, P, P, P, I, F, F, F, P, P, P, P]),
    "sgx_ema_multi": (I, [P, P, P, I, F, P]),
    "sgx_gradnorm_clip_coef": (I, [P, P, I, F, P, P, P]),
}


def build(verbose: bool = False) -> str:
    """Compile the gfx950 kernels in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libsgx_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


def lib():
    """Load libsgx_hip.so (after torch, so the HIP runtime torch already loaded is the one it binds to)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                       "(there is no non-HIP fallback for this path)")
                l = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(l, name)
                    fn.restype, fn.argtypes = res, args
                _lib = l
    return _lib


class SgxError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().sgx_last_error().decode(errors="replace")
        raise SgxError(f"{what} failed (code {rc}): {msg}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def stream() -> int:
    """hipStream_t of torch's current stream on the current device (the raw getter: this is called once per launch)."""
    if _raw_stream is not None:
        return _raw_stream(_get_device())                   # torch.cuda.current_device() minus its lazy-init check
    return torch.cuda.current_stream().cuda_stream


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SgxError("libsgx_hip kernels need GPU tensors (no CPU fallback exists for this path)")
    if not t.is_contiguous():
        raise SgxError("internal: non-contiguous tensor passed to a kernel")
    return t.data_ptr()


# Host -> device uploads of small descriptor tables: pinned staging + non_blocking, so the copy is stream ordered and the
# host never waits for the GPU queue.  While a hipGraph is being captured the copy becomes a graph node that re-reads the
# pinned buffer at every replay, so the buffer is kept alive (and may be rewritten in place before a replay).
_ARENA = []                     # [pinned uint8 chunk, bytes used]: staging for copies captured into hipGraphs


def capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def reserve_capture_staging(nbytes: int = 1 << 20):
    """Make sure ``nbytes`` of pinned staging exist BEFORE a capture starts: allocating pinned memory while a stream is
    capturing (hipHostMalloc synchronises) invalidates the capture -- seen as a timing-dependent capture failure when
    torch's pinned-block cache happened to be empty."""
    if capturing():
        return
    if not _ARENA or _ARENA[-1][0].numel() - _ARENA[-1][1] < nbytes:
        _ARENA.append([torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory(), 0])


def _arena_take(nbytes: int) -> torch.Tensor:
    nbytes = (nbytes + 63) // 64 * 64
    if not _ARENA or _ARENA[-1][0].numel() - _ARENA[-1][1] < nbytes:
        if capturing():
            raise SgxError("pinned staging exhausted during hipGraph capture (reserve_capture_staging() before capturing)")
        reserve_capture_staging(nbytes)
    chunk = _ARENA[-1]
    out = chunk[0][chunk[1]:chunk[1] + nbytes]
    chunk[1] += nbytes
    return out


class PinnedRing:
    """Small pinned staging slots allocated ONCE and reused round-robin (descriptor-table uploads, the two loss scalars of a
    step).  torch's pinned cache hands a freed block out again only after the GPU has passed the copy that used it, so a host
    that runs AHEAD of the GPU (the point of the deferred losses) makes it call hipHostMalloc -- which synchronises -- several
    times per step: measured as a timed region 1.1 ms per step slower than the 4-step calibration of the same mode.  A slot
    is reused after ``slots`` later requests; its event (recorded after the copy that used it) is waited for first, and a
    still-unread owner of the slot (a DeferredLoss) is resolved before the slot is overwritten."""

    def __init__(self, slots=512, slot_bytes=8192):
        self.slots, self.slot_bytes = slots, slot_bytes
        self.buf = None
        self.next = 0
        self.lock = threading.Lock()                                 # (backward passes run on autograd's worker threads)

    def take(self, nbytes):
        """-> (pinned uint8 view of ``nbytes``, slot index), or (None, -1) when the request does not fit a slot."""
        if nbytes > self.slot_bytes:
            return None, -1
        with self.lock:
            return self._take(nbytes)

    def _take(self, nbytes):
        if self.buf is None:
            self.buf = torch.empty(self.slots * self.slot_bytes, dtype=torch.uint8).pin_memory()
            self.events = [None] * self.slots
            self.owners = [None] * self.slots
        i = self.next
        self.next = (i + 1) % self.slots
        own = self.owners[i]
        if own is not None:
            o = own()
            if o is not None:
                o.item()                                            # read the value before the slot is overwritten
            self.owners[i] = None
        ev = self.events[i]
        if ev is not None:
            ev[0].synchronize()                                     # (long done: ``slots`` requests ago)
        return self.buf[i * self.slot_bytes:i * self.slot_bytes + nbytes], i

    def mark(self, i, owner=None):
        """Record the slot's event on the current stream (after the copy that uses the slot) -> the event."""
        # (a torch Event is bound to the device of its first record(): a process that drives several GPUs gets a fresh event when
        # the slot is next used from another device; the old one was synchronised by _take before the slot was handed out)
        dev = _get_device()
        ent = self.events[i]
        if ent is None or ent[1] != dev:
            ent = self.events[i] = (torch.cuda.Event(), dev)
        ev = ent[0]
        ev.record()
        if owner is not None:
            import weakref
            self.owners[i] = weakref.ref(owner)
        return ev


RING = PinnedRing()


def _host_copy(dst: torch.Tensor, src: torch.Tensor):
    """dst.copy_(src) for two host tensors as ONE memcpy on this thread.  torch's CPU copy goes parallel above 32768 elements: on
    the 256-core hosts of the GPU boxes waking that thread pool cost 5-20 ms per 256 KB copy (the style-mixing latents of a batch
    of 128), i.e. 60-90 ms per iteration of a step whose kernels take 4 ms (profiles/r04_sweep_host_copy.txt)."""
    if src.device.type == "cpu" and src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype and not src.requires_grad:
        import numpy as np
        try:
            np.copyto(dst.numpy(), src.numpy())
            return
        except (TypeError, RuntimeError):                    # (a dtype numpy does not have: bfloat16)
            pass
    dst.copy_(src)


_BIG = {}                       # size class (power of two) -> {"slots": [[pinned buffer, (event, device) or None], ...], "next": i}
_BIG_SLOTS = 4


def _big_slot(nbytes: int):
    """A pinned staging buffer of at least ``nbytes`` for uploads that do not fit a ring slot (the style-mixing latents of a big
    batch: 256 KB at batch 128): ``_BIG_SLOTS`` buffers per power-of-two size class, allocated once and reused round-robin after
    waiting for the event of the copy that last used them.  ``Tensor.pin_memory()`` per upload instead costs a hipHostMalloc each
    time the host runs ahead of the GPU (torch's pinned cache only recycles a block the GPU is done with): measured 21 ms per
    call at batch 128, i.e. 87 ms per iteration of a 4 ms step (profiles/r04_sweep_pin_memory.txt)."""
    key = 1 << max(13, (int(nbytes) - 1).bit_length())
    with _lock:
        ring = _BIG.get(key)
        if ring is None:
            # all slots of a size class at its first use (a hipHostMalloc costs 10-40 ms: paid once, in the first iteration, not
            # spread over the first _BIG_SLOTS uploads)
            ring = _BIG[key] = {"slots": [[torch.empty(key, dtype=torch.uint8).pin_memory(), None] for _ in range(_BIG_SLOTS)], "next": 0}
        ent = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % _BIG_SLOTS
    if ent[1] is not None:
        ent[1][0].synchronize()                            # (_BIG_SLOTS uploads of this size ago)
    return ent


def upload(t: torch.Tensor, device):
    """-> (device tensor, pinned staging tensor).  Outside a capture: a slot of the pinned ring (one of a few persistent pinned
    buffers of its size class if it is too big for a slot).  Inside: a slice of the pre-reserved arena that lives as long as the
    process (the graph re-reads it at every replay)."""
    nbytes = t.numel() * t.element_size()
    if capturing():
        pinned = _arena_take(nbytes)[:nbytes].view(t.dtype).view(t.shape)
        pinned.copy_(t)
    else:
        view, slot = RING.take(nbytes)
        if view is not None:
            pinned = view.view(t.dtype).view(t.shape)
            _host_copy(pinned, t)
            dev = pinned.to(device, non_blocking=True)
            RING.mark(slot)
            return dev, pinned
        ent = _big_slot(nbytes)
        pinned = ent[0][:nbytes].view(t.dtype).view(t.shape)
        _host_copy(pinned, t)
        dev = pinned.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ent[1] = (ev, _get_device())
        return dev, pinned
    return pinned.to(device, non_blocking=True), pinned


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ---------------------------------------------------------------------------------------------------
# per-launch profiler of the library (HIP events on the launch stream, recorded inside libsgx_hip.so)
def prof_start(mode: int = 1, only_of: int = 0) -> None:
    """mode 1: record every kernel launch; 2: only launches of the kernel record ``only_of`` belongs to; 0: stop."""
    check(lib().sgx_prof_start(int(mode), int(only_of)), "sgx_prof_start")


def prof_records():
    """[(kernel name as rocprofv3 prints it, milliseconds, flops, algorithmic bytes, layer description), ...]."""
    L = lib()
    name, desc = ctypes.create_string_buffer(256), ctypes.create_string_buffer(64)
    ms, fl, by = ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
    out = []
    for i in range(L.sgx_prof_count()):
        check(L.sgx_prof_get(i, ctypes.addressof(name), 256, ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(by),
                             ctypes.addressof(desc), 64), "sgx_prof_get")
        out.append((name.value.decode(), ms.value, fl.value, by.value, desc.value.decode()))
    return out
