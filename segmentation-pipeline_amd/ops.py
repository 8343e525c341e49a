"""torch.autograd bindings of the HIP kernels (one Function per op family).

PyTorch is plumbing here: device memory (caching allocator), the current HIP
stream, and the autograd graph.  Every tensor op on the hot path is a call
through the C ABI of libm355seg.so; nothing falls back to torch or to the CPU.
"""
import ctypes as C
import os
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import ACT_LEAKY_RELU, ACT_NONE, ACT_RELU, ConvDesc, NormDesc, check

# Arithmetic of the 3x3x3 convolutions:
#   "fp32" (default)  fp32 tensors, fp32 results.  The 3x3x3 convolutions with >= 8 input and > 4 output channels (forward / data
#                     gradient; weight gradient: > 4 channels on both sides) and the k2 s2 conv-transpose forward run on the bf16 matrix pipe: every operand is split EXACTLY into three bf16
#                     values (8 + 8 + 8 significant bits) and six plane products per fp32 product are accumulated in fp32
#                     (M355_COMPUTE_F32X3, csrc/conv3d_f32x3.hip) -- measured against fp64 as accurate as the fp32 MFMA
#                     kernels (the error of both is the fp32 accumulation's), 1.5-1.7x their rate; the output conv and the
#                     edge layers' weight gradients run the fp32 MFMA / vector-ALU kernels.  Every 1e-4 parity claim refers to this mode.
#                     M355_FP32_SPLIT=0 in the environment maps "fp32" to "fp32_mfma".
#   "fp32_mfma"       every convolution on v_mfma_f32_32x32x2_f32 (M355_COMPUTE_F32): the same tensors, layouts and flow,
#                     only the conv kernels differ; results differ from "fp32" by summation-order noise.
#   "bf16" / "fp16"   operands rounded to 16 bits, fp32 accumulate (BASELINE cfg3 / cfg5).
# In the 16-bit modes the forward / data-gradient / weight-gradient kernels of the 3x3x3 stride-1 convolutions with
# more than 4 channels on both sides read the c8 activation layout (include/m355seg.h; `H16_TRAIN_C8` below); the
# edge layers go through the fp32-tensor entry points, which round the operands while staging them (Cin <= 4
# forward) or take the vector-ALU fp32 path (Cout <= 4).  Normalisation, pooling, conv-transpose, softmax and loss
# kernels always compute in fp32.
# Under torch.no_grad() the activations between conv -> norm/act -> conv (-> pool) live ONLY in c8 (`Act16`):
# no fp32 copy is written or read.
FP32_SPLIT = os.environ.get("M355_FP32_SPLIT", "1") != "0"
_COMPUTE = {"fp32": _lib.COMPUTE_F32X3 if FP32_SPLIT else _lib.COMPUTE_F32, "fp32_mfma": _lib.COMPUTE_F32,
            "bf16": _lib.COMPUTE_BF16, "fp16": _lib.COMPUTE_F16}
_DT16 = {_lib.COMPUTE_BF16: torch.bfloat16, _lib.COMPUTE_F16: torch.float16}
_compute_mode = "fp32"


def set_precision(mode: str):
    global _compute_mode
    if mode not in _COMPUTE:
        raise ValueError(f"precision must be one of {sorted(_COMPUTE)}, not {mode!r}")
    _compute_mode = mode


def get_precision() -> str:
    return _compute_mode


def is_fp32() -> bool:
    """fp32 tensors between the layers ("fp32" / "fp32_mfma")"""
    return _COMPUTE[_compute_mode] not in _DT16


class precision:
    """Context manager: `with ops.precision("bf16"): y = model(x)`."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_precision()
        set_precision(self.mode)

    def __exit__(self, *exc):
        set_precision(self.prev)


# Optional instrumentation used by bench.py: when set to a list, every conv launch appends
# (tag, flops, start_event, end_event, plan, algorithmic bytes) recorded on the launch stream.
CONV_PROFILE: Optional[list] = None
# restrict CONV_PROFILE to launches whose (tag, plan) is in this set: bench.py event-times only the dominant kernel
# inside its timed region (two timing events around each of the ~56 conv launches of a step are a measurable
# perturbation of a 12 ms step)
CONV_PROFILE_KEYS: Optional[set] = None
# ... and only every CONV_PROFILE_EVERY-th of those launches (a stride coprime with their count per step walks through
# all layers over a few steps)
CONV_PROFILE_EVERY = 1
_prof_seen = 0


def _prof_gate(tag, desc=None, which=0):
    """-> (CONV_PROFILE or None, plan) for a launch about to be issued"""
    prof = CONV_PROFILE
    if prof is None:
        return None, None
    plan = conv_plan(desc, which) if desc is not None else None
    if CONV_PROFILE_KEYS is not None and (tag, plan) not in CONV_PROFILE_KEYS:
        return None, plan
    if CONV_PROFILE_EVERY > 1:
        global _prof_seen
        _prof_seen += 1
        if _prof_seen % CONV_PROFILE_EVERY:
            return None, plan
    return prof, plan


def _conv_bytes(N, Cin, Cout, voxels, taps, in_elem, out_elem):
    """algorithmic HBM bytes of one conv launch: input once + output once + weights once"""
    return float(N) * voxels * (Cin * in_elem + Cout * out_elem) + 4.0 * Cin * Cout * taps


# ----------------------------------------------------------------- helpers
# torch.cuda.current_stream() builds a Stream object through three layers of Python (device-index parsing, availability
# checks): ~8 us a call, once per launch -- a quarter of the host time of a forward where the step is host-bound
# (msseg2 in the 16-bit modes: ~120 launches in 3.4 ms).  The raw handle of the current stream is one C call.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _cur_device is not None:
        return C.c_void_p(_raw_stream(_cur_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _require(*tensors, dtype=torch.float32):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.M355Error(
                "segmentation_pipeline_amd ops run only on the GPU through libm355seg.so "
                f"(got a {t.device} tensor); there is no CPU fallback")
        if t.dtype != dtype:
            raise _lib.M355Error(f"expected {dtype}, got {t.dtype}")


def _workspace(nbytes, device):
    # (every conv launch asks for its workspace here first: the one place that hands the library its work-queue pool)
    if device.index not in _lib._queue_pools:
        _lib.ensure_queue_pool(device)
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _dense_channels(t):
    """Return (tensor, batch_stride) for a 5-D tensor whose (C, D, H, W) block is
    dense; channel slices of a concat buffer qualify.  Anything else is compacted."""
    N, Cc, D, H, W = t.shape
    st = t.stride()
    S = D * H * W
    ok = ((W == 1 or st[4] == 1) and (H == 1 or st[3] == W) and (D == 1 or st[2] == H * W)
          and (Cc == 1 or st[1] == S))
    if not ok:
        t = t.contiguous()
        return t, Cc * S
    return t, (st[0] if N > 1 else Cc * S)


class OutSlot:
    """A destination inside a pre-allocated concat buffer.

    Wrapped in a plain object so autograd does not see the buffer as an input:
    producers write straight into their channel slice (no torch.cat copy, no
    CopySlices nodes) and return a fresh tensor aliasing the slice.
    """

    def __init__(self, buf: Optional[torch.Tensor], c0: int, c1: int, buf16=None):
        # buf16: the c8 twin of the buffer (an `Act16` spanning all its channels) in the 16-bit no-grad flow;
        # `buf` may then be None (nothing is kept in fp32)
        self.buf, self.c0, self.c1, self.buf16 = buf, c0, c1, buf16

    def view(self):
        return self.buf[:, self.c0:self.c1]

    def act16(self):
        return None if self.buf16 is None else self.buf16.slot(self.c0, self.c1)


class Concat:
    """Channel concatenation of tensors that already live in adjacent slices of
    `buf` (models/modular_unet.py:97: torch.cat([x_up, x_skip], dim=1))."""

    def __init__(self, buf, parts: Sequence):
        # buf: the fp32 concat buffer, or an `Act16` (c8 flow of the 16-bit modes under no_grad)
        self.buf, self.parts = buf, list(parts)
        assert sum(p.shape[1] for p in self.parts) == buf.shape[1]

    @property
    def shape(self):
        return self.buf.shape


def _alloc_out(out: Optional[OutSlot], shape, like):
    if out is None:
        return torch.empty(shape, dtype=like.dtype, device=like.device)
    v = out.view()
    if tuple(v.shape) != tuple(shape):
        raise _lib.M355Error(f"output slot shape {tuple(v.shape)} != op output shape {tuple(shape)}")
    # fresh tensor object aliasing the slice (see OutSlot)
    return torch.as_strided(out.buf, v.shape, v.stride(), v.storage_offset())


# ----------------------------------------------------- c8 activations (16-bit modes)
def h16_flow() -> int:
    """The 16-bit compute code when activations should flow in the c8 layout (a 16-bit precision mode; with autograd
    recording the c8-only training flow, H16_TRAIN_C8ONLY), else 0."""
    c = _COMPUTE[_compute_mode]
    if c not in _DT16:
        return 0
    if not torch.is_grad_enabled():
        return c
    # with autograd: the c8-only training flow (`_*C8Fn` below) -- since round 4 under synchronised batch norm too
    # (m355_norm_act_bwd_c8_reduce / _apply: the backward split around the all-reduce, on c8 tensors)
    return c if H16_TRAIN_C8ONLY else 0


class Act16:
    """An activation in the c8 layout: `data` is [N, CB_total, S, 8] (bf16 / fp16); this activation
    occupies the channel blocks [cb0, cb0 + ceil(C / 8)) of it (a slot of a concat buffer, or all of it)."""

    def __init__(self, data: torch.Tensor, C: int, spatial, compute: int, cb0: int = 0, t: Optional[torch.Tensor] = None):
        self.data, self.C, self.spatial, self.compute, self.cb0 = data, int(C), tuple(spatial), compute, cb0
        # c8-only TRAINING flow: the autograd handle of this activation -- a [N, CB, S, 8] tensor aliasing exactly its
        # channel blocks of `data` (produced by one of the `_*C8Fn` functions below); None outside autograd
        self.t = t

    @property
    def requires_grad(self):
        return self.t is not None and self.t.requires_grad

    @property
    def CB(self):
        return (self.C + 7) // 8

    def alias(self):
        """a fresh [N, CB, S, 8] tensor over exactly this activation's blocks (no autograd history)"""
        d = self.data
        N, CBt, S = d.shape[0], d.shape[1], d.shape[2]
        return torch.as_strided(d.detach(), (N, self.CB, S, 8), (CBt * S * 8, S * 8, 8, 1), d.storage_offset() + self.cb0 * S * 8)

    @staticmethod
    def empty(N, C, spatial, compute, device):
        S = spatial[0] * spatial[1] * spatial[2]
        return Act16(torch.empty((N, (C + 7) // 8, S, 8), dtype=_DT16[compute], device=device), C, spatial, compute)

    @property
    def shape(self):
        return (self.data.shape[0], self.C) + self.spatial

    @property
    def device(self):
        return self.data.device

    @property
    def S(self):
        return self.data.shape[2]

    def ptr(self):
        return C.c_void_p(self.data.data_ptr() + self.cb0 * self.S * 16)

    def batch_stride(self):
        return self.data.shape[1] * self.S * 8

    def slot(self, c0, c1):
        if c0 % 8:
            raise _lib.M355Error(f"a c8 slot must start at a multiple of 8 channels (got {c0})")
        return Act16(self.data, c1 - c0, self.spatial, self.compute, self.cb0 + c0 // 8)

    def to_f32(self):
        if torch.is_grad_enabled() and self.requires_grad:      # c8 training flow: differentiable (fallback ops)
            return _UnpackFn.apply(self.t, self)
        N = self.data.shape[0]
        x = torch.empty((N, self.C) + self.spatial, dtype=torch.float32, device=self.device)
        check(_lib.lib().m355_act16_unpack(self.ptr(), _p(x), N, self.C, self.S, self.batch_stride(), 0, self.compute,
                                           _stream()), "act16_unpack")
        return x


def pack_act16(x: torch.Tensor, compute: int, out: Optional[Act16] = None) -> Act16:
    """fp32 NCDHW tensor -> c8 (into `out`, a slot of matching shape, or a fresh buffer)."""
    if torch.is_grad_enabled() and x.requires_grad and h16_flow():   # c8 training flow: differentiable (fallback ops)
        N, Cc = x.shape[:2]
        if out is None:
            out = Act16.empty(N, Cc, x.shape[2:], compute, x.device)
        elif out.shape != tuple(x.shape):
            raise _lib.M355Error(f"c8 slot shape {out.shape} != tensor shape {tuple(x.shape)}")
        t = _PackFn.apply(x, out)
        return Act16(out.data, out.C, out.spatial, out.compute, out.cb0, t)
    _require(x)
    x, xbs = _dense_channels(x)
    N, Cc = x.shape[:2]
    if out is None:
        out = Act16.empty(N, Cc, x.shape[2:], compute, x.device)
    elif out.shape != tuple(x.shape):
        raise _lib.M355Error(f"c8 slot shape {out.shape} != tensor shape {tuple(x.shape)}")
    check(_lib.lib().m355_act16_pack(_p(x), out.ptr(), N, Cc, out.S, xbs, out.batch_stride(), compute, _stream()),
          "act16_pack")
    return out


def as_f32(x):
    """fp32 NCDHW view of an activation (c8 activations are converted: the fallback for ops without a c8 kernel)."""
    return x.to_f32() if isinstance(x, Act16) else x


# Packed weights are cached per parameter VERSION: the kernels want the filter re-laid-out ([c][tap][o], 16-bit for
# the 16-bit modes, flipped / transposed for the data gradient) and the library used to repack before every
# launch (234 launches per profiled run).  Weights only change at optimizer.step(), which bumps `_version`.
PACK_CACHE = True
# 16-bit precision modes, training: pack the conv input / output gradient to c8 once in the autograd function and run
# forward, data gradient and weight gradient on the c8 entry points (False: fp32 operands, staged inside the library)
H16_TRAIN_C8 = True
# ... and keep activations AND activation gradients of ModularUNet / Block3d only in c8 while training (round 3: the
# fp32 tensors the twin flow keeps beside the c8 ones bound the 16-bit step); False: the round-2 twin flow
H16_TRAIN_C8ONLY = os.environ.get("M355_TRAIN_C8ONLY", "1") != "0"
_bn_sync_group = None      # (defined properly with batch_norm_sync below)

# fp16 carries activation gradients multiplied by a power of two (include/m355seg.h, "grad_scale"); bf16 has fp32's exponent
# range: always 1.  The scale belongs to ONE backward pass: every node of the c8 training flow holds the `GradScale` cell of
# the forward pass that created it (`grad_scale_scope`, opened by the models' forward), the node where the gradient ENTERS
# the flow (the output convolution's backward, `_UnpackFn`) resolves the value, every other node of that pass reads it --
# two models forwarded before either backward, or an ensemble member of another size, never see each other's scale
# (round 3 kept one process-wide number, set as a side effect of the softmax out conv's forward).
# FP16_GRAD_SCALE = "auto": the value is CALIBRATED on the gradient that enters: 2^round(log2(target / max|dL/dlogits|)),
# measured once per (head, shape) -- one host read -- and again after an overflow; so a mean-reduced loss (gradient
# ~1 / (N * voxels)), a sum-reduced one (~1) and a StochasticMatrix head all travel at max|g| ~ `target` = 16, three decimal
# orders below the fp16 maximum (normalisation backward multiplies by rstd) and four above its normal minimum.  While a
# hipGraph is being captured nothing can be read: the cached calibration is used, or the size-derived estimate
# 2^(floor(log2(N * voxels)) + 3) of a mean-type loss.  A number fixes the scale.
# Overflow: the kernels OR into a device word (m355_overflow_flag_set) when a scaled gradient had to be clamped or a
# parameter gradient came out non-finite; `fp16_overflow()` reads and clears it -- trainer.train_step then skips the
# optimizer step, and the next backward re-calibrates with a 4x lower target.
FP16_GRAD_SCALE = "auto"
FP16_CHECK_OVERFLOW = True          # trainer.train_step: read the overflow word after every fp16 backward (one host sync)
_FP16_TARGET_MAX = 16.0
_fp16_target = _FP16_TARGET_MAX     # max|g| * scale aimed at by the calibration (lowered after an overflow)
_fp16_clean_checks = 0
_fp16_calibration = {}              # (head id, gradient shape) -> scale
_last_scale = 2.0 ** 16             # the most recently resolved fp16 scale (ops.grad_scale: tests, reporting)


class GradScale:
    """the loss scale of one forward / backward pass of the c8 training flow"""
    __slots__ = ("value", "auto")

    def __init__(self):
        self.value = None     # resolved where the gradient enters the flow
        self.auto = None      # size-derived estimate, recorded by the head's forward

    def get(self, compute) -> float:
        if compute != _lib.COMPUTE_F16:
            return 1.0
        if self.value is not None:
            return self.value
        return self.auto if self.auto is not None else _last_scale

    def resolve(self, compute, grad: torch.Tensor, key) -> float:
        """called by an entry node with the fp32 gradient that is about to be packed"""
        global _last_scale
        if compute != _lib.COMPUTE_F16:
            return 1.0
        if self.value is None:
            if FP16_GRAD_SCALE != "auto":
                self.value = float(FP16_GRAD_SCALE)
            else:
                cal = _fp16_calibration.get(key)
                if cal is None and not torch.cuda.is_current_stream_capturing():
                    import math
                    amax = float(grad.detach().abs().max())
                    if math.isfinite(amax) and amax > 0.0:
                        cal = 2.0 ** min(24, max(-8, round(math.log2(_fp16_target / amax))))
                        _fp16_calibration[key] = cal
                if cal is None:
                    cal = self.auto if self.auto is not None else _last_scale
                self.value = cal
            _last_scale = self.value
        return self.value


_gs_default = GradScale()
_gs_current = None


def _gs() -> GradScale:
    """the cell of the forward pass being recorded (outside a `grad_scale_scope`: one process-wide cell, as in round 3)"""
    return _gs_current if _gs_current is not None else _gs_default


class grad_scale_scope:
    """`with grad_scale_scope():` around ONE forward pass (ModularUNet / NestedResUNet open it themselves): the c8 nodes
    created inside share a fresh GradScale cell.  Nested scopes (a model called by an ensemble that is itself inside a
    scope) keep the outer cell -- one loss, one scale."""

    def __enter__(self):
        global _gs_current
        self.prev = _gs_current
        if _gs_current is None:
            _gs_current = GradScale()
        return _gs_current

    def __exit__(self, *exc):
        global _gs_current
        _gs_current = self.prev


def grad_scale(compute) -> float:
    """the fp16 loss scale most recently resolved (1.0 for bf16)"""
    return _last_scale if compute == _lib.COMPUTE_F16 else 1.0


def _auto_grad_scale(n_elements) -> float:
    import math
    return 2.0 ** min(24, max(0, int(math.floor(math.log2(max(1, n_elements)))) + 3))


_overflow_words = {}   # device index -> int32 device word handed to the library (m355_overflow_flag_set)


def _overflow_word(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    w = _overflow_words.get(idx)
    if w is None and not torch.cuda.is_current_stream_capturing():
        w = torch.zeros(1, dtype=torch.int32, device=device)
        check(_lib.lib().m355_overflow_flag_set(_p(w), idx), "overflow_flag_set")
        _overflow_words[idx] = w
    return w


class _DeferredOverflow:
    """the overflow word of the LAST step on its way to the host (pinned copy + event): read when it has arrived, never
    waited for -- the scale adaptation may lag a step, the decision to skip does not (it is taken on the device)"""

    def __init__(self):
        self.host = None
        self.event = None
        self.armed = False


_deferred = {}      # device index -> _DeferredOverflow


def _adapt_after_overflow(hit: int):
    global _fp16_target, _fp16_clean_checks
    if hit:
        _fp16_calibration.clear()
        _fp16_target = max(_fp16_target / 4.0, 2.0 ** -6)
        _fp16_clean_checks = 0
    else:
        _fp16_clean_checks += 1
        if _fp16_clean_checks >= 200 and _fp16_target < _FP16_TARGET_MAX:
            _fp16_target = min(_FP16_TARGET_MAX, _fp16_target * 2.0)
            _fp16_calibration.clear()
            _fp16_clean_checks = 0


def fp16_found_inf(device) -> Optional[torch.Tensor]:
    """The overflow word as the `found_inf` tensor of torch's fused optimizers (float32, 1.0 = skip this step), WITHOUT a
    host synchronisation: torch.optim.SGD / Adam / AdamW built with fused=True take it (`optimizer.found_inf`, the protocol
    torch.cuda.amp.GradScaler uses) and leave parameters and optimizer state untouched on the device when it is set.  The
    word is cleared on the device; its value travels to the host asynchronously and lowers the loss-scale target when it
    has arrived (usually before the next backward).  None when no fp16 pass has run on this device."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    w = _overflow_words.get(idx)
    if w is None:
        return None
    d = _deferred.get(idx)
    if d is None:
        d = _deferred[idx] = _DeferredOverflow()
        d.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        d.event = torch.cuda.Event()
    elif d.armed and d.event.query():          # last step's word has arrived
        _adapt_after_overflow(int(d.host[0]))
        d.armed = False
    found = (w != 0).to(torch.float32).reshape(())       # 0-dim, as torch.cuda.amp.GradScaler hands it over
    if not d.armed:
        d.host.copy_(w, non_blocking=True)
        d.event.record()
        d.armed = True
    w.zero_()
    return found


def fp16_overflow(device=None) -> int:
    """Read and clear the overflow word of the fp16 training flow (one host synchronisation): non-zero when, since the last
    call, a loss-scaled gradient was clamped to the fp16 range (bit 0) or a parameter gradient came out non-finite (bit 1).  The caller
    skips its optimizer step; the next backward re-calibrates the scale with a 4x lower target (raised again, 2x per 200
    clean checks)."""
    hit = 0
    for idx, w in _overflow_words.items():
        if device is not None and device.index not in (None, idx):
            continue
        v = int(w.item())
        if v:
            w.zero_()
            hit |= v
    _adapt_after_overflow(hit)
    return hit
# an encoder block's last norm + activation pass also emits the AvgPool3d(2, 2) the next level consumes
FUSE_POOL = os.environ.get("M355_FUSE_POOL", "1") != "0"


# Parameters that own packed forms (id -> weak reference).  optimizer.step changes every one of them at once; the first
# conv that then finds its weight stale re-packs ALL stale registered forms in one launch (m355_conv3d_pack_batch)
# instead of ~37 small packs spread over the step, each on the critical path in front of its conv.
_pack_registry = {}
PACK_BATCH = os.environ.get("M355_PACK_BATCH", "1") != "0"
# While a train step is being CAPTURED into a hipGraph the batched re-pack must contain exactly the captured model's
# forms: `pack_scope(params)` marks them stale (so the re-pack kernel is part of the captured step whatever ran before --
# another model's eager forward between this model's last optimizer step and the capture refreshes ALL stale registered
# forms, and a capture that then finds its weights "fresh" would replay with the packed weights of the capture step
# forever) and keeps other models' forms out of the graph.
_pack_scope = None


class pack_scope:
    def __init__(self, params):
        self.ids = {id(p) for p in params}
        self.params = list(params)

    def __enter__(self):
        global _pack_scope
        self.prev, _pack_scope = _pack_scope, self.ids
        for p in self.params:
            cache = getattr(p, "_m355_packed", None)
            if cache is not None:
                p._m355_packed = (-1, cache[1], cache[2])
        return self

    def __exit__(self, *exc):
        global _pack_scope
        _pack_scope = self.prev


def _repack_stale(L, device):
    """Refresh, in place and with one launch, every cached packed form whose parameter has a new version."""
    items = []
    for key, ref in list(_pack_registry.items()):
        if _pack_scope is not None and key not in _pack_scope:
            continue
        w = ref()
        cache = getattr(w, "_m355_packed", None) if w is not None else None
        if cache is None or cache[1] != w.data_ptr() or w.device != device:
            if w is None or cache is None or cache[1] != w.data_ptr():
                del _pack_registry[key]   # gone, or its storage moved: the per-weight path starts a new cache
            continue
        if cache[0] == w._version:
            continue
        for (which, *_), (buf, d) in cache[2].items():
            items.append(_lib.PackItem(d, which, w.data_ptr(), buf.data_ptr()))
        w._m355_packed = (w._version, cache[1], cache[2])
    if items:
        arr = (_lib.PackItem * len(items))(*items)
        check(L.m355_conv3d_pack_batch(C.cast(arr, C.c_void_p), len(items), _stream()), "conv3d_pack_batch")


def _packed_weight(weight, d, which):
    """-> (pointer argument, flags) for a conv entry point: the cached packed form of `weight` for descriptor
    `d` (which: 0 forward, 1 data gradient) with M355_CONV_W_PACKED, or the plain weight with flags 0."""
    if not PACK_CACHE:
        return weight, 0
    L = _lib.lib()
    ver, ptr = weight._version, weight.data_ptr()
    cache = getattr(weight, "_m355_packed", None)
    if (cache is not None and cache[0] != ver and cache[1] == ptr and PACK_BATCH and id(weight) in _pack_registry):
        _repack_stale(L, weight.device)   # a parameter after optimizer.step: all stale forms of all parameters at once
        cache = weight._m355_packed
    if cache is None or cache[0] != ver or cache[1] != ptr:
        cache = (ver, ptr, {})
        try:
            weight._m355_packed = cache
        except (AttributeError, RuntimeError):
            return weight, 0
    # The packed layout depends on the channel counts, the arithmetic and the kernel family only -- not on the spatial
    # extent or the batch: ONE buffer per (direction, family) serves every input shape (variable-size validation
    # volumes used to add a full copy of every weight per distinct shape, all re-packed after each optimizer.step).
    small = 1 if (which == 0 and d.Cout <= 4 and conv_plan(d, 0)[0] == 2) else 0   # Cout <= 4 forward: its own layout
    key = (which, d.Cin, d.Cout, d.compute, small)
    ent = cache[2].get(key)
    if ent is None:
        nbytes = L.m355_conv3d_packed_bytes(C.byref(d), which)
        if nbytes == 0:
            return weight, 0
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=weight.device)
        check(L.m355_conv3d_pack(C.byref(d), which, _p(weight), _p(buf), _stream()), "conv3d_pack")
        d0 = ConvDesc(d.N, d.Cin, d.Cout, d.D, d.H, d.W, d.k, d.stride, d.pad, d.out_pad, 0, 0, d.compute, 0)
        ent = cache[2][key] = (buf, d0)
        if weight.is_leaf and isinstance(weight, torch.nn.Parameter):   # (derived weights -- blur / WS -- are new tensors every step)
            _pack_registry[id(weight)] = weakref.ref(weight)
    return ent[0], _lib.CONV_W_PACKED


def _c8_twin(t, compute):
    """the c8 twin a producer pass attached to this fp32 tensor (norm_act forward / backward in the 16-bit
    training flow), if it describes exactly this tensor"""
    rec = getattr(t, "_m355_c8", None)
    if rec is None:
        return None
    tw, version = rec
    if (version == t._version and tw.compute == compute and tw.shape == tuple(t.shape) and tw.device == t.device):
        return tw
    return None   # (an in-place op on the fp32 tensor since the twin was written bumps _version: the twin is stale)


def _with_flags(d, flags):
    if flags == d.flags:
        return d
    return ConvDesc(d.N, d.Cin, d.Cout, d.D, d.H, d.W, d.k, d.stride, d.pad, d.out_pad, d.x_batch_stride,
                    d.y_batch_stride, d.compute, flags)


def _c8_channel_partials(x16: Act16, stats: dict):
    """statistics partials of a c8 tensor (pre-norm output of a conv without fused statistics)"""
    L = _lib.lib()
    N, Cc = x16.shape[:2]
    slots = int(L.m355_act16_partials_slots(x16.S))
    part = torch.empty((N, slots, Cc, 2), dtype=torch.float32, device=x16.device)
    check(L.m355_act16_channel_partials(x16.ptr(), x16.batch_stride(), N, Cc, x16.S, x16.compute, _p(part), _stream()),
          "act16_channel_partials")
    stats["partials"], stats["slots"] = part, slots


def _conv3d_act16(x16: Act16, weight, bias, add, stats, c8_out=False, softmax=False):
    """3x3x3 / s1 / p1 forward on a c8 input (no autograd: the c8 flow only exists under no_grad).  c8_out: the
    result (a pre-norm tensor) is written by the conv epilogue as c8 too and returned as an Act16.  softmax:
    nn.Softmax(dim=1) of the result (the conv's epilogue where the kernel variant has one, else a separate pass)."""
    L = _lib.lib()
    _require(weight, bias, add)
    weight = weight.contiguous()
    N, Cin, D, H, W = x16.shape
    Cout = weight.shape[0]
    c8_out = c8_out and add is None
    if add is not None:
        add = add.contiguous()
    d = _conv_desc(N, Cin, Cout, D, H, W, 3, 1, 1, 0, 0, compute=x16.compute)
    wbuf, flags = _packed_weight(weight, d, 0)
    fuse_sm = (softmax and not c8_out and add is None and stats is None
               and L.m355_conv3d_fuses_softmax(C.byref(d)) != 0)
    d = _with_flags(d, flags | (_lib.CONV_SOFTMAX if fuse_sm else 0))
    ws = _workspace(L.m355_conv3d_h16_workspace(C.byref(d), 0), x16.device)
    prof, plan = _prof_gate("conv3d_fwd", d, 0)
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    part = None
    slots = 0
    if stats is not None:
        slots = L.m355_conv3d_stats_slots_c8(C.byref(d)) if c8_out else L.m355_conv3d_stats_slots(C.byref(d))
    if slots > 0:
        part = torch.empty((N, slots, Cout, 2), dtype=torch.float32, device=x16.device)
        stats["partials"], stats["slots"] = part, slots
    if c8_out:
        y = Act16.empty(N, Cout, (D, H, W), x16.compute, x16.device)
        check(L.m355_conv3d_fwd_h16_c8(C.byref(d), x16.ptr(), x16.batch_stride(), _p(wbuf), _p(bias), y.ptr(),
                                       y.batch_stride(), _p(part), _p(ws), ws.numel(), _stream()), "conv3d_fwd_h16_c8")
    else:
        y = torch.empty((N, Cout, D, H, W), dtype=torch.float32, device=x16.device)
        check(L.m355_conv3d_fwd_h16(C.byref(d), x16.ptr(), x16.batch_stride(), _p(wbuf), _p(bias), _p(add), _p(y),
                                    _p(part), _p(ws), ws.numel(), _stream()), "conv3d_fwd_h16")
    if prof is not None:
        e1.record()
        prof.append(("conv3d_fwd", 2.0 * 27 * Cin * Cout * N * D * H * W, e0, e1, plan,
                     _conv_bytes(N, Cin, Cout, D * H * W, 27, 2, 2 if c8_out else 4)))
    if c8_out and stats is not None and slots == 0:
        _c8_channel_partials(y, stats)
    if softmax and not fuse_sm:
        y = softmax_channels(as_f32(y))
    return y


# ------------------------------------------------- c8-only TRAINING flow (16-bit modes, autograd on)
# Activations and their gradients are [N, CB, S, 8] 16-bit tensors (the c8 layout) on BOTH sides of every function:
# forward kernels as in the no-grad flow, backward kernels from csrc/train16.hip / convt.hip (include/m355seg.h,
# "c8-only TRAINING flow").  An `Act16` carries the autograd handle `t` of its tensor; the functions read their
# operands through the `Act16` objects in `meta` (pointers, batch strides) and take the handles only as graph edges.
def _c8t(t):
    """(tensor, batch stride) of a [N, CB, S, 8] c8 tensor whose samples are dense -- a slot alias or a slice of a
    gradient buffer along the block axis; anything else is compacted first"""
    N, CB, S, _ = t.shape
    st = t.stride()
    if st[3] != 1 or st[2] != 8 or (CB > 1 and st[1] != S * 8) or (N > 1 and st[0] % 8):
        t = t.contiguous()
        return t, CB * S * 8
    return t, (st[0] if N > 1 else CB * S * 8)


def _pack_scaled(x, compute, scale):
    """fp32 NCDHW gradient -> dense c8 tensor, multiplied by the loss scale of the mode"""
    x, xbs = _dense_channels(x)
    N, Cc = x.shape[:2]
    S = x.shape[2] * x.shape[3] * x.shape[4]
    out = torch.empty((N, (Cc + 7) // 8, S, 8), dtype=_DT16[compute], device=x.device)
    check(_lib.lib().m355_act16_pack_scaled(_p(x), _p(out), N, Cc, S, xbs, 0, compute, float(scale), _stream()),
          "act16_pack_scaled")
    return out


def _unpack_scaled(t, Cc, spatial, compute, scale):
    t, bs = _c8t(t)
    N, S = t.shape[0], t.shape[2]
    x = torch.empty((N, Cc) + tuple(spatial), dtype=torch.float32, device=t.device)
    check(_lib.lib().m355_act16_unpack_scaled(_p(t), _p(x), N, Cc, S, bs, 0, compute, float(scale), _stream()),
          "act16_unpack_scaled")
    return x


class _PackFn(torch.autograd.Function):
    """fp32 NCDHW -> c8 (the generic entry of a tensor into the c8 training flow); backward: the c8 gradient back to
    fp32, loss scale removed"""

    @staticmethod
    def forward(ctx, x, out: Act16):
        _require(x)
        xd, xbs = _dense_channels(x)
        N, Cc = xd.shape[:2]
        check(_lib.lib().m355_act16_pack(_p(xd), out.ptr(), N, Cc, out.S, xbs, out.batch_stride(), out.compute, _stream()),
              "act16_pack")
        ctx.info = (Cc, out.spatial, out.compute)
        ctx.gs = _gs()
        return out.alias()

    @staticmethod
    def backward(ctx, dy16):
        Cc, spatial, compute = ctx.info
        return _unpack_scaled(dy16, Cc, spatial, compute, 1.0 / ctx.gs.get(compute)), None


class _UnpackFn(torch.autograd.Function):
    """c8 -> fp32 NCDHW (ops without a c8 kernel: trilinear upsampling, dropout, space-to-depth ...); backward: the fp32
    gradient enters the c8 flow (loss scale applied)"""

    @staticmethod
    def forward(ctx, t, a: Act16):
        N = a.data.shape[0]
        x = torch.empty((N, a.C) + a.spatial, dtype=torch.float32, device=a.device)
        check(_lib.lib().m355_act16_unpack(a.ptr(), _p(x), N, a.C, a.S, a.batch_stride(), 0, a.compute, _stream()),
              "act16_unpack")
        ctx.compute = a.compute
        ctx.gs = _gs()
        if a.compute == _lib.COMPUTE_F16:
            _overflow_word(a.device)
        return x

    @staticmethod
    def backward(ctx, dy):
        # (a gradient ENTERS the c8 flow here: this node may be the one that fixes the scale of the pass)
        scale = ctx.gs.resolve(ctx.compute, dy, ("unpack", tuple(dy.shape)))
        return _pack_scaled(dy, ctx.compute, scale), None


@dataclass
class _C8ConvMeta:
    x16: "Act16"                      # the whole (possibly concatenated) conv input
    part_blocks: Sequence[int]        # channel blocks of each autograd part of it
    out16: Optional["Act16"] = None   # c8 destination (None with f32_out)
    f32_out: bool = False             # fp32 NCDHW result (the out conv), optionally with the fused softmax
    softmax: bool = False
    stats: Optional[dict] = None
    has_add: bool = False


class _Conv3dC8Fn(torch.autograd.Function):
    """3x3x3 / s1 / p1 convolution of the c8 training flow: c8 in -> c8 out (pre-norm tensor, statistics fused), or
    -> fp32 (+ softmax) for the output convolution.  Backward: weight / bias gradient from the two c8 operands
    (m355_conv3d_bwd_weight_c8), data gradient c8 -> c8 (m355_conv3d_bwd_data_h16_c8) sliced per concat part."""

    @staticmethod
    def forward(ctx, weight, bias, add, meta: _C8ConvMeta, *part_ts):
        L = _lib.lib()
        x16 = meta.x16
        _require(weight, bias, add)
        weight = weight.contiguous()
        N, Cin, D, H, W = x16.shape
        Cout = weight.shape[0]
        d = _conv_desc(N, Cin, Cout, D, H, W, 3, 1, 1, 0, 0, compute=x16.compute)
        wbuf, flags = _packed_weight(weight, d, 0)
        fuse_sm = meta.softmax and add is None and L.m355_conv3d_fuses_softmax(C.byref(d)) != 0
        dk = _with_flags(d, flags | (_lib.CONV_SOFTMAX if fuse_sm else 0))
        ws = _workspace(L.m355_conv3d_h16_workspace(C.byref(d), 0), x16.device)
        prof, plan = _prof_gate("conv3d_fwd", d, 0)
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        part = None
        slots = 0
        if meta.stats is not None:
            slots = L.m355_conv3d_stats_slots(C.byref(d)) if meta.f32_out else L.m355_conv3d_stats_slots_c8(C.byref(d))
        if slots > 0:
            part = torch.empty((N, slots, Cout, 2), dtype=torch.float32, device=x16.device)
            meta.stats["partials"], meta.stats["slots"] = part, slots
        if meta.f32_out:
            if add is not None:
                add = add.contiguous()
            y = torch.empty((N, Cout, D, H, W), dtype=torch.float32, device=x16.device)
            check(L.m355_conv3d_fwd_h16(C.byref(dk), x16.ptr(), x16.batch_stride(), _p(wbuf), _p(bias), _p(add), _p(y),
                                        _p(part), _p(ws), ws.numel(), _stream()), "conv3d_fwd_h16")
            if meta.softmax and not fuse_sm:
                logits, y = y, torch.empty_like(y)
                check(L.m355_softmax_fwd(_p(logits), _p(y), N, Cout, 1, D * H * W, 0.0, _stream()), "softmax_fwd")
            if x16.compute == _lib.COMPUTE_F16:     # the head of an fp16 pass: size-derived estimate (capture fallback)
                _gs().auto = _auto_grad_scale(N * D * H * W)
                _overflow_word(x16.device)
            out = y
        else:
            y16 = meta.out16
            check(L.m355_conv3d_fwd_h16_c8(C.byref(dk), x16.ptr(), x16.batch_stride(), _p(wbuf), _p(bias), y16.ptr(),
                                           y16.batch_stride(), _p(part), _p(ws), ws.numel(), _stream()), "conv3d_fwd_h16_c8")
            out = y16.alias()
        if prof is not None:
            e1.record()
            prof.append(("conv3d_fwd", 2.0 * 27 * Cin * Cout * N * D * H * W, e0, e1, plan,
                         _conv_bytes(N, Cin, Cout, D * H * W, 27, 2, 4 if meta.f32_out else 2)))
        ctx.meta, ctx.desc = meta, d
        ctx.gs = _gs()
        ctx.has_bias = bias is not None
        ctx.x_info = (x16.C, x16.spatial, x16.compute)
        xa = x16.alias()
        if meta.softmax:
            ctx.save_for_backward(xa, weight, out)
        else:
            ctx.save_for_backward(xa, weight)
        return out

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        meta, d = ctx.meta, ctx.desc
        compute = d.compute
        if meta.softmax:
            xa, weight, y = ctx.saved_tensors
            dy = dy.contiguous()
            dl = torch.empty_like(y)
            check(L.m355_softmax_bwd(_p(y), _p(dy), _p(dl), d.N, d.Cout, 1, y.numel() // (d.N * d.Cout), _stream()),
                  "softmax_bwd")
            dy = dl
        else:
            xa, weight = ctx.saved_tensors
        dadd = dy if (meta.has_add and ctx.needs_input_grad[2]) else None
        if meta.f32_out:        # the gradient enters the c8 flow here: this node fixes the scale of the pass
            scale = ctx.gs.resolve(compute, dy, (id(weight), tuple(dy.shape)))
            dy16 = _pack_scaled(dy, compute, scale)
        else:
            scale = ctx.gs.get(compute)
            dy16 = dy
        dy16, dybs = _c8t(dy16)
        xa, xbs = _c8t(xa)
        need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and ctx.has_bias
        need_x = any(ctx.needs_input_grad[4:])
        dw = db = None
        if need_w or need_b:
            dw = torch.empty_like(weight)
            db = torch.empty(d.Cout, dtype=weight.dtype, device=weight.device) if ctx.has_bias else None
            prof, _ = _prof_gate("conv3d_bwd_weight")
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ws = _workspace(L.m355_conv3d_bwd_weight_c8_workspace(C.byref(d)), weight.device)
            check(L.m355_conv3d_bwd_weight_c8(C.byref(d), _p(xa), xbs, _p(dy16), dybs, _p(dw), _p(db), 1.0 / scale, _p(ws),
                                              ws.numel(), _stream()), "conv3d_bwd_weight_c8")
            if prof is not None:
                e1.record()
                vox = d.D * d.H * d.W
                prof.append(("conv3d_bwd_weight", 2.0 * 27 * d.Cin * d.Cout * d.N * vox, e0, e1, None,
                             _conv_bytes(d.N, d.Cin, d.Cout, vox, 27, 2, 2)))
        dparts: List[Optional[torch.Tensor]] = [None] * len(meta.part_blocks)
        if need_x:
            S = d.D * d.H * d.W
            dx16 = torch.empty((d.N, (d.Cin + 7) // 8, S, 8), dtype=_DT16[compute], device=weight.device)
            dd = _conv_desc(d.N, d.Cin, d.Cout, d.D, d.H, d.W, 3, 1, 1, 0, 0, compute=compute)
            wbuf, flags = _packed_weight(weight, dd, 1)
            ws = _workspace(L.m355_conv3d_h16_workspace(C.byref(dd), 1), weight.device)
            prof, plan = _prof_gate("conv3d_bwd_data", dd, 1)
            dd = _with_flags(dd, flags)
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(L.m355_conv3d_bwd_data_h16_c8(C.byref(dd), _p(dy16), dybs, _p(wbuf), _p(dx16), 0, _p(ws), ws.numel(),
                                                _stream()), "conv3d_bwd_data_h16_c8")
            if prof is not None:
                e1.record()
                prof.append(("conv3d_bwd_data", 2.0 * 27 * d.Cin * d.Cout * d.N * S, e0, e1, plan,
                             _conv_bytes(d.N, d.Cin, d.Cout, S, 27, 2, 2)))
            b0 = 0
            for i, nb in enumerate(meta.part_blocks):
                if ctx.needs_input_grad[4 + i]:
                    dparts[i] = dx16 if len(meta.part_blocks) == 1 else dx16[:, b0:b0 + nb]
                b0 += nb
        return (dw if need_w else None, db if need_b else None, dadd, None, *dparts)


def _conv3d_c8_train(x16: Act16, parts, weight, bias, add, stats, c8_out, softmax, out16=None):
    """dispatch of `_conv3d_act16` while autograd records (c8 training flow)"""
    if parts is None:
        parts = [x16]
    if any(p.C % 8 for p in parts[:-1]):
        raise _lib.M355Error("c8 training flow: concat parts must be whole channel blocks")
    if c8_out and out16 is None:
        N, _, D, H, W = x16.shape
        out16 = Act16.empty(N, weight.shape[0], (D, H, W), x16.compute, x16.device)
    meta = _C8ConvMeta(x16, [p.CB for p in parts], out16=out16, f32_out=not c8_out, softmax=softmax, stats=stats,
                       has_add=add is not None)
    out = _Conv3dC8Fn.apply(weight, bias, add, meta, *[p.t for p in parts])
    if not c8_out:
        return out
    y16 = meta.out16
    if stats is not None and "partials" not in stats:
        _c8_channel_partials(y16, stats)
    return Act16(y16.data, y16.C, y16.spatial, y16.compute, y16.cb0, out)


@dataclass
class _C8NormMeta:
    x16: "Act16"
    add16: Optional["Act16"]
    out16: Optional["Act16"]
    cfg: "NormCfg"
    pool: bool = False
    pooled16: Optional["Act16"] = None


class _NormActC8Fn(torch.autograd.Function):
    """normalisation + activation (+ residual add) of the c8 training flow, c8 -> c8; optionally nn.AvgPool3d(2, 2) of
    the result as a second output (an encoder block's output feeds the skip connection and the next level).  Backward:
    ONE call of m355_norm_act_bwd_c8, which also sums the skip-path and the un-pooled gradient."""

    @staticmethod
    def forward(ctx, x_t, gamma, beta, add_t, meta: _C8NormMeta):
        L = _lib.lib()
        x, cfg = meta.x16, meta.cfg
        _require(gamma, beta)
        N, Cc, spatial, S = x.shape[0], x.C, x.spatial, x.S
        out16 = meta.out16
        d = NormDesc(N, Cc, S, cfg.groups, cfg.act, cfg.eps, cfg.slope, 0, 0, 0)
        use_batch = cfg.groups > 0 or cfg.training or cfg.running_mean is None
        if use_batch and not (cfg.stats and cfg.stats.get("partials") is not None):
            cfg.stats = {} if cfg.stats is None else cfg.stats
            _c8_channel_partials(x, cfg.stats)
        mean, rstd, use_batch = _norm_statistics(L, d, None, cfg, N, Cc, device=x.device)
        a16 = meta.add16
        check(L.m355_norm_act_fwd_c8(C.byref(d), x.ptr(), x.batch_stride(), _p(mean), _p(rstd), _p(gamma), _p(beta),
                                     a16.ptr() if a16 is not None else None, a16.batch_stride() if a16 is not None else 0,
                                     out16.ptr(), out16.batch_stride(), x.compute, _stream()), "norm_act_fwd_c8")
        ctx.desc, ctx.batch_stats, ctx.has_add, ctx.has_affine = d, use_batch, a16 is not None, gamma is not None
        ctx.compute, ctx.spatial = x.compute, spatial
        ctx.gs = _gs()
        ctx.sync = cfg.sync             # synchronised batch norm: (process group, element count over all ranks)
        ctx.save_for_backward(x.alias(), mean, rstd, gamma, beta)
        if not meta.pool:
            return out16.alias()
        D, H, W = spatial
        p16 = meta.pooled16
        check(L.m355_avgpool3d_2x_fwd_h16(out16.ptr(), p16.ptr(), N, Cc, D, H, W, out16.batch_stride(), p16.batch_stride(),
                                          x.compute, _stream()), "avgpool3d_2x_fwd_h16")
        return out16.alias(), p16.alias()

    @staticmethod
    def backward(ctx, dy16, dpool16=None):
        L = _lib.lib()
        xa, mean, rstd, gamma, beta = ctx.saved_tensors
        d = ctx.desc
        compute = ctx.compute
        xa, xbs = _c8t(xa)
        dyb = dpb = 0
        if dy16 is not None:
            dy16, dyb = _c8t(dy16)
        if dpool16 is not None:
            dpool16, dpb = _c8t(dpool16)
        dx16 = torch.empty((d.N, (d.C + 7) // 8, d.S, 8), dtype=_DT16[compute], device=xa.device)
        dgamma = torch.empty(d.C, dtype=torch.float32, device=xa.device) if ctx.has_affine else None
        dbeta = torch.empty(d.C, dtype=torch.float32, device=xa.device) if ctx.has_affine else None
        ws = _workspace(L.m355_norm_workspace(C.byref(d)), xa.device)
        D, H, W = ctx.spatial
        training, unscale = 1 if ctx.batch_stats else 0, 1.0 / ctx.gs.get(compute)
        if ctx.sync is not None:        # the two halves around the all-reduce of the per-channel gradient means
            import torch.distributed as dist
            group, total = ctx.sync
            stat_m = torch.empty(2 * d.C, dtype=torch.float32, device=xa.device)
            check(L.m355_norm_act_bwd_c8_reduce(C.byref(d), _p(xa), xbs, _p(dy16), dyb, _p(dpool16), dpb, D, H, W, _p(mean),
                                                _p(rstd), _p(gamma), _p(beta), _p(dgamma), _p(dbeta), training, _p(total),
                                                unscale, _p(stat_m), compute, _p(ws), ws.numel(), _stream()),
                  "norm_act_bwd_c8_reduce")
            dist.all_reduce(stat_m, group=group)
            check(L.m355_norm_act_bwd_c8_apply(C.byref(d), _p(xa), xbs, _p(dy16), dyb, _p(dpool16), dpb, D, H, W, _p(mean),
                                               _p(rstd), _p(gamma), _p(beta), _p(stat_m), _p(dx16), 0, compute, _stream()),
                  "norm_act_bwd_c8_apply")
        else:
            check(L.m355_norm_act_bwd_c8(C.byref(d), _p(xa), xbs, _p(dy16), dyb, _p(dpool16), dpb, D, H, W, _p(mean), _p(rstd),
                                         _p(gamma), _p(beta), _p(dx16), 0, _p(dgamma), _p(dbeta), training, unscale, compute,
                                         _p(ws), ws.numel(), _stream()), "norm_act_bwd_c8")
        dadd = dy16 if (ctx.has_add and ctx.needs_input_grad[3]) else None
        return dx16, dgamma, dbeta, dadd, None


def _norm_act_c8_train(x: Act16, gamma, beta, add, cfg: "NormCfg", pool=False):
    out16 = cfg.out.act16() if cfg.out is not None else None
    if out16 is not None and out16.shape != x.shape:
        raise _lib.M355Error(f"c8 slot shape {out16.shape} != op output shape {x.shape}")
    add16 = None
    if add is not None:
        add16 = add if isinstance(add, Act16) else pack_act16(add, x.compute)
    if out16 is None:
        out16 = Act16.empty(x.shape[0], x.C, x.spatial, x.compute, x.device)
    pooled16 = None
    if pool:
        D, H, W = x.spatial
        pooled16 = Act16.empty(x.shape[0], x.C, (D // 2, H // 2, W // 2), x.compute, x.device)
    meta = _C8NormMeta(x, add16, out16, cfg, pool, pooled16)
    res = _NormActC8Fn.apply(x.t, gamma, beta, add16.t if add16 is not None else None, meta)
    o = meta.out16
    if pool:
        y = Act16(o.data, o.C, o.spatial, o.compute, o.cb0, res[0])
        p = meta.pooled16
        return y, Act16(p.data, p.C, p.spatial, p.compute, p.cb0, res[1])
    return Act16(o.data, o.C, o.spatial, o.compute, o.cb0, res)


class _PoolC8Fn(torch.autograd.Function):
    """nn.AvgPool3d(2, 2) c8 -> c8; skip=True: also returns the input as it continues into the skip connection, so that
    the backward receives both gradients and sums them in the pool-backward pass"""

    @staticmethod
    def forward(ctx, x_t, x: Act16, out16: Optional[Act16], skip: bool):
        N, Cc, D, H, W = x.shape
        y16 = out16 if out16 is not None else Act16.empty(N, Cc, (D // 2, H // 2, W // 2), x.compute, x.device)
        check(_lib.lib().m355_avgpool3d_2x_fwd_h16(x.ptr(), y16.ptr(), N, Cc, D, H, W, x.batch_stride(), y16.batch_stride(),
                                                   x.compute, _stream()), "avgpool3d_2x_fwd_h16")
        ctx.info = (N, Cc, D, H, W, x.compute)
        return (x_t.view_as(x_t), y16.alias()) if skip else y16.alias()

    @staticmethod
    def backward(ctx, *grads):
        N, Cc, D, H, W, compute = ctx.info
        g_skip, g_pool = (grads if len(grads) == 2 else (None, grads[0]))
        if g_pool is None:
            return g_skip, None, None, None
        g_pool, pbs = _c8t(g_pool)
        sbs = 0
        if g_skip is not None:
            g_skip, sbs = _c8t(g_skip)
        dx16 = torch.empty((N, (Cc + 7) // 8, D * H * W, 8), dtype=_DT16[compute], device=g_pool.device)
        check(_lib.lib().m355_avgpool3d_2x_bwd_h16(_p(g_pool), _p(g_skip), _p(dx16), N, Cc, D, H, W, pbs, sbs, 0, compute,
                                                   _stream()), "avgpool3d_2x_bwd_h16")
        return dx16, None, None, None


class _ConvTC8Fn(torch.autograd.Function):
    """nn.ConvTranspose3d(kernel_size=2, stride=2) of the c8 training flow: c8 -> c8 straight into its concat slot;
    backward on the 16-bit MFMA from c8 operands where the library has the kernels (Cout <= 128), else through the
    fp32 kernels (the deepest levels: a few MB)."""

    @staticmethod
    def forward(ctx, x_t, weight, bias, x: Act16, y16: Act16):
        L = _lib.lib()
        _require(weight, bias)
        weight = weight.contiguous()
        N, Cin, D, H, W = x.shape
        Cout = weight.shape[1]
        if y16.shape != (N, Cout, 2 * D, 2 * H, 2 * W):
            raise _lib.M355Error(f"c8 slot shape {y16.shape} != op output shape {(N, Cout, 2 * D, 2 * H, 2 * W)}")
        d = _conv_desc(N, Cin, Cout, D, H, W, 2, 2, 0, 0, 0)
        check(L.m355_conv_transpose3d_fwd_h16(C.byref(d), x.ptr(), x.batch_stride(), _p(weight), _p(bias), y16.ptr(),
                                              y16.batch_stride(), x.compute, _stream()), "conv_transpose3d_fwd_h16")
        ctx.desc, ctx.compute, ctx.has_bias = d, x.compute, bias is not None
        ctx.gs = _gs()
        ctx.save_for_backward(x.alias(), weight)
        return y16.alias()

    @staticmethod
    def backward(ctx, dy16):
        L = _lib.lib()
        xa, weight = ctx.saved_tensors
        d, compute = ctx.desc, ctx.compute
        scale = ctx.gs.get(compute)
        dy16, dybs = _c8t(dy16)
        xa, xbs = _c8t(xa)
        S = d.D * d.H * d.W
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dx16 = dw = db = None
        if L.m355_conv_transpose3d_h16_bwd_supported(C.byref(d)):
            if need_x:
                dx16 = torch.empty((d.N, (d.Cin + 7) // 8, S, 8), dtype=_DT16[compute], device=weight.device)
                check(L.m355_conv_transpose3d_bwd_data_h16(C.byref(d), _p(dy16), dybs, _p(weight), _p(dx16), 0, compute,
                                                           _stream()), "conv_transpose3d_bwd_data_h16")
            if need_w:
                dw = torch.empty_like(weight)
                db = torch.empty(d.Cout, dtype=weight.dtype, device=weight.device) if ctx.has_bias else None
                ws = _workspace(L.m355_conv_transpose3d_h16_bwd_workspace(C.byref(d)), weight.device)
                check(L.m355_conv_transpose3d_bwd_weight_h16(C.byref(d), _p(xa), xbs, _p(dy16), dybs, _p(dw), _p(db),
                                                             1.0 / scale, compute, _p(ws), ws.numel(), _stream()),
                      "conv_transpose3d_bwd_weight_h16")
        else:
            dy = _unpack_scaled(dy16, d.Cout, (2 * d.D, 2 * d.H, 2 * d.W), compute, 1.0 / scale)
            ws = _workspace(L.m355_conv_transpose3d_workspace(C.byref(d)), weight.device)
            if need_x:
                dx = torch.empty((d.N, d.Cin, d.D, d.H, d.W), dtype=torch.float32, device=weight.device)
                check(L.m355_conv_transpose3d_bwd_data(C.byref(d), _p(dy), _p(weight), _p(dx), _p(ws), ws.numel(), _stream()),
                      "conv_transpose3d_bwd_data")
                dx16 = _pack_scaled(dx, compute, scale)
            if need_w:
                x = _unpack_scaled(xa, d.Cin, (d.D, d.H, d.W), compute, 1.0)
                dw = torch.empty_like(weight)
                db = torch.empty(d.Cout, dtype=weight.dtype, device=weight.device) if ctx.has_bias else None
                check(L.m355_conv_transpose3d_bwd_weight(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(ws), ws.numel(),
                                                         _stream()), "conv_transpose3d_bwd_weight")
        return dx16, dw, db, None, None


def _act16_tracks(*xs):
    """autograd is recording and one of these c8 activations carries a gradient"""
    return torch.is_grad_enabled() and any(isinstance(x, Act16) and x.requires_grad for x in xs)


# ------------------------------------------------------------------ conv3d
@dataclass
class _ConvMeta:
    k: int
    stride: int
    pad: int
    catbuf: Optional[torch.Tensor] = None
    out: Optional[OutSlot] = None
    tag: str = "conv3d"
    stats: Optional[dict] = None   # filled with {"partials", "slots"} when the statistics are fused
    softmax: bool = False          # Softmax(dim=1) over the output channels fused into this op (out conv + hypothesis)
    h16_train: int = 0             # set by the forward: the 16-bit compute code when the c8 training flow ran


def _conv_desc(N, Cin, Cout, D, H, W, k, stride, pad, xbs, ybs, out_pad=0, compute=0):
    return ConvDesc(N, Cin, Cout, D, H, W, k, stride, pad, out_pad, xbs, ybs, compute, 0)


def conv_plan(desc, which=0):
    """(mfma, ntw, gx, ksplit) of the kernel variant the library picks (profiling only)."""
    out = (C.c_int32 * 4)()
    check(_lib.lib().m355_conv3d_plan(C.byref(desc), which, out), "conv3d_plan")
    return tuple(out)


class _Conv3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, bias, add, meta: _ConvMeta, *parts):
        L = _lib.lib()
        x = meta.catbuf if meta.catbuf is not None else parts[0]
        _require(x, weight, bias, add)
        x, xbs = _dense_channels(x)
        weight = weight.contiguous()
        N, Cin, D, H, W = x.shape
        Cout, k = weight.shape[0], meta.k
        od = lambda n: (n + 2 * meta.pad - k) // meta.stride + 1
        oshape = (N, Cout, od(D), od(H), od(W))
        y = _alloc_out(meta.out, oshape, x)
        y, ybs = _dense_channels(y)
        if add is not None:
            add, abs_ = _dense_channels(add)
            if abs_ != ybs:
                add = add.contiguous()
                if ybs != Cout * oshape[2] * oshape[3] * oshape[4]:
                    raise _lib.M355Error("conv3d: a fused `add` needs the output's batch stride (write to a dense "
                                         "tensor and copy_into the slot instead)")
        d = _conv_desc(N, Cin, Cout, D, H, W, k, meta.stride, meta.pad, xbs, ybs, compute=_COMPUTE[_compute_mode])
        wbuf, flags = _packed_weight(weight, d, 0)
        fuse_sm = meta.softmax and L.m355_conv3d_fuses_softmax(C.byref(d)) != 0
        dk = _with_flags(d, flags | (_lib.CONV_SOFTMAX if fuse_sm else 0))
        # 16-bit training flow: the conv input is packed to c8 ONCE here (the library would stage the same copy
        # internally), consumed by the forward kernel and kept for the weight gradient (m355_conv3d_bwd_weight_h16
        # reads both operands as c8); the fp32 input is then not saved at all
        x16 = None
        if (H16_TRAIN_C8 and d.compute in _DT16 and k == 3 and meta.stride == 1 and meta.pad == 1
                and Cin > 4 and Cout > 4 and not meta.softmax and D * H * W * 64 < 2 ** 31
                and ctx.needs_input_grad[0]):
            x16 = _c8_twin(x, d.compute)
            if x16 is None:
                x16 = pack_act16(x, d.compute)
            meta.h16_train = d.compute
        ws = _workspace(L.m355_conv3d_h16_workspace(C.byref(d), 0) if x16 is not None
                        else L.m355_conv3d_fwd_workspace(C.byref(d)), x.device)
        prof, plan = _prof_gate("conv3d_fwd", d, 0)
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        slots = L.m355_conv3d_stats_slots(C.byref(d)) if meta.stats is not None else 0
        if x16 is not None:
            part = torch.empty((N, slots, Cout, 2), dtype=torch.float32, device=x.device) if slots > 0 else None
            check(L.m355_conv3d_fwd_h16(C.byref(dk), x16.ptr(), x16.batch_stride(), _p(wbuf), _p(bias), _p(add), _p(y),
                                        _p(part), _p(ws), ws.numel(), _stream()), "conv3d_fwd_h16")
            if part is not None:
                meta.stats["partials"], meta.stats["slots"] = part, slots
        elif slots > 0:
            # statistics of the following normalisation fused into the conv epilogue (per-wave partials)
            part = torch.empty((N, slots, Cout, 2), dtype=torch.float32, device=x.device)
            check(L.m355_conv3d_fwd_stats(C.byref(dk), _p(x), _p(wbuf), _p(bias), _p(add), _p(y), _p(part), _p(ws),
                                          ws.numel(), _stream()), "conv3d_fwd_stats")
            meta.stats["partials"], meta.stats["slots"] = part, slots
        else:
            check(L.m355_conv3d_fwd(C.byref(dk), _p(x), _p(wbuf), _p(bias), _p(add), _p(y), _p(ws),
                                    ws.numel(), _stream()), "conv3d_fwd")
        if prof is not None:
            e1.record()
            flops = 2.0 * k ** 3 * Cin * Cout * N * oshape[2] * oshape[3] * oshape[4]
            prof.append(("conv3d_fwd", flops, e0, e1, plan,
                         _conv_bytes(N, Cin, Cout, oshape[2] * oshape[3] * oshape[4], k ** 3, 4, 4)))
        if meta.softmax and not fuse_sm:   # kernel variants without the fused epilogue: a separate softmax pass
            logits = y.contiguous()
            y = torch.empty_like(logits)
            check(L.m355_softmax_fwd(_p(logits), _p(y), N, Cout, 1, oshape[2] * oshape[3] * oshape[4], 0.0, _stream()),
                  "softmax_fwd")
        ctx.meta, ctx.desc = meta, d
        ctx.has_bias, ctx.has_add = bias is not None, add is not None
        ctx.part_channels = [p.shape[1] for p in parts]
        ctx.x16 = None
        if meta.softmax:
            ctx.save_for_backward(x, weight, y)
        elif x16 is not None:
            ctx.x16 = (x16.C, x16.spatial, x16.compute)
            ctx.x_meta = (x.dtype, x.device)
            ctx.save_for_backward(x16.data, weight)
        else:
            ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        meta = ctx.meta
        d = ctx.desc
        if meta.softmax:   # dlogits = y * (dy - sum_c y dy)
            x, weight, y = ctx.saved_tensors
            dy = dy.contiguous()
            dl = torch.empty_like(y)
            check(L.m355_softmax_bwd(_p(y.contiguous()), _p(dy), _p(dl), d.N, d.Cout, 1, y.numel() // (d.N * d.Cout),
                                     _stream()), "softmax_bwd")
            dy = dl
        else:
            x, weight = ctx.saved_tensors
        dy, dybs = _dense_channels(dy)
        if dybs != d.y_batch_stride:
            d = _conv_desc(d.N, d.Cin, d.Cout, d.D, d.H, d.W, d.k, d.stride, d.pad, d.x_batch_stride, dybs, compute=d.compute)
        need_w, need_b, need_add = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        need_x = any(ctx.needs_input_grad[4:])
        dw = db = dadd = None
        dparts: List[Optional[torch.Tensor]] = [None] * len(ctx.part_channels)
        x16 = dy16 = None
        if ctx.x16 is not None:   # 16-bit training flow: `x` is the saved c8 conv input; dy is packed once for both gradients
            x16 = Act16(x, *ctx.x16)
            x_dtype, x_device = ctx.x_meta
            dy16 = _c8_twin(dy, x16.compute)
            if dy16 is None:
                dy16 = pack_act16(dy, x16.compute)
        else:
            x_dtype, x_device = x.dtype, x.device
        if need_w or (need_b and ctx.has_bias):
            dw = torch.empty_like(weight)
            db = torch.empty(d.Cout, dtype=weight.dtype, device=weight.device) if ctx.has_bias else None
            prof, bplan = _prof_gate("conv3d_bwd_weight", d if x16 is None else None, 2)
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if x16 is not None:
                ws = _workspace(L.m355_conv3d_bwd_weight_h16_workspace(C.byref(d)), x_device)
                check(L.m355_conv3d_bwd_weight_h16(C.byref(d), x16.ptr(), x16.batch_stride(), dy16.ptr(), dy16.batch_stride(),
                                                   _p(dy) if db is not None else None, _p(dw), _p(db), _p(ws), ws.numel(),
                                                   _stream()), "conv3d_bwd_weight_h16")
            else:
                ws = _workspace(L.m355_conv3d_bwd_weight_workspace(C.byref(d)), x_device)
                check(L.m355_conv3d_bwd_weight(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(ws), ws.numel(),
                                               _stream()), "conv3d_bwd_weight")
            if prof is not None:
                e1.record()
                vox = dy.shape[2] * dy.shape[3] * dy.shape[4]
                prof.append(("conv3d_bwd_weight", 2.0 * d.k ** 3 * d.Cin * d.Cout * d.N * vox, e0, e1, bplan,
                             _conv_bytes(d.N, d.Cin, d.Cout, vox, d.k ** 3, 4, 4)))
        if need_x:
            # gradient w.r.t. the (possibly concatenated) input, dense, then sliced per part
            dx = torch.empty((d.N, d.Cin, d.D, d.H, d.W), dtype=x_dtype, device=x_device)
            dd = _conv_desc(d.N, d.Cin, d.Cout, d.D, d.H, d.W, d.k, d.stride, d.pad, 0, d.y_batch_stride, compute=d.compute)
            wbuf, flags = _packed_weight(weight, dd, 1)
            ws = _workspace(L.m355_conv3d_h16_workspace(C.byref(dd), 1) if dy16 is not None
                            else L.m355_conv3d_bwd_data_workspace(C.byref(dd)), x_device)
            prof, plan = _prof_gate("conv3d_bwd_data", dd, 1)
            dd = _with_flags(dd, flags)
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if dy16 is not None:
                check(L.m355_conv3d_bwd_data_h16(C.byref(dd), dy16.ptr(), dy16.batch_stride(), _p(wbuf), _p(dx), _p(ws),
                                                 ws.numel(), _stream()), "conv3d_bwd_data_h16")
            else:
                check(L.m355_conv3d_bwd_data(C.byref(dd), _p(dy), _p(wbuf), _p(dx), _p(ws), ws.numel(),
                                             _stream()), "conv3d_bwd_data")
            if prof is not None:
                e1.record()
                vox = dy.shape[2] * dy.shape[3] * dy.shape[4]
                prof.append(("conv3d_bwd_data", 2.0 * d.k ** 3 * d.Cin * d.Cout * d.N * vox, e0, e1, plan,
                             _conv_bytes(d.N, d.Cin, d.Cout, vox, d.k ** 3, 4, 4)))
            c0 = 0
            for i, cc in enumerate(ctx.part_channels):
                if ctx.needs_input_grad[4 + i]:
                    dparts[i] = dx if len(ctx.part_channels) == 1 else dx[:, c0:c0 + cc]
                c0 += cc
        if need_add and ctx.has_add:
            dadd = dy
        return (dw if need_w else None, db if need_b else None, dadd, None, *dparts)


def conv3d(x, weight, bias=None, add=None, stride=1, padding=1, out: Optional[OutSlot] = None,
           stats: Optional[dict] = None, c8_out: bool = False, softmax: bool = False):
    """nn.Conv3d forward (cubic kernel).  `x` is a tensor or a `Concat`; `add` is fused
    into the epilogue (y = conv(x) + bias + add).  `stats`: an empty dict asks the kernel to also
    emit the partial sums the following normalisation needs (filled in when the kernel variant
    supports it; pass it on as `NormCfg.stats`).  `c8_out`: in the c8 flow (x is an `Act16`) return the result
    as an `Act16` written by the conv epilogue (for the normalisation pass that follows).  `softmax`: apply
    nn.Softmax(dim=1) to the result (fused into the conv epilogue where the kernel variant allows)."""
    k = weight.shape[2]
    if not (weight.shape[2] == weight.shape[3] == weight.shape[4]):
        raise NotImplementedError("only cubic kernels are supported")
    c8_parts = None
    if isinstance(x, Concat) and isinstance(x.buf, Act16):
        c8_parts, x = x.parts, x.buf
    if out is not None and out.buf16 is not None:   # c8 flow: the destination slot only exists in c8
        return pack_act16(conv3d(x, weight, bias, add, stride, padding, None, stats), out.buf16.compute, out.act16())
    if isinstance(x, Act16):
        if k == 3 and stride == 1 and padding == 1:
            if torch.is_grad_enabled() and (weight.requires_grad or _act16_tracks(x, *(c8_parts or ()))):
                # c8 training flow (a conv with a fused fp32 `add` keeps its result in fp32)
                return _conv3d_c8_train(x, c8_parts, weight, bias, as_f32(add) if add is not None else None, stats,
                                        c8_out and not softmax and add is None, softmax)
            return _conv3d_act16(x, weight, bias, as_f32(add) if add is not None else None, stats, c8_out and not softmax,
                                 softmax)
        if c8_parts is not None and _act16_tracks(*c8_parts):
            raise NotImplementedError("c8 training flow: only 3x3x3 / stride 1 / padding 1 convolutions read a concat buffer")
        x = x.to_f32()
    if isinstance(x, Concat):
        meta = _ConvMeta(k, stride, padding, catbuf=x.buf, out=out, stats=stats, softmax=softmax)
        y = _Conv3dFn.apply(weight, bias, add, meta, *x.parts)
    else:
        meta = _ConvMeta(k, stride, padding, out=out, stats=stats, softmax=softmax)
        y = _Conv3dFn.apply(weight, bias, add, meta, x)
    if meta.h16_train and add is None:
        # 16-bit training flow: the normalisation that follows may emit its input gradient (= this conv's output
        # gradient) as c8 in the same pass (m355_norm_act_bwd_h16) -- not with a fused residual `add`, whose
        # gradient is the same tensor and goes elsewhere
        y._m355_c8_grad = meta.h16_train
    return y


# -------------------------------------------------------- conv-transpose3d
class _ConvT3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, meta):
        L = _lib.lib()
        _require(x, weight, bias)
        k, stride, pad, out_pad, out = meta
        x, xbs = _dense_channels(x)
        weight = weight.contiguous()
        N, Cin, D, H, W = x.shape
        Cout = weight.shape[1]
        od = lambda n: (n - 1) * stride - 2 * pad + k + out_pad
        oshape = (N, Cout, od(D), od(H), od(W))
        y = _alloc_out(out, oshape, x)
        y, ybs = _dense_channels(y)
        # (the fp32 entry points honour M355_COMPUTE_F32X3 -- the k2 s2 forward on the split kernels -- and ignore the 16-bit codes)
        d = _conv_desc(N, Cin, Cout, D, H, W, k, stride, pad, xbs, ybs, out_pad, compute=_COMPUTE[_compute_mode])
        ws = _workspace(L.m355_conv_transpose3d_workspace(C.byref(d)), x.device)
        check(L.m355_conv_transpose3d_fwd(C.byref(d), _p(x), _p(weight), _p(bias), _p(y), _p(ws), ws.numel(),
                                          _stream()), "conv_transpose3d_fwd")
        ctx.desc, ctx.has_bias = d, bias is not None
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, weight = ctx.saved_tensors
        d = ctx.desc
        dy, dybs = _dense_channels(dy)
        d = _conv_desc(d.N, d.Cin, d.Cout, d.D, d.H, d.W, d.k, d.stride, d.pad, d.x_batch_stride, dybs, d.out_pad, compute=d.compute)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((d.N, d.Cin, d.D, d.H, d.W), dtype=x.dtype, device=x.device)
            dd = _conv_desc(d.N, d.Cin, d.Cout, d.D, d.H, d.W, d.k, d.stride, d.pad, 0, dybs, d.out_pad, compute=d.compute)
            ws = _workspace(L.m355_conv_transpose3d_workspace(C.byref(dd)), x.device)
            check(L.m355_conv_transpose3d_bwd_data(C.byref(dd), _p(dy), _p(weight), _p(dx), _p(ws), ws.numel(),
                                                   _stream()), "conv_transpose3d_bwd_data")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(weight)
            db = torch.empty(d.Cout, dtype=weight.dtype, device=weight.device) if ctx.has_bias else None
            ws = _workspace(L.m355_conv_transpose3d_workspace(C.byref(d)), x.device)
            check(L.m355_conv_transpose3d_bwd_weight(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(ws),
                                                     ws.numel(), _stream()), "conv_transpose3d_bwd_weight")
        return dx, dw, db, None


def conv_transpose3d(x, weight, bias=None, stride=2, padding=0, output_padding=0, out: Optional[OutSlot] = None):
    """nn.ConvTranspose3d forward (cubic kernel, weight [Cin, Cout, k, k, k])."""
    k = weight.shape[2]
    if isinstance(x, Act16):
        if k == 2 and stride == 2 and padding == 0 and output_padding == 0:   # c8 -> c8 kernel
            if torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad):
                y16 = out.act16() if (out is not None and out.buf16 is not None) else None
                if y16 is None:
                    N, Cin, D, H, W = x.shape
                    y16 = Act16.empty(N, weight.shape[1], (2 * D, 2 * H, 2 * W), x.compute, x.device)
                t = _ConvTC8Fn.apply(x.t, weight, bias, x, y16)
                return Act16(y16.data, y16.C, y16.spatial, y16.compute, y16.cb0, t)
            _require(weight, bias)
            N, Cin, D, H, W = x.shape
            Cout = weight.shape[1]
            y16 = out.act16() if (out is not None and out.buf16 is not None) else None
            if y16 is None:
                y16 = Act16.empty(N, Cout, (2 * D, 2 * H, 2 * W), x.compute, x.device)
            d = _conv_desc(N, Cin, Cout, D, H, W, 2, 2, 0, 0, 0)
            check(_lib.lib().m355_conv_transpose3d_fwd_h16(C.byref(d), x.ptr(), x.batch_stride(), _p(weight.contiguous()),
                                                           _p(bias), y16.ptr(), y16.batch_stride(), x.compute, _stream()),
                  "conv_transpose3d_fwd_h16")
            return y16
        y = _ConvT3dFn.apply(x.to_f32(), weight, bias, (k, stride, padding, output_padding, None))
        return _into_c8_slot(y, out, x.compute)
    return _ConvT3dFn.apply(x, weight, bias, (k, stride, padding, output_padding, out))


# ----------------------------------------------------------- norm + activation
@dataclass
class NormCfg:
    groups: int            # 0 = batch norm
    eps: float
    act: int = ACT_NONE
    slope: float = 0.01
    training: bool = True  # BN: use batch statistics (and update running stats)
    momentum: float = 0.1
    running_mean: Optional[torch.Tensor] = None
    running_var: Optional[torch.Tensor] = None
    out: Optional[OutSlot] = None
    stats: Optional[dict] = None   # partial sums from the producing conv (ops.conv3d(..., stats=...))
    c8: int = 0                    # 16-bit no-grad flow: emit the result ONLY in the c8 layout (compute code)
    # 16-bit TRAINING flow (H16_TRAIN_C8), set by norm_act() / its autograd function:
    dx_twin: int = 0               # compute code when the conv that produced x wants its output gradient as c8 too
    twin: Optional["Act16"] = None  # c8 twin of the fp32 result, handed from the forward to norm_act()
    sync: Optional[tuple] = None   # synchronised batch norm: (process group, device-side total element count)


# Synchronised batch norm: while a process group is set here (distributed.PatchParallel(sync_batch_norm=True) does so
# around the forward), BatchNorm layers in training mode take their statistics over the batch of ALL ranks -- the
# reference is one process normalising the whole batch (segmentation_trainer.py:189-262).  Two small all-reduces per
# layer and step: the per-channel sums forward, the per-channel gradient means backward.
class batch_norm_sync:
    """Context manager: BatchNorm training statistics over all ranks of `group` (None = off)."""

    def __init__(self, group):
        self.group = group

    def __enter__(self):
        global _bn_sync_group
        self.prev, _bn_sync_group = _bn_sync_group, self.group
        return self

    def __exit__(self, *exc):
        global _bn_sync_group
        _bn_sync_group = self.prev
        return False


def _norm_statistics(L, d, x, cfg, N, Cc, device=None):
    """mean / rstd of the normalisation: from the conv epilogue partials, from x, or from BN running stats."""
    device = x.device if device is None else device
    ns = L.m355_norm_num_stats(C.byref(d))
    mean = torch.empty(ns, dtype=torch.float32, device=device)
    rstd = torch.empty(ns, dtype=torch.float32, device=device)
    use_batch = cfg.groups > 0 or cfg.training or cfg.running_mean is None
    fused = cfg.stats.get("partials") if cfg.stats else None
    if fused is not None and tuple(fused.shape) != (N, cfg.stats["slots"], Cc, 2):
        raise _lib.M355Error(f"norm_act: statistics partials {tuple(fused.shape)} do not belong to this tensor")
    upd = cfg.groups == 0 and cfg.training and use_batch
    cfg.sync = None
    if upd and _bn_sync_group is not None:
        import torch.distributed as dist
        ws = _workspace(L.m355_norm_workspace(C.byref(d)), device)
        sums = torch.empty(2 * Cc + 1, dtype=torch.float64, device=device)   # {sum, sum of squares} per channel, count
        check(L.m355_norm_sums(C.byref(d), _p(x) if fused is None else None, _p(fused),
                               int(cfg.stats["slots"]) if fused is not None else 0, _p(sums), _p(ws), ws.numel(),
                               _stream()), "norm_sums")
        dist.all_reduce(sums, group=_bn_sync_group)
        check(L.m355_norm_stats_from_sums(C.byref(d), _p(sums), _p(mean), _p(rstd), _p(cfg.running_mean),
                                          _p(cfg.running_var), float(cfg.momentum), _stream()), "norm_stats_from_sums")
        cfg.sync = (_bn_sync_group, sums[2 * Cc:])
        return mean, rstd, use_batch
    if use_batch and fused is not None:
        ws = _workspace(L.m355_norm_workspace(C.byref(d)), device)
        check(L.m355_norm_stats_from_partials(C.byref(d), _p(fused), int(cfg.stats["slots"]), _p(mean), _p(rstd),
                                              _p(cfg.running_mean) if upd else None,
                                              _p(cfg.running_var) if upd else None,
                                              float(cfg.momentum), _p(ws), ws.numel(), _stream()),
              "norm_stats_from_partials")
    elif use_batch:
        ws = _workspace(L.m355_norm_workspace(C.byref(d)), x.device)
        check(L.m355_norm_stats(C.byref(d), _p(x), _p(mean), _p(rstd),
                                _p(cfg.running_mean) if upd else None,
                                _p(cfg.running_var) if upd else None,
                                float(cfg.momentum), _p(ws), ws.numel(), _stream()), "norm_stats")
    else:
        check(L.m355_norm_stats_from_running(C.byref(d), _p(cfg.running_mean), _p(cfg.running_var),
                                             _p(mean), _p(rstd), _stream()), "norm_stats_from_running")
    return mean, rstd, use_batch


def _norm_act_c8(x, gamma, beta, add, cfg: NormCfg) -> Act16:
    """norm + activation (+ residual) whose ONLY output is c8 (16-bit no-grad flow).  x: the fp32 conv output
    (read once, 2 bytes per element written) or the c8 pre-norm tensor of m355_conv3d_fwd_h16_c8 (2 + 2 bytes)."""
    L = _lib.lib()
    _require(gamma, beta)
    x_c8 = isinstance(x, Act16)
    if x_c8:
        N, Cc, spatial, S, xbs = x.shape[0], x.C, x.spatial, x.S, 0
    else:
        _require(x)
        x, xbs = _dense_channels(x)
        N, Cc, spatial = x.shape[0], x.shape[1], tuple(x.shape[2:])
        S = spatial[0] * spatial[1] * spatial[2]
    out16 = cfg.out.act16() if cfg.out is not None else None
    if out16 is None:
        out16 = Act16.empty(N, Cc, spatial, cfg.c8, gamma.device if gamma is not None else (x.device))
    elif out16.shape != (N, Cc) + tuple(spatial):
        raise _lib.M355Error(f"c8 slot shape {out16.shape} != op output shape {(N, Cc) + tuple(spatial)}")
    abs_ = 0
    if add is not None and not x_c8:
        add, abs_ = _dense_channels(as_f32(add))
        _require(add)
    d = NormDesc(N, Cc, S, cfg.groups, cfg.act, cfg.eps, cfg.slope, xbs, 0, abs_)
    if x_c8:
        use_batch = cfg.groups > 0 or cfg.training or cfg.running_mean is None
        if use_batch and not (cfg.stats and cfg.stats.get("partials") is not None):
            cfg.stats = {} if cfg.stats is None else cfg.stats
            _c8_channel_partials(x, cfg.stats)
        mean, rstd, _ = _norm_statistics(L, d, None, cfg, N, Cc, device=x.device)
        add16 = None
        if add is not None:
            add16 = add if isinstance(add, Act16) else pack_act16(add, cfg.c8)
        check(L.m355_norm_act_fwd_c8(C.byref(d), x.ptr(), x.batch_stride(), _p(mean), _p(rstd), _p(gamma), _p(beta),
                                     add16.ptr() if add16 is not None else None,
                                     add16.batch_stride() if add16 is not None else 0, out16.ptr(),
                                     out16.batch_stride(), cfg.c8, _stream()), "norm_act_fwd_c8")
        return out16
    mean, rstd, _ = _norm_statistics(L, d, x, cfg, N, Cc)
    check(L.m355_norm_act_fwd_h16(C.byref(d), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(add), None,
                                  out16.ptr(), out16.batch_stride(), cfg.c8, _stream()), "norm_act_fwd_h16")
    return out16


def _norm_backward(L, d, x, dy, mean, rstd, gamma, beta, dx, dgamma, dbeta, batch_stats, sync, dx16=None, compute=0):
    """The normalisation backward; with `sync` (synchronised batch norm) as its two halves around the all-reduce of the
    per-channel gradient means."""
    ws = _workspace(L.m355_norm_workspace(C.byref(d)), x.device)
    training = 1 if batch_stats else 0
    if sync is not None:
        import torch.distributed as dist
        group, total = sync
        stat_m = torch.empty(2 * d.C, dtype=torch.float32, device=x.device)
        check(L.m355_norm_act_bwd_reduce(C.byref(d), _p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dgamma),
                                         _p(dbeta), training, _p(total), _p(stat_m), _p(ws), ws.numel(), _stream()),
              "norm_act_bwd_reduce")
        dist.all_reduce(stat_m, group=group)
        check(L.m355_norm_act_bwd_apply(C.byref(d), _p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(stat_m),
                                        _p(dx), dx16.ptr() if dx16 is not None else None,
                                        dx16.batch_stride() if dx16 is not None else 0, compute, _stream()),
              "norm_act_bwd_apply")
    elif dx16 is not None:
        check(L.m355_norm_act_bwd_h16(C.byref(d), _p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dx),
                                      _p(dgamma), _p(dbeta), training, dx16.ptr(), dx16.batch_stride(), compute,
                                      _p(ws), ws.numel(), _stream()), "norm_act_bwd_h16")
    else:
        check(L.m355_norm_act_bwd(C.byref(d), _p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dx),
                                  _p(dgamma), _p(dbeta), training, _p(ws), ws.numel(), _stream()), "norm_act_bwd")


class _NormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, add, cfg: NormCfg):
        L = _lib.lib()
        _require(x, gamma, beta, add)
        x, xbs = _dense_channels(x)
        N, Cc = x.shape[0], x.shape[1]
        S = x.shape[2] * x.shape[3] * x.shape[4]
        if xbs != Cc * S:  # keep the saved input dense so the backward writes a dense dx
            x, xbs = x.contiguous(), Cc * S
        y = _alloc_out(cfg.out, x.shape, x)
        y, ybs = _dense_channels(y)
        abs_ = 0
        if add is not None:
            add, abs_ = _dense_channels(add)
        d = NormDesc(N, Cc, S, cfg.groups, cfg.act, cfg.eps, cfg.slope, xbs, ybs, abs_)
        mean, rstd, use_batch = _norm_statistics(L, d, x, cfg, N, Cc)
        # 16-bit training flow (H16_TRAIN_C8): a dense output also leaves as a c8 twin for the convolution that
        # usually consumes it (handed over through `cfg.twin`, attached to the returned tensor by norm_act());
        # ctx.dx_twin: the convolution that produced x wants its output gradient in c8 as well (see backward)
        compute = _COMPUTE[_compute_mode]
        ctx.dx_twin = cfg.dx_twin
        cfg.twin = None
        if H16_TRAIN_C8 and compute in _DT16 and cfg.out is None and ybs == Cc * S and Cc > 4:
            cfg.twin = Act16.empty(N, Cc, tuple(x.shape[2:]), compute, x.device)
            check(L.m355_norm_act_fwd_h16(C.byref(d), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(add), _p(y),
                                          cfg.twin.ptr(), cfg.twin.batch_stride(), compute, _stream()), "norm_act_fwd_h16")
        else:
            check(L.m355_norm_act_fwd(C.byref(d), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(add), _p(y),
                                      _stream()), "norm_act_fwd")
        ctx.desc, ctx.batch_stats, ctx.has_add = d, use_batch, add is not None
        ctx.has_affine, ctx.sync = gamma is not None, cfg.sync
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        d0 = ctx.desc
        dy, dybs = _dense_channels(dy)
        d = NormDesc(d0.N, d0.C, d0.S, d0.groups, d0.act, d0.eps, d0.act_slope, d0.x_batch_stride, dybs, 0)
        dx = torch.empty_like(x)  # x was saved dense
        dgamma = torch.empty(d0.C, dtype=torch.float32, device=x.device) if ctx.has_affine else None
        dbeta = torch.empty(d0.C, dtype=torch.float32, device=x.device) if ctx.has_affine else None
        if ctx.dx_twin:   # dx is the output gradient of an h16-flow convolution: emit it as c8 too, in the same pass
            dx16 = Act16.empty(d0.N, d0.C, tuple(x.shape[2:]), ctx.dx_twin, x.device)
            _norm_backward(L, d, x, dy, mean, rstd, gamma, beta, dx, dgamma, dbeta, ctx.batch_stats, ctx.sync, dx16,
                           ctx.dx_twin)
            dx._m355_c8 = (dx16, dx._version)
        else:
            _norm_backward(L, d, x, dy, mean, rstd, gamma, beta, dx, dgamma, dbeta, ctx.batch_stats, ctx.sync)
        dadd = dy if (ctx.has_add and ctx.needs_input_grad[3]) else None
        return dx, dgamma, dbeta, dadd, None


class _NormActPoolFn(torch.autograd.Function):
    """norm + activation with AvgPool3d(2, 2) of the result as a second output (m355_norm_act_pool_fwd): the last
    pass of an encoder block also produces the next level's input, so the pool never re-reads the activated tensor.
    Backward: the two incoming gradients (skip path, pooled path) are summed in the pool-backward pass
    (m355_avgpool3d_2x_bwd_add), then the usual normalisation backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, cfg: NormCfg):
        L = _lib.lib()
        _require(x, gamma, beta)
        x, xbs = _dense_channels(x)
        N, Cc, D, H, W = x.shape
        S = D * H * W
        if xbs != Cc * S:
            x, xbs = x.contiguous(), Cc * S
        y = _alloc_out(cfg.out, x.shape, x)
        y, ybs = _dense_channels(y)
        pooled = torch.empty((N, Cc, D // 2, H // 2, W // 2), dtype=x.dtype, device=x.device)
        d = NormDesc(N, Cc, S, cfg.groups, cfg.act, cfg.eps, cfg.slope, xbs, ybs, 0)
        mean, rstd, use_batch = _norm_statistics(L, d, x, cfg, N, Cc)
        check(L.m355_norm_act_pool_fwd(C.byref(d), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(y), _p(pooled), 0,
                                       D, H, W, _stream()), "norm_act_pool_fwd")
        ctx.desc, ctx.batch_stats, ctx.has_affine, ctx.dims = d, use_batch, gamma is not None, (D, H, W)
        ctx.sync = cfg.sync
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        return y, pooled

    @staticmethod
    def backward(ctx, dy, dpool):
        L = _lib.lib()
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        d0 = ctx.desc
        D, H, W = ctx.dims
        if dpool is not None:   # gradient of the activated tensor = skip-path gradient + un-pooled gradient, one pass
            dpool, gpbs = _dense_channels(dpool)
            tot = torch.empty_like(x)
            if dy is None:
                check(L.m355_avgpool3d_2x_bwd(_p(dpool), _p(tot), d0.N, d0.C, D, H, W, gpbs, 0, _stream()), "avgpool3d_2x_bwd")
            else:
                dy, gsbs = _dense_channels(dy)
                check(L.m355_avgpool3d_2x_bwd_add(_p(dpool), _p(dy), _p(tot), d0.N, d0.C, D, H, W, gpbs, gsbs, 0,
                                                  _stream()), "avgpool3d_2x_bwd_add")
            dy = tot
        dy, dybs = _dense_channels(dy)
        d = NormDesc(d0.N, d0.C, d0.S, d0.groups, d0.act, d0.eps, d0.act_slope, d0.x_batch_stride, dybs, 0)
        dx = torch.empty_like(x)
        dgamma = torch.empty(d0.C, dtype=torch.float32, device=x.device) if ctx.has_affine else None
        dbeta = torch.empty(d0.C, dtype=torch.float32, device=x.device) if ctx.has_affine else None
        _norm_backward(L, d, x, dy, mean, rstd, gamma, beta, dx, dgamma, dbeta, ctx.batch_stats, ctx.sync)
        return dx, dgamma, dbeta, None


def norm_act_pool(x, gamma, beta, cfg: NormCfg):
    """act(norm(x)) AND AvgPool3d(2, 2) of it from one pass -> (y, pooled); fp32 tensors, even spatial sizes."""
    return _NormActPoolFn.apply(x, gamma, beta, cfg)


def norm_act(x, gamma, beta, cfg: NormCfg, add=None, pool=False):
    """normalization_class + activation_class of Block3d (components.py:52-55), optionally
    fused with the residual sum (components.py:67-68): act(norm(x)) + add.  With `cfg.c8` set (16-bit
    precision mode under no_grad) the result is an `Act16` and nothing is written in fp32."""
    if cfg.c8 and torch.is_grad_enabled():                                   # c8 training flow
        if not isinstance(x, Act16):     # the pre-norm tensor of a conv without a c8 kernel (strided / Blur): joins here
            x = pack_act16(x, cfg.c8)
        return _norm_act_c8_train(x, gamma, beta, add, cfg, pool=pool)
    if pool:
        raise _lib.M355Error("norm_act(pool=True) is the c8 training flow's fused pool; use norm_act_pool for fp32 tensors")
    if cfg.c8 and not torch.is_grad_enabled():
        return _norm_act_c8(x, gamma, beta, add, cfg)
    x = as_f32(x)
    cfg.dx_twin = getattr(x, "_m355_c8_grad", 0) if torch.is_grad_enabled() else 0
    y = _NormActFn.apply(x, gamma, beta, as_f32(add) if add is not None else None, cfg)
    if cfg.twin is not None:
        # the twin describes y as written by this pass: recorded with y's version, so an in-place op on y voids it
        y._m355_c8 = (cfg.twin, y._version)
        cfg.twin = None
    return y


# ------------------------------------------------------------- pool / upsample
class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out):
        L = _lib.lib()
        _require(x)
        x, xbs = _dense_channels(x)
        N, Cc, D, H, W = x.shape
        y = _alloc_out(out, (N, Cc, D // 2, H // 2, W // 2), x)
        y, ybs = _dense_channels(y)
        check(L.m355_avgpool3d_2x_fwd(_p(x), _p(y), N, Cc, D, H, W, xbs, ybs, _stream()), "avgpool3d_2x_fwd")
        ctx.shape = (N, Cc, D, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        N, Cc, D, H, W = ctx.shape
        dy, dybs = _dense_channels(dy)
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        check(L.m355_avgpool3d_2x_bwd(_p(dy), _p(dx), N, Cc, D, H, W, dybs, 0, _stream()), "avgpool3d_2x_bwd")
        return dx, None


class _PoolSkipFn(torch.autograd.Function):
    """AvgPool3d(2, 2) of a tensor that ALSO continues as a skip connection (models/modular_unet.py:90-92).
    Returns (skip alias, pooled); the backward receives both gradients together and sums them in the pool-backward
    pass (m355_avgpool3d_2x_bwd_add) -- autograd would otherwise add them with a separate torch kernel."""

    @staticmethod
    def forward(ctx, x):
        L = _lib.lib()
        _require(x)
        xd, xbs = _dense_channels(x)
        N, Cc, D, H, W = xd.shape
        y = torch.empty((N, Cc, D // 2, H // 2, W // 2), dtype=x.dtype, device=x.device)
        check(L.m355_avgpool3d_2x_fwd(_p(xd), _p(y), N, Cc, D, H, W, xbs, 0, _stream()), "avgpool3d_2x_fwd")
        ctx.shape = (N, Cc, D, H, W)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, g_skip, g_pool):
        L = _lib.lib()
        N, Cc, D, H, W = ctx.shape
        if g_pool is None:
            return g_skip
        g_pool, gpbs = _dense_channels(g_pool)
        dx = torch.empty(ctx.shape, dtype=g_pool.dtype, device=g_pool.device)
        if g_skip is None:
            check(L.m355_avgpool3d_2x_bwd(_p(g_pool), _p(dx), N, Cc, D, H, W, gpbs, 0, _stream()), "avgpool3d_2x_bwd")
        else:
            g_skip, gsbs = _dense_channels(g_skip)
            check(L.m355_avgpool3d_2x_bwd_add(_p(g_pool), _p(g_skip), _p(dx), N, Cc, D, H, W, gpbs, gsbs, 0, _stream()),
                  "avgpool3d_2x_bwd_add")
        return dx


def avgpool3d_2x_with_skip(x, out: Optional[OutSlot] = None):
    """-> (x as it continues into the skip connection, AvgPool3d(2, 2)(x)); see _PoolSkipFn.  `out` (c8 flow): the pooled
    tensor is written into this concat slot."""
    if isinstance(x, Act16):
        N, Cc, D, H, W = x.shape
        y16 = out.act16() if out is not None else None
        if y16 is None:
            y16 = Act16.empty(N, Cc, (D // 2, H // 2, W // 2), x.compute, x.device)
        elif y16.shape != (N, Cc, D // 2, H // 2, W // 2):
            raise _lib.M355Error(f"c8 slot shape {y16.shape} != op output shape {(N, Cc, D // 2, H // 2, W // 2)}")
        skip_t, pooled_t = _PoolC8Fn.apply(x.t, x, y16, True)
        return (Act16(x.data, x.C, x.spatial, x.compute, x.cb0, skip_t),
                Act16(y16.data, y16.C, y16.spatial, y16.compute, y16.cb0, pooled_t))
    return _PoolSkipFn.apply(x)


def avgpool3d_2x(x, out: Optional[OutSlot] = None):
    """nn.AvgPool3d(kernel_size=2, stride=2, count_include_pad=False); c8 -> c8 in the 16-bit no-grad flow."""
    if isinstance(x, Act16):
        N, Cc, D, H, W = x.shape
        y16 = out.act16() if out is not None else None
        if y16 is None:
            y16 = Act16.empty(N, Cc, (D // 2, H // 2, W // 2), x.compute, x.device)
        if _act16_tracks(x):
            return Act16(y16.data, y16.C, y16.spatial, y16.compute, y16.cb0, _PoolC8Fn.apply(x.t, x, y16, False))
        check(_lib.lib().m355_avgpool3d_2x_fwd_h16(x.ptr(), y16.ptr(), N, Cc, D, H, W, x.batch_stride(),
                                                   y16.batch_stride(), x.compute, _stream()), "avgpool3d_2x_fwd_h16")
        return y16
    return _AvgPoolFn.apply(x, out)


class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out):
        L = _lib.lib()
        _require(x)
        x, xbs = _dense_channels(x)
        N, Cc, D, H, W = x.shape
        y = _alloc_out(out, (N, Cc, 2 * D, 2 * H, 2 * W), x)
        y, ybs = _dense_channels(y)
        check(L.m355_upsample_trilinear2x_fwd(_p(x), _p(y), N, Cc, D, H, W, xbs, ybs, _stream()),
              "upsample_trilinear2x_fwd")
        ctx.shape = (N, Cc, D, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        N, Cc, D, H, W = ctx.shape
        dy, dybs = _dense_channels(dy)
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        check(L.m355_upsample_trilinear2x_bwd(_p(dy), _p(dx), N, Cc, D, H, W, dybs, 0, _stream()),
              "upsample_trilinear2x_bwd")
        return dx, None


def _into_c8_slot(y32, out: Optional[OutSlot], compute):
    """c8 flow: a producer without a c8 epilogue (conv-transpose, trilinear upsampling) computed `y32`
    densely; pack it into its slot of the c8 concat buffer."""
    return pack_act16(y32, compute, out.act16() if out is not None else None)


class _UpsampleC8Fn(torch.autograd.Function):
    """trilinear x2 upsampling c8 -> c8 (straight into its concat slot); backward: the gather-form kernel on the c8
    gradient"""

    @staticmethod
    def forward(ctx, x_t, x: Act16, y16: Act16):
        N, Cc, D, H, W = x.shape
        check(_lib.lib().m355_upsample_trilinear2x_fwd_h16(x.ptr(), y16.ptr(), N, Cc, D, H, W, x.batch_stride(),
                                                           y16.batch_stride(), x.compute, _stream()),
              "upsample_trilinear2x_fwd_h16")
        ctx.info = (N, Cc, D, H, W, x.compute)
        return y16.alias()

    @staticmethod
    def backward(ctx, dy16):
        N, Cc, D, H, W, compute = ctx.info
        dy16, dybs = _c8t(dy16)
        dx16 = torch.empty((N, (Cc + 7) // 8, D * H * W, 8), dtype=_DT16[compute], device=dy16.device)
        check(_lib.lib().m355_upsample_trilinear2x_bwd_h16(_p(dy16), _p(dx16), N, Cc, D, H, W, dybs, 0, compute, _stream()),
              "upsample_trilinear2x_bwd_h16")
        return dx16, None, None


def upsample_trilinear2x(x, out: Optional[OutSlot] = None):
    """nn.Upsample(scale_factor=2, mode='trilinear', align_corners=True); c8 -> c8 in the 16-bit flows."""
    if isinstance(x, Act16):
        N, Cc, D, H, W = x.shape
        y16 = out.act16() if out is not None else None
        if y16 is None:
            y16 = Act16.empty(N, Cc, (2 * D, 2 * H, 2 * W), x.compute, x.device)
        elif y16.shape != (N, Cc, 2 * D, 2 * H, 2 * W):
            raise _lib.M355Error(f"c8 slot shape {y16.shape} != op output shape {(N, Cc, 2 * D, 2 * H, 2 * W)}")
        if _act16_tracks(x):
            return Act16(y16.data, y16.C, y16.spatial, y16.compute, y16.cb0, _UpsampleC8Fn.apply(x.t, x, y16))
        check(_lib.lib().m355_upsample_trilinear2x_fwd_h16(x.ptr(), y16.ptr(), N, Cc, D, H, W, x.batch_stride(),
                                                           y16.batch_stride(), x.compute, _stream()),
              "upsample_trilinear2x_fwd_h16")
        return y16
    return _UpsampleFn.apply(x, out)


# ------------------------------------------------------------------- softmax
class _SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inner, diag_bias):
        L = _lib.lib()
        _require(x)
        x = x.contiguous()
        N, Ctot = x.shape[0], x.shape[1]
        Cc = Ctot // inner
        S = x.numel() // (N * Ctot)
        y = torch.empty_like(x)
        check(L.m355_softmax_fwd(_p(x), _p(y), N, Cc, inner, S, float(diag_bias), _stream()), "softmax_fwd")
        ctx.dims = (N, Cc, inner, S)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        (y,) = ctx.saved_tensors
        N, Cc, inner, S = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        check(L.m355_softmax_bwd(_p(y), _p(dy), _p(dx), N, Cc, inner, S, _stream()), "softmax_bwd")
        return dx, None, None


def softmax_channels(x, inner=1, diag_bias=0.0):
    """nn.Softmax(dim=1); inner=C gives StochasticMatrix's reshape+eye-bias+softmax."""
    return _SoftmaxFn.apply(as_f32(x), inner, diag_bias)


# ----------------------------------------------------------------------- loss
class _HybridLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, t, dice_weight, class_weights, square_dice):
        L = _lib.lib()
        _require(p, t, class_weights)
        if p.shape != t.shape:
            raise _lib.M355Error(f"prediction {tuple(p.shape)} and target {tuple(t.shape)} differ in shape")
        p, t = p.contiguous(), t.contiguous()
        N, Cc = p.shape[0], p.shape[1]
        S = p.numel() // (N * Cc)
        out3 = torch.empty(3, dtype=torch.float32, device=p.device)
        sums = torch.empty(N * Cc * 4, dtype=torch.float32, device=p.device)
        ws = _workspace(L.m355_hybrid_loss_workspace(N, Cc, S), p.device)
        check(L.m355_hybrid_loss_fwd(_p(p), _p(t), N, Cc, S, float(dice_weight), _p(class_weights),
                                     1 if square_dice else 0, _p(out3), _p(sums), _p(ws), ws.numel(),
                                     _stream()), "hybrid_loss_fwd")
        ctx.args = (N, Cc, S, float(dice_weight), 1 if square_dice else 0)
        ctx.save_for_backward(p, t, sums, class_weights)
        # three independent 0-dim tensors for autograd's sake: not views of `out3` (a Function must not return views of one
        # base) and not clones either (three 4-byte device copies of ~10 us each per step) -- fresh tensor objects over the
        # same storage, as ops._alloc_out makes them for concat slots
        st = out3.untyped_storage()
        loss, dice, logistic = (torch.empty((), dtype=torch.float32, device=p.device).set_(st, i, ()) for i in range(3))
        ctx.mark_non_differentiable(dice, logistic)
        return loss, dice, logistic

    @staticmethod
    def backward(ctx, dloss, _d1, _d2):
        L = _lib.lib()
        p, t, sums, cw = ctx.saved_tensors
        N, Cc, S, dwt, sq = ctx.args
        dloss = dloss.contiguous().to(torch.float32)
        dp = torch.empty_like(p)
        check(L.m355_hybrid_loss_bwd(_p(p), _p(t), _p(sums), _p(dloss), N, Cc, S, dwt, _p(cw), sq, _p(dp),
                                     _stream()), "hybrid_loss_bwd")
        return dp, None, None, None, None


def hybrid_logistic_dice_loss(prediction, target, dice_weight=0.5, class_weights=None, square_dice=True):
    """HybridLogisticDiceLoss.forward -> (loss, dice_loss, logistic_loss); the gradient
    flows through `loss` (the only entry the trainer back-propagates, segmentation_trainer.py:177)."""
    return _HybridLossFn.apply(prediction, target, dice_weight, class_weights, square_dice)


# ---------------------------------------------------------------- elementwise
class _ChannelScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        L = _lib.lib()
        _require(x, scale)
        x = x.contiguous()
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        y = torch.empty_like(x)
        check(L.m355_channel_scale(_p(x), _p(scale), _p(y), N, Cc, S, _stream()), "channel_scale")
        ctx.save_for_backward(scale)
        ctx.dims = (N, Cc, S)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        (scale,) = ctx.saved_tensors
        N, Cc, S = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(L.m355_channel_scale(_p(dy), _p(scale), _p(dx), N, Cc, S, _stream()), "channel_scale")
        return dx, None


def _channel_scale_c8(src_ptr, sbs, scale, dst_ptr, dbs, N, Cc, S, compute):
    check(_lib.lib().m355_act16_channel_scale(src_ptr, _p(scale), dst_ptr, N, Cc, S, sbs, dbs, compute, _stream()),
          "act16_channel_scale")


class _ChannelScaleC8Fn(torch.autograd.Function):
    """Dropout3d with a pre-drawn mask on a c8 activation (into its concat slot); backward: the same kernel on the c8
    gradient"""

    @staticmethod
    def forward(ctx, x_t, scale, x: Act16, y16: Act16):
        N, Cc = x.shape[:2]
        _channel_scale_c8(x.ptr(), x.batch_stride(), scale, y16.ptr(), y16.batch_stride(), N, Cc, x.S, x.compute)
        ctx.info = (N, Cc, x.S, x.compute)
        ctx.save_for_backward(scale)
        return y16.alias()

    @staticmethod
    def backward(ctx, dy16):
        (scale,) = ctx.saved_tensors
        N, Cc, S, compute = ctx.info
        dy16, dybs = _c8t(dy16)
        dx16 = torch.empty((N, (Cc + 7) // 8, S, 8), dtype=_DT16[compute], device=dy16.device)
        _channel_scale_c8(_p(dy16), dybs, scale, _p(dx16), 0, N, Cc, S, compute)
        return dx16, None, None, None


def channel_scale(x, scale, out: Optional[OutSlot] = None):
    """y[n,c,...] = x[n,c,...] * scale[n,c]  (Dropout3d with a pre-drawn mask); `out`: written into this concat slot.
    A c8 activation stays c8."""
    scale = scale.contiguous().view(-1)
    if isinstance(x, Act16):
        _require(scale)
        y16 = out.act16() if out is not None else None
        if y16 is None:
            y16 = Act16.empty(x.shape[0], x.C, x.spatial, x.compute, x.device)
        elif y16.shape != x.shape:
            raise _lib.M355Error(f"c8 slot shape {y16.shape} != op output shape {x.shape}")
        if _act16_tracks(x):
            return Act16(y16.data, y16.C, y16.spatial, y16.compute, y16.cb0, _ChannelScaleC8Fn.apply(x.t, scale, x, y16))
        _channel_scale_c8(x.ptr(), x.batch_stride(), scale, y16.ptr(), y16.batch_stride(), x.shape[0], x.C, x.S, x.compute)
        return y16
    y = _ChannelScaleFn.apply(x, scale)
    return y if out is None else copy_into(y, out)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        L = _lib.lib()
        _require(a, b)
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        check(L.m355_add(_p(a), _p(b), _p(y), a.numel(), _stream()), "add")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _AddFn.apply(as_f32(a), as_f32(b))


class _CopyIntoFn(torch.autograd.Function):
    """Copy a tensor into a concat slot (the generic torch.cat path for producers
    that cannot write into the buffer themselves)."""

    @staticmethod
    def forward(ctx, x, out):
        L = _lib.lib()
        _require(x)
        x, xbs = _dense_channels(x)
        N, Cc = x.shape[0], x.shape[1]
        S = x.shape[2] * x.shape[3] * x.shape[4]
        y = _alloc_out(out, x.shape, x)
        y, ybs = _dense_channels(y)
        check(L.m355_copy_channels(_p(x), _p(y), N, Cc, S, xbs, ybs, _stream()), "copy_channels")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None


def copy_into(x, out: OutSlot):
    if out.buf16 is not None:   # c8 flow: the slot lives in the c8 twin of the concat buffer
        return pack_act16(as_f32(x), out.buf16.compute, out.act16())
    return _CopyIntoFn.apply(as_f32(x), out)


class _S2DFn(torch.autograd.Function):
    """space-to-depth (to_depth=True) or depth-to-space by 2; each is the other's backward."""

    @staticmethod
    def forward(ctx, x, to_depth, out):
        L = _lib.lib()
        _require(x)
        x, xbs = _dense_channels(x)
        N, Cc, D, H, W = x.shape
        if to_depth:
            full = (N, Cc, D, H, W)
            y = _alloc_out(out, (N, Cc * 8, D // 2, H // 2, W // 2), x)
        else:
            full = (N, Cc // 8, 2 * D, 2 * H, 2 * W)
            y = _alloc_out(out, full, x)
        y, ybs = _dense_channels(y)
        fn = L.m355_space_to_depth2 if to_depth else L.m355_depth_to_space2
        check(fn(_p(x), _p(y), full[0], full[1], full[2], full[3], full[4], xbs, ybs, _stream()),
              "space_to_depth2" if to_depth else "depth_to_space2")
        ctx.to_depth, ctx.full = to_depth, full
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        dy, dybs = _dense_channels(dy)
        N, Cf, D, H, W = ctx.full
        if ctx.to_depth:   # gradient of s2d is d2s
            dx = torch.empty(ctx.full, dtype=dy.dtype, device=dy.device)
            check(L.m355_depth_to_space2(_p(dy), _p(dx), N, Cf, D, H, W, dybs, 0, _stream()), "depth_to_space2")
        else:
            dx = torch.empty((N, Cf * 8, D // 2, H // 2, W // 2), dtype=dy.dtype, device=dy.device)
            check(L.m355_space_to_depth2(_p(dy), _p(dx), N, Cf, D, H, W, dybs, 0, _stream()), "space_to_depth2")
        return dx, None, None


def _s2d_c8(src_ptr, sbs, dst_ptr, dbs, full_shape, compute, to_depth):
    """c8 -> c8; `full_shape` = (N, C, D, H, W) of the FULL-resolution tensor"""
    N, Cc, D, H, W = full_shape
    L = _lib.lib()
    fn = L.m355_space_to_depth2_h16 if to_depth else L.m355_depth_to_space2_h16
    check(fn(src_ptr, dst_ptr, N, Cc, D, H, W, sbs, dbs, compute, _stream()),
          "space_to_depth2_h16" if to_depth else "depth_to_space2_h16")


class _S2DC8Fn(torch.autograd.Function):
    """space-to-depth (to_depth=True) or depth-to-space by 2 on c8 activations; each is the other's backward (on the c8
    gradient)"""

    @staticmethod
    def forward(ctx, x_t, x: Act16, y16: Act16, to_depth: bool):
        full = x.shape if to_depth else y16.shape
        _s2d_c8(x.ptr(), x.batch_stride(), y16.ptr(), y16.batch_stride(), full, x.compute, to_depth)
        ctx.info = (full, x.compute, to_depth, x.shape)
        return y16.alias()

    @staticmethod
    def backward(ctx, dy16):
        full, compute, to_depth, xshape = ctx.info
        dy16, dybs = _c8t(dy16)
        S = xshape[2] * xshape[3] * xshape[4]
        dx16 = torch.empty((xshape[0], (xshape[1] + 7) // 8, S, 8), dtype=_DT16[compute], device=dy16.device)
        _s2d_c8(_p(dy16), dybs, _p(dx16), 0, full, compute, not to_depth)
        return dx16, None, None, None


def _s2d_act16(x: Act16, to_depth: bool, out: Optional[OutSlot]):
    N, Cc, D, H, W = x.shape
    if to_depth:
        oshape = (N, Cc * 8, D // 2, H // 2, W // 2)
    else:
        if Cc % 8:
            raise _lib.M355Error(f"depth_to_space2: {Cc} channels are not 8 parities per output channel")
        oshape = (N, Cc // 8, 2 * D, 2 * H, 2 * W)
    y16 = out.act16() if out is not None else None
    if y16 is None:
        y16 = Act16.empty(oshape[0], oshape[1], oshape[2:], x.compute, x.device)
    elif y16.shape != oshape:
        raise _lib.M355Error(f"c8 slot shape {y16.shape} != op output shape {oshape}")
    if _act16_tracks(x):
        return Act16(y16.data, y16.C, y16.spatial, y16.compute, y16.cb0, _S2DC8Fn.apply(x.t, x, y16, to_depth))
    _s2d_c8(x.ptr(), x.batch_stride(), y16.ptr(), y16.batch_stride(), x.shape if to_depth else oshape, x.compute, to_depth)
    return y16


def space_to_depth2(x):
    """[N, C, D, H, W] -> [N, 8C, D/2, H/2, W/2], channel c*8 + (pz*4 + py*2 + px).  A c8 activation stays c8 (the packed
    channels of input channel c are exactly c8 block c)."""
    if isinstance(x, Act16) and all(v % 2 == 0 for v in x.shape[2:]):
        return _s2d_act16(x, True, None)
    return _S2DFn.apply(as_f32(x), True, None)


def depth_to_space2(x, out: Optional[OutSlot] = None):
    if isinstance(x, Act16) and x.C % 8 == 0 and (out is None or out.buf16 is not None):
        return _s2d_act16(x, False, out)
    if out is not None and out.buf16 is not None:
        return pack_act16(_S2DFn.apply(as_f32(x), False, None), out.buf16.compute, out.act16())
    return _S2DFn.apply(as_f32(x), False, out)


class _BlurWeightFn(torch.autograd.Function):
    """Blur-convolution weight transform (reference models/components.py:112-119, 145-152) ->
    sparse 3x3x3 filter of the space-to-depth formulation; see m355_blur_weight_fwd."""

    @staticmethod
    def forward(ctx, w, scale, standardize, transposed):
        L = _lib.lib()
        _require(w, scale)
        w, scale = w.contiguous(), scale.contiguous()
        A, B = w.shape[0], w.shape[1]
        shape = (8 * B, A, 3, 3, 3) if transposed else (A, 8 * B, 3, 3, 3)
        wexp = torch.empty(shape, dtype=w.dtype, device=w.device)
        ms = torch.empty((A, 2), dtype=torch.float32, device=w.device) if standardize else None
        check(L.m355_blur_weight_fwd(_p(w), _p(scale), _p(wexp), _p(ms), A, B, int(standardize), int(transposed),
                                     _stream()), "blur_weight_fwd")
        ctx.flags = (A, B, int(standardize), int(transposed))
        ctx.save_for_backward(w, scale, ms)
        return wexp

    @staticmethod
    def backward(ctx, dwexp):
        L = _lib.lib()
        w, scale, ms = ctx.saved_tensors
        A, B, standardize, transposed = ctx.flags
        dw = torch.empty_like(w)
        check(L.m355_blur_weight_bwd(_p(dwexp.contiguous()), _p(w), _p(scale), _p(ms), _p(dw), A, B, standardize,
                                     transposed, _stream()), "blur_weight_bwd")
        return dw, None, None, None


def blur_weight(w, scale, standardize=False, transposed=False):
    """[A, B, 3, 3, 3] module weight -> conv3d weight of the space-to-depth form of the Blur convs:
    [A, 8B, 3, 3, 3] (strided conv) or [8B, A, 3, 3, 3] (transposed conv).  `scale`: [B] values of the
    module's `kernel` buffer."""
    if tuple(w.shape[2:]) != (3, 3, 3) or scale.numel() != w.shape[1]:
        raise _lib.M355Error(f"blur_weight: needs a 3x3x3 filter and one scale per dim-1 channel, got {tuple(w.shape)}")
    return _BlurWeightFn.apply(w, scale, bool(standardize), bool(transposed))


class _WeightStandardizeFn(torch.autograd.Function):
    """WSConv3d's (w - mean) / (std + 1e-5) per output filter (reference models/components.py:81-88)."""

    @staticmethod
    def forward(ctx, w):
        L = _lib.lib()
        _require(w)
        w = w.contiguous()
        A, n = w.shape[0], w[0].numel()
        wn = torch.empty_like(w)
        ms = torch.empty((A, 2), dtype=torch.float32, device=w.device)
        check(L.m355_weight_standardize_fwd(_p(w), _p(wn), _p(ms), A, n, _stream()), "weight_standardize_fwd")
        ctx.save_for_backward(w, ms)
        return wn

    @staticmethod
    def backward(ctx, dwn):
        L = _lib.lib()
        w, ms = ctx.saved_tensors
        dw = torch.empty_like(w)
        check(L.m355_weight_standardize_bwd(_p(dwn.contiguous()), _p(w), _p(ms), _p(dw), w.shape[0], w[0].numel(),
                                            _stream()), "weight_standardize_bwd")
        return dw


def weight_standardize(w):
    return _WeightStandardizeFn.apply(w)


# --------------------------------------------------- sliding window / evaluation
def patch_gather(volume, locations, patch_size):
    """volume [C,V0,V1,V2], locations int32 [P,3] (i0,j0,k0) -> patches [P,C,*patch_size]"""
    L = _lib.lib()
    _require(volume)
    _require(locations, dtype=torch.int32)
    volume, locations = volume.contiguous(), locations.contiguous()
    Cc, V0, V1, V2 = volume.shape
    P = locations.shape[0]
    ps0, ps1, ps2 = patch_size
    patches = torch.empty((P, Cc, ps0, ps1, ps2), dtype=volume.dtype, device=volume.device)
    check(L.m355_patch_gather(_p(volume), _p(locations), _p(patches), P, Cc, V0, V1, V2, ps0, ps1, ps2,
                              _stream()), "patch_gather")
    return patches


def sampler_build(prob_map, patch_size):
    """cumulative table of a [V0, V1, V2] probability map for `sampler_draw` (float64, (V / 1024 + 1) entries, the last
    one the total weight of the centres whose patch fits)"""
    L = _lib.lib()
    _require(prob_map)
    prob_map = prob_map.contiguous()
    V0, V1, V2 = prob_map.shape
    table = torch.empty(int(L.m355_sampler_table_bytes(V0, V1, V2)) // 8, dtype=torch.float64, device=prob_map.device)
    check(L.m355_sampler_build(_p(prob_map), V0, V1, V2, *[int(p) for p in patch_size], _p(table), _stream()), "sampler_build")
    return table


def sampler_draw(prob_map, table, patch_size, u):
    """u: float64 [P] uniform in [0, 1) on the device -> int32 [P, 3] patch corners (tio.WeightedSampler semantics)"""
    L = _lib.lib()
    _require(prob_map)
    _require(table, u, dtype=torch.float64)
    V0, V1, V2 = prob_map.shape
    P = u.numel()
    loc = torch.empty((P, 3), dtype=torch.int32, device=prob_map.device)
    check(L.m355_sampler_draw(_p(prob_map), _p(table), V0, V1, V2, *[int(p) for p in patch_size], _p(u.contiguous()), P,
                              _p(loc), _stream()), "sampler_draw")
    return loc


PAD_MODES = {"constant": 0, "edge": 1, "reflect": 2, "symmetric": 3, "wrap": 4}


def patch_gather_padded(volume, locations, patch_size, border, mode, value=0.0):
    """As patch_gather on the volume padded by `border` voxels per side (numpy.pad mode `mode`, or 'constant' with
    `value`) -- locations in PADDED coordinates; the padded volume is never materialised."""
    L = _lib.lib()
    _require(volume)
    _require(locations, dtype=torch.int32)
    volume, locations = volume.contiguous(), locations.contiguous()
    Cc, V0, V1, V2 = volume.shape
    P = locations.shape[0]
    ps0, ps1, ps2 = patch_size
    patches = torch.empty((P, Cc, ps0, ps1, ps2), dtype=volume.dtype, device=volume.device)
    check(L.m355_patch_gather_padded(_p(volume), _p(locations), _p(patches), P, Cc, V0, V1, V2, ps0, ps1, ps2,
                                     border[0], border[1], border[2], PAD_MODES[mode], float(value), _stream()),
          "patch_gather_padded")
    return patches


def patch_finalize_crop(accum, count, border):
    """accum [C, P0, P1, P2] / count with `border` voxels cropped off every side."""
    L = _lib.lib()
    _require(accum, count)
    Cc, P0, P1, P2 = accum.shape
    out = torch.empty((Cc, P0 - 2 * border[0], P1 - 2 * border[1], P2 - 2 * border[2]), dtype=accum.dtype,
                      device=accum.device)
    check(L.m355_patch_finalize_crop(_p(accum), _p(count), _p(out), Cc, P0, P1, P2, border[0], border[1], border[2],
                                     _stream()), "patch_finalize_crop")
    return out


def patch_accumulate(patches, locations, accum, count):
    """accum[C,V] += patches (in patch order), count[V] += 1 over each patch footprint"""
    L = _lib.lib()
    _require(patches, accum, count)
    _require(locations, dtype=torch.int32)
    patches, locations = patches.contiguous(), locations.contiguous()
    P, Cc, ps0, ps1, ps2 = patches.shape
    _, V0, V1, V2 = accum.shape
    check(L.m355_patch_accumulate(_p(patches), _p(locations), _p(accum), _p(count), P, Cc, V0, V1, V2,
                                  ps0, ps1, ps2, _stream()), "patch_accumulate")


def patch_aggregate_grid(tiles, axes, volume_shape, border=(0, 0, 0)):
    """tiles [n0*n1*n2, C, *patch] in GridSampler order over the per-axis start lists `axes` (padded coordinates) ->
    the averaged volume [C, *volume_shape] (the padded volume's interior when `border` > 0); one pass."""
    L = _lib.lib()
    _require(tiles)
    tiles = tiles.contiguous()
    P, Cc, ps0, ps1, ps2 = tiles.shape
    n0, n1, n2 = (len(a) for a in axes)
    if P != n0 * n1 * n2:
        raise _lib.M355Error(f"patch_aggregate_grid: {P} tiles for a {n0} x {n1} x {n2} grid")
    starts = torch.tensor([int(v) for a in axes for v in a], dtype=torch.int32, device=tiles.device)
    out = torch.empty((Cc,) + tuple(volume_shape), dtype=torch.float32, device=tiles.device)
    check(L.m355_patch_aggregate_grid(_p(tiles), _p(starts), n0, n1, n2, _p(out), Cc, *[int(v) for v in volume_shape], ps0, ps1,
                                      ps2, *[int(b) for b in border], _stream()), "patch_aggregate_grid")
    return out


def patch_finalize(accum, count):
    L = _lib.lib()
    _require(accum, count)
    Cc = accum.shape[0]
    V = count.numel()
    out = torch.empty_like(accum)
    check(L.m355_patch_finalize(_p(accum), _p(count), _p(out), Cc, V, _stream()), "patch_finalize")
    return out


def argmax_confusion(prob, target):
    """prob [N,C,...] float, target [N,...] int32 -> (argmax int32 [N,...], counts int64 [N,C,4] = TP,FP,FN,TN)"""
    L = _lib.lib()
    _require(prob)
    _require(target, dtype=torch.int32)
    prob, target = prob.contiguous(), target.contiguous()
    N, Cc = prob.shape[0], prob.shape[1]
    S = prob.numel() // (N * Cc)
    am = torch.empty(target.shape, dtype=torch.int32, device=prob.device)
    counts = torch.empty((N, Cc, 4), dtype=torch.int64, device=prob.device)
    check(L.m355_argmax_confusion(_p(prob), _p(target), _p(am), _p(counts), N, Cc, S, _stream()),
          "argmax_confusion")
    return am, counts
