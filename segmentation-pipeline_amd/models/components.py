"""Building blocks with the reference's constructor surface, executed by HIP kernels.

Mirror of segmentation_pipeline/models/components.py (reference): Block3d (:17-73),
WSConv3d (:76-88), BlurConv3d (:91-121), BlurConvTranspose3d (:124-154),
StochasticMatrix (:157-185).  Constructors take the same arguments (including the
stock ``nn.Conv3d`` / ``nn.GroupNorm`` / ``nn.ReLU`` classes as plug-ins) and create
the same sub-modules in the same order, so ``state_dict`` keys, parameter shapes and
seeded initialisation are interchangeable with reference checkpoints.  The stock
modules are used as parameter containers only: ``forward`` never calls them, it
dispatches to the hand-written kernels in ``ops`` and raises for plug-ins that have
no kernel (there is no torch fallback).
"""
from collections import OrderedDict
from numbers import Number
from typing import Optional
import weakref

import torch
from torch import nn
import torch.nn.functional as F

from .. import ops
from .._lib import ACT_LEAKY_RELU, ACT_NONE, ACT_RELU


def _volume(stride):
    v = 1
    for s in stride:
        v *= s
    return v


def _uniform_int(value, what):
    """nn modules store kernel/stride/padding as tuples; the kernels are isotropic."""
    if isinstance(value, int):
        return value
    vals = set(int(v) for v in value)
    if len(vals) != 1:
        raise NotImplementedError(f"anisotropic {what}={tuple(value)} has no HIP kernel")
    return vals.pop()


def _box_blur(weight, kernel):
    """F.conv3d(weight, kernel, padding=1, groups=in_channels) for the all-equal 2x2x2
    `kernel` buffer of the Blur convolutions (components.py:118,151): a k -> k+1 box
    sum of the zero-padded filter, scaled by the buffer value.  Written as 8 shifted
    adds on the (tiny) weight tensor so autograd differentiates it."""
    if kernel.shape[0] != weight.shape[1]:
        # the reference only works when the grouped conv is valid (in == out channels)
        raise RuntimeError(
            f"Blur convolution needs kernel rows ({kernel.shape[0]}) == weight.shape[1] ({weight.shape[1]})")
    k = weight.shape[2]
    wp = F.pad(weight, (1, 1, 1, 1, 1, 1))
    scale = kernel.reshape(kernel.shape[0], -1)[:, 0].reshape(1, -1, 1, 1, 1)
    acc = None
    for a in (0, 1):
        for b in (0, 1):
            for c in (0, 1):
                part = wp[:, :, a:a + k + 1, b:b + k + 1, c:c + k + 1]
                acc = part if acc is None else acc + part
    return acc * scale


def _blur_scale(kernel, weight):
    """per dim-1-channel value of the Blur modules' all-equal 2x2x2 `kernel` buffer"""
    if kernel.shape[0] != weight.shape[1]:
        # the reference only works when the grouped conv is valid (in == out channels)
        raise RuntimeError(
            f"Blur convolution needs kernel rows ({kernel.shape[0]}) == weight.shape[1] ({weight.shape[1]})")
    return kernel.reshape(kernel.shape[0], -1)[:, 0].contiguous()


# Derived filters (box blur + rearrangement, weight standardisation) depend only on the parameters: outside autograd
# they are computed once per parameter version and module, so an inference forward neither re-derives nor re-packs them
# (the same tensor object comes back, and ops caches its packed forms).  Keyed by module in a weak dictionary: nothing is
# added to the module itself (state_dict / pickled checkpoints stay as the reference's).
_DERIVED = weakref.WeakKeyDictionary()


def _derived_weight(mod, tag, make):
    w = mod.weight
    if torch.is_grad_enabled() and w.requires_grad:
        return make()
    k = getattr(mod, "kernel", None)
    key = (tag, w._version, w.data_ptr(), None if k is None else (k._version, k.data_ptr()),
           getattr(mod, "weight_standardization", None))
    hit = _DERIVED.get(mod)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = (key, make())
        _DERIVED[mod] = hit
    return hit[1]


class WSConv3d(nn.Conv3d):
    """Weight-standardised convolution (reference components.py:76-88)."""

    def __init__(self, in_channels, out_channels, kernel_size, **kwargs):
        super().__init__(in_channels, out_channels, kernel_size, **kwargs)
        self.kwargs = kwargs

    def effective(self):
        # the reference forwards only **kwargs to F.conv3d, so the bias is unused (:86)
        return _derived_weight(self, "ws", lambda: ops.weight_standardize(self.weight)), None

    def forward(self, x):
        return run_conv(self, x)


class BlurConv3d(nn.Conv3d):
    """Strided convolution with a box-blurred filter (reference components.py:91-121)."""

    def __init__(self, in_channels, out_channels, kernel_size, weight_standardization=False, **kwargs):
        super().__init__(in_channels, out_channels, kernel_size, **kwargs)
        self.weight_standardization = weight_standardization
        kernel = torch.ones(out_channels, 1, 2, 2, 2) / 8
        self.register_buffer('kernel', kernel)
        self.kernel = self.kernel / _volume(self.stride)  # volume shrinks by stride^3 (:108)
        self.kwargs = kwargs

    def effective(self):
        w = self.weight
        if self.weight_standardization:
            w = ops.weight_standardize(w)
        return _box_blur(w, self.kernel), None  # bias never used (:119)

    def forward(self, x):
        stride, pad = _uniform_int(self.stride, "stride"), _uniform_int(self.padding, "padding")
        # (a c8 activation of the 16-bit flows is converted by space_to_depth2 -- it must not fall to the generic
        # strided kernel below, which is a direct fp32 kernel)
        even = isinstance(x, (torch.Tensor, ops.Act16)) and all(s % 2 == 0 for s in x.shape[2:])
        if _uniform_int(self.kernel_size, "kernel_size") == 3 and stride == 2 and pad == 1 and even:
            # effective 4x4x4 / stride 2 / padding 1 = a stride-1 3x3x3 conv over the space-to-depth input
            # (MFMA path); standardisation + box blur + rearrangement of the filter in one HIP kernel
            wexp = _derived_weight(self, "s2d", lambda: ops.blur_weight(
                self.weight, _blur_scale(self.kernel, self.weight), self.weight_standardization))
            # (16-bit flows: a c8 activation is rearranged in c8 and the conv returns c8 for the block that follows)
            return ops.conv3d(ops.space_to_depth2(x), wexp, None, stride=1, padding=1, c8_out=isinstance(x, ops.Act16))
        return run_conv(self, x)


class BlurConvTranspose3d(nn.ConvTranspose3d):
    """Transposed convolution with a box-blurred filter (reference components.py:124-154)."""

    def __init__(self, in_channels, out_channels, kernel_size, weight_standardization=False, **kwargs):
        super().__init__(in_channels, out_channels, kernel_size, **kwargs)
        self.weight_standardization = weight_standardization
        kernel = torch.ones(out_channels, 1, 2, 2, 2)
        kernel = kernel / torch.sum(kernel)  # normalised over ALL C*8 entries (:137)
        self.register_buffer('kernel', kernel)
        self.kernel = self.kernel * _volume(self.stride)  # volume grows by stride^3 (:141)
        self.kwargs = kwargs

    def forward(self, x, output_size=None, out=None):
        if (_uniform_int(self.kernel_size, "kernel_size") == 3 and _uniform_int(self.stride, "stride") == 2
                and _uniform_int(self.padding, "padding") == 1
                and _uniform_int(self.output_padding, "output_padding") == 0):
            # effective 4x4x4 / stride 2 / padding 1: a 3x3x3 conv producing the 8 output parities, then
            # depth-to-space (MFMA path); filter transform in one HIP kernel
            wexp = _derived_weight(self, "d2s", lambda: ops.blur_weight(
                self.weight, _blur_scale(self.kernel, self.weight), self.weight_standardization, transposed=True))
            c8 = isinstance(x, ops.Act16) and (out is None or out.buf16 is not None)   # 16-bit flows: parities in c8, then c8 -> c8
            return ops.depth_to_space2(ops.conv3d(x, wexp, None, stride=1, padding=1, c8_out=c8), out=out)
        w = self.weight
        if self.weight_standardization:
            w = ops.weight_standardize(w)
        w = _box_blur(w, self.kernel)
        return ops.conv_transpose3d(
            x, w, None, stride=_uniform_int(self.stride, "stride"),
            padding=_uniform_int(self.padding, "padding"),
            output_padding=_uniform_int(self.output_padding, "output_padding"), out=out)


def _check_conv_module(m):
    if not isinstance(m, nn.Conv3d):
        raise NotImplementedError(
            f"conv_class {type(m).__name__} has no HIP kernel (supported: nn.Conv3d, WSConv3d, BlurConv3d)")
    if m.groups != 1 or _uniform_int(m.dilation, "dilation") != 1 or m.padding_mode != 'zeros':
        raise NotImplementedError("grouped / dilated / non-zero-padded Conv3d has no HIP kernel")
    if isinstance(m.padding, str):
        raise NotImplementedError("string padding modes have no HIP kernel")
    _uniform_int(m.kernel_size, "kernel_size")
    _uniform_int(m.stride, "stride")
    _uniform_int(m.padding, "padding")


def run_conv(m, x, add=None, out=None, stats=None, c8_out=False, softmax=False):
    """Execute an nn.Conv3d-like parameter container with the HIP conv kernels."""
    if hasattr(m, "effective"):
        weight, bias = m.effective()
    else:
        weight, bias = m.weight, m.bias
    return ops.conv3d(x, weight, bias, add=add, stride=_uniform_int(m.stride, "stride"),
                      padding=_uniform_int(m.padding, "padding"), out=out, stats=stats, c8_out=c8_out, softmax=softmax)


def _act_code(m):
    if m is None or isinstance(m, nn.Identity):
        return ACT_NONE, 0.0
    if isinstance(m, nn.ReLU):
        return ACT_RELU, 0.0
    if isinstance(m, nn.LeakyReLU):
        return ACT_LEAKY_RELU, float(m.negative_slope)
    raise NotImplementedError(
        f"activation_class {type(m).__name__} has no HIP kernel (supported: ReLU, LeakyReLU, Identity)")


def _check_norm_module(m):
    if m is None or isinstance(m, (nn.GroupNorm, nn.BatchNorm3d, nn.InstanceNorm3d)):
        return
    raise NotImplementedError(
        f"normalization_class {type(m).__name__} has no HIP kernel "
        "(supported: BatchNorm3d, GroupNorm, InstanceNorm3d)")


def wants_batch_stats(norm):
    """True when the normalisation computes statistics from its input (so the producing conv should
    emit the partial sums): GroupNorm / InstanceNorm always, BatchNorm3d in training mode."""
    if isinstance(norm, (nn.GroupNorm, nn.InstanceNorm3d)):
        return True
    return isinstance(norm, nn.BatchNorm3d) and (norm.training or norm.running_mean is None)


def run_norm_act(norm, act, x, add=None, out=None, stats=None, c8=0, pool=False):
    """normalization + activation (+ residual add) through the fused HIP passes.  `stats`: partial
    sums emitted by the conv that produced `x` (saves the statistics pass over x).  `c8`: 16-bit compute
    code -> the result is written only in the c8 layout the next convolution reads (ops.Act16).  `pool`: also
    return nn.AvgPool3d(2, 2) of the result, computed in the same pass -> (y, pooled)."""
    code, slope = _act_code(act)
    fn = ops.norm_act
    if pool and c8:      # c8 training flow: the pooled tensor is a second output of the c8 -> c8 pass
        assert add is None
        fn = lambda x_, g_, b_, cfg_, add=None: ops.norm_act(x_, g_, b_, cfg_, pool=True)
    elif pool:
        assert add is None
        fn = lambda x_, g_, b_, cfg_, add=None: ops.norm_act_pool(x_, g_, b_, cfg_)
    if norm is None:
        # activation only: identity statistics
        Cc = x.shape[1]
        cfg = ops.NormCfg(groups=0, eps=0.0, act=code, slope=slope, training=False,
                          running_mean=torch.zeros(Cc, device=x.device),
                          running_var=torch.ones(Cc, device=x.device), out=out, c8=c8)
        return fn(x, None, None, cfg, add=add)
    if isinstance(norm, nn.GroupNorm):
        cfg = ops.NormCfg(groups=norm.num_groups, eps=norm.eps, act=code, slope=slope, out=out, stats=stats, c8=c8)
        return fn(x, norm.weight, norm.bias, cfg, add=add)
    if isinstance(norm, nn.InstanceNorm3d):
        if norm.track_running_stats:
            raise NotImplementedError("InstanceNorm3d(track_running_stats=True) has no HIP kernel")
        cfg = ops.NormCfg(groups=x.shape[1], eps=norm.eps, act=code, slope=slope, out=out, stats=stats, c8=c8)
        return fn(x, norm.weight, norm.bias, cfg, add=add)
    # BatchNorm3d: batch statistics in training mode (and whenever no running stats exist)
    training = norm.training or norm.running_mean is None
    momentum = norm.momentum
    if training and norm.running_mean is not None:
        norm.num_batches_tracked.add_(1)
        if momentum is None:  # cumulative moving average
            momentum = 1.0 / float(norm.num_batches_tracked)
    cfg = ops.NormCfg(groups=0, eps=norm.eps, act=code, slope=slope, training=training,
                      momentum=0.0 if momentum is None else momentum,
                      running_mean=norm.running_mean, running_var=norm.running_var, out=out,
                      stats=stats if training else None, c8=c8)
    return fn(x, norm.weight, norm.bias, cfg, add=add)


class Block3d(nn.Module):
    """[conv -> norm -> activation] x num_convs (+ residual conv, + Dropout3d).

    Same signature, defaults and sub-module names as the reference Block3d
    (components.py:17-60).  forward (:62-73) runs: conv kernels (fp32 MFMA implicit
    GEMM), a fused statistics + normalise + activation pass per stage, the residual
    branch fused into the last pass, and an optional direct write into a concat slot.
    """

    def __init__(
            self,
            in_channels,
            out_channels,
            conv_class=nn.Conv3d,
            conv_params=None,
            normalization_class=nn.BatchNorm3d,
            normalization_params=None,
            activation_class=nn.ReLU,
            activation_params=None,
            residual=False,
            residual_params=None,
            dropout_p=0.0,
            num_convs=2,
    ):
        super().__init__()
        conv_params = {'bias': False, 'kernel_size': 3, 'padding': 1} if conv_params is None else conv_params
        normalization_params = {} if normalization_params is None else normalization_params
        activation_params = {'inplace': True} if activation_params is None else activation_params
        residual_params = ({'bias': True, 'kernel_size': 3, 'padding': 1}
                           if residual_params is None else residual_params)

        self.residual = residual
        if self.residual:
            self.res_conv = conv_class(in_channels, out_channels, **residual_params)
            _check_conv_module(self.res_conv)

        stages = OrderedDict()
        for i in range(num_convs):
            conv = conv_class(in_channels if i == 0 else out_channels, out_channels, **conv_params)
            _check_conv_module(conv)
            stages[f'conv{i}'] = conv
            if normalization_class is not None:
                norm = normalization_class(out_channels, **normalization_params)
                _check_norm_module(norm)
                stages[f'norm{i}'] = norm
            if activation_class is not None:
                act = activation_class(**activation_params)
                _act_code(act)
                stages[f'activation{i}'] = act
        self.layers = nn.Sequential(stages)
        self._num_convs = num_convs

        self.dropout = None
        if dropout_p != 0.0:
            self.dropout = nn.Dropout3d(p=dropout_p)

    def forward(self, x, out: Optional[ops.OutSlot] = None, c8_out: bool = False, pool: bool = False):
        """`c8_out` (or an `out` slot of a c8 concat buffer): in a 16-bit precision mode under no_grad the
        block's result is returned as an `ops.Act16`; the activations BETWEEN its convolutions always are.
        `pool`: the caller wants nn.AvgPool3d(2, 2) of the result as well; where the block can produce it from its
        last normalise + activation pass (fp32 tensors, no residual branch, no dropout, even sizes) it returns
        (result, pooled) -- otherwise just the result, and the caller pools."""
        drop = self.dropout is not None and self.training and self.dropout.p > 0.0
        final_out = None if drop else out
        flow = ops.h16_flow()
        c8_out = bool(flow) and (c8_out or (out is not None and out.buf16 is not None))
        last_norm = (getattr(self.layers, f'norm{self._num_convs - 1}', None) is not None or
                     getattr(self.layers, f'activation{self._num_convs - 1}', None) is not None) if self._num_convs else False
        # the residual branch is added inside the last norm/act pass: in the c8 flow it is a c8 tensor too
        res = run_conv(self.res_conv, x, c8_out=c8_out and last_norm) if self.residual else None

        h = x
        for i in range(self._num_convs):
            last = i == self._num_convs - 1
            conv = getattr(self.layers, f'conv{i}')
            norm = getattr(self.layers, f'norm{i}', None)
            act = getattr(self.layers, f'activation{i}', None)
            add = res if last else None
            slot = final_out if last else None
            if norm is None and act is None:
                h = run_conv(conv, h, add=add, out=slot)
            else:
                # the conv epilogue emits the partial sums of the normalisation that follows
                stats = {} if wants_batch_stats(norm) else None
                c8 = flow if (not last or c8_out) else 0
                h = run_conv(conv, h, stats=stats, c8_out=bool(c8))   # c8 flow: the pre-norm tensor is c8 as well
                if (pool and ops.FUSE_POOL and last and add is None and not drop and not c8 and ops.is_fp32()
                        and isinstance(h, torch.Tensor) and all(v % 2 == 0 for v in h.shape[2:])):
                    return run_norm_act(norm, act, h, out=slot, stats=stats, pool=True)   # -> (result, pooled)
                if (pool and last and add is None and not drop and c8 and torch.is_grad_enabled() and norm is not None
                        and isinstance(h, ops.Act16) and all(v % 2 == 0 for v in h.shape[2:])):
                    # c8 training flow: (result, pooled) from one autograd node, whose backward sums the skip-path and
                    # the un-pooled gradient inside the normalisation backward
                    return run_norm_act(norm, act, h, out=slot, stats=stats, c8=c8, pool=True)
                h = run_norm_act(norm, act, h, add=add, out=slot, stats=stats, c8=c8)
        if self._num_convs == 0 and res is not None:
            h = ops.add(res, h)

        if drop:
            # Dropout3d: whole channels zeroed with probability p, survivors scaled by 1/(1-p)
            p = self.dropout.p
            noise = torch.empty(h.shape[0] * h.shape[1], device=h.device).bernoulli_(1.0 - p).div_(1.0 - p)
            h = ops.channel_scale(h, noise, out=out)     # (a c8 activation stays c8, scaled into its slot)
        return h


class StochasticMatrix(nn.Module):
    """Reshape (N, C*C, ...) -> (N, C, C, ...), add diag_bias * I, softmax over dim 1
    (reference components.py:157-185), as one strided softmax kernel."""

    def __init__(self, channels: int, diag_bias: Optional[Number] = None):
        super().__init__()
        self.channels = channels
        self.diag_bias = diag_bias

    def forward(self, x):
        C = self.channels
        if x.shape[1] != C * C:
            raise RuntimeError("Expected dim 1 of input tensor to be the square of the number of out channels")
        bias = 0.0 if self.diag_bias is None else float(self.diag_bias)
        return ops.softmax_channels(x, inner=C, diag_bias=bias)
