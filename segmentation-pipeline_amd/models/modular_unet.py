"""ModularUNet with the reference's plug-in surface, executed by HIP kernels.

Mirror of segmentation_pipeline/models/modular_unet.py:11-102 (reference): same
constructor signature and defaults (:12-46), same sub-module names and creation
order (:50-84) -> identical state_dict keys and seeded initialisation.  forward
(:86-102) keeps the reference's dataflow but never materialises torch.cat: the
encoder block and the upsampler of a level write straight into the two channel
slices of one pre-allocated buffer, which the decoder block's first convolution
reads whole ([x_up | x_skip], upsampled part first as in :97).
"""
from typing import Dict, Optional, Sequence, Union

import torch
from torch import nn

from .. import ops
from .components import Block3d, BlurConv3d, BlurConvTranspose3d, StochasticMatrix, run_conv, _uniform_int
from .utils import filter_kwargs, is_sequence


def _run_downsample(m, x):
    if isinstance(m, nn.AvgPool3d):
        k, s = _uniform_int(m.kernel_size, "kernel_size"), _uniform_int(m.stride, "stride")
        if k != 2 or s != 2 or _uniform_int(m.padding, "padding") != 0 or m.ceil_mode:
            raise NotImplementedError("only AvgPool3d(kernel_size=2, stride=2) has a HIP kernel")
        return None if x is None else ops.avgpool3d_2x(x)
    if isinstance(m, BlurConv3d):
        return m(x)      # (a c8 activation of the 16-bit flows stays c8: space-to-depth and the conv run on c8)
    if isinstance(m, nn.Conv3d):  # WSConv3d / strided nn.Conv3d
        return run_conv(m, x)
    raise NotImplementedError(f"downsample_class {type(m).__name__} has no HIP kernel")


def _is_avgpool2(m):
    """nn.AvgPool3d(kernel_size=2, stride=2) -- the pool an encoder block can emit from its last pass"""
    if not isinstance(m, nn.AvgPool3d) or m.ceil_mode:
        return False
    try:
        return (_uniform_int(m.kernel_size, "kernel_size") == 2 and _uniform_int(m.stride, "stride") == 2
                and _uniform_int(m.padding, "padding") == 0)
    except (NotImplementedError, ValueError, TypeError):
        return False


def _run_upsample(m, x, out=None):
    if isinstance(m, nn.Upsample):
        sf = m.scale_factor
        sf = sf if not isinstance(sf, (tuple, list)) else (sf[0] if len(set(sf)) == 1 else None)
        if m.mode != 'trilinear' or not m.align_corners or sf is None or float(sf) != 2.0:
            raise NotImplementedError(
                "only Upsample(scale_factor=2, mode='trilinear', align_corners=True) has a HIP kernel")
        return ops.upsample_trilinear2x(x, out=out)
    if isinstance(m, BlurConvTranspose3d):
        return m(x, out=out)
    if isinstance(m, nn.ConvTranspose3d):
        if m.groups != 1 or _uniform_int(m.dilation, "dilation") != 1:
            raise NotImplementedError("grouped / dilated ConvTranspose3d has no HIP kernel")
        return ops.conv_transpose3d(
            x, m.weight, m.bias, stride=_uniform_int(m.stride, "stride"),
            padding=_uniform_int(m.padding, "padding"),
            output_padding=_uniform_int(m.output_padding, "output_padding"), out=out)
    raise NotImplementedError(f"upsample_class {type(m).__name__} has no HIP kernel")


def _upsample_out_channels(m, cin):
    return m.out_channels if isinstance(m, nn.ConvTranspose3d) else cin


def _run_hypothesis(m, x):
    if isinstance(m, nn.Softmax):
        if m.dim != 1:
            raise NotImplementedError("only Softmax(dim=1) has a HIP kernel")
        return ops.softmax_channels(x)
    if isinstance(m, StochasticMatrix):
        return m(x)
    if isinstance(m, nn.Identity):
        return x
    raise NotImplementedError(f"hypothesis_class {type(m).__name__} has no HIP kernel")


class ModularUNet(nn.Module):
    def __init__(
            self,
            in_channels: int,
            out_channels: int,
            filters: Union[int, Sequence[int]],
            depth: int,
            block_class: nn.Module = Block3d,
            block_params: Optional[Dict] = None,
            upsample_class: nn.Module = nn.Upsample,
            upsample_params: Optional[Dict] = None,
            downsample_class: nn.Module = nn.AvgPool3d,
            downsample_params: Optional[Dict] = None,
            out_conv_class: nn.Module = nn.Conv3d,
            out_conv_params: Optional[Dict] = None,
            hypothesis_class: nn.Module = nn.Softmax,
            hypothesis_params: Optional[Dict] = None,
    ):
        super().__init__()

        if isinstance(filters, int):
            filters = [filters] * depth
        elif is_sequence(filters) and len(filters) != depth:
            raise ValueError(f"Sequence of filters {filters} does not match depth {depth}")

        block_params = {} if block_params is None else block_params
        if upsample_params is None:
            upsample_params = {'scale_factor': 2, 'mode': 'trilinear', 'align_corners': True}
        if downsample_params is None:
            downsample_params = {'kernel_size': 2, 'stride': 2, 'count_include_pad': False}
        if out_conv_params is None:
            out_conv_params = {'in_channels': filters[0], 'out_channels': out_channels, 'kernel_size': 3,
                               'padding': 1}
        if hypothesis_params is None:
            hypothesis_params = {"dim": 1}

        self.depth = depth
        self._filters = list(filters)

        # creation order matters for seeded initialisation: down blocks, downsamplers,
        # up blocks, upsamplers, out conv (reference :50-84)
        self.down_blocks = nn.ModuleList(
            [block_class(in_channels if i == 0 else filters[i - 1], filters[i], **block_params)
             for i in range(depth)])

        self.downsampling = nn.ModuleList()
        for i in range(depth - 1):
            downsample_params.update(filter_kwargs(
                downsample_class, in_channels=filters[i], out_channels=filters[i], channels=filters[i]))
            self.downsampling.append(downsample_class(**downsample_params))

        self.up_blocks = nn.ModuleList(
            [block_class(filters[i] + filters[i + 1], filters[i], **block_params) for i in range(depth - 1)])

        self.upsampling = nn.ModuleList()
        for i in range(1, depth):
            upsample_params.update(filter_kwargs(
                upsample_class, in_channels=filters[i], out_channels=filters[i], channels=filters[i]))
            self.upsampling.append(upsample_class(**upsample_params))

        self.out_conv = out_conv_class(**out_conv_params)
        self.hypothesis = hypothesis_class(**hypothesis_params)

        for blk in list(self.down_blocks) + list(self.up_blocks):
            if not isinstance(blk, Block3d):
                raise NotImplementedError(
                    f"block_class {type(blk).__name__} has no HIP execution path (use models.Block3d)")

    def forward(self, x):
        # one forward pass = one loss-scale cell of the fp16 training flow (ops.GradScale; a no-op in every other mode)
        with ops.grad_scale_scope():
            return self._forward(x)

    def _forward(self, x):
        f = self._filters
        N, spatial = x.shape[0], tuple(x.shape[2:])
        # 16-bit precision mode under no_grad: activations live only in the c8 layout the conv kernels read
        # (ops.Act16); concat buffers are c8 buffers whose slots start at multiples of 8 channels
        flow = ops.h16_flow()
        ups = [_upsample_out_channels(self.upsampling[i], f[i + 1]) for i in range(self.depth - 1)]
        if flow and any(c % 8 for c in list(f) + ups):
            flow = 0
        if flow:
            x = ops.pack_act16(x, flow)   # the (few-channel) network input joins the c8 flow: its conv writes c8 too
        skips = []
        for i in range(self.depth):
            if i != self.depth - 1:
                # level-i concat buffer: [upsampled (f[i+1]) | skip (f[i])]
                c_up = ups[i]
                if flow:
                    buf = ops.Act16.empty(N, c_up + f[i], spatial, flow, x.device)
                    slot = ops.OutSlot(None, c_up, c_up + f[i], buf16=buf)
                else:
                    buf = torch.empty((N, c_up + f[i]) + spatial, dtype=torch.float32, device=x.device)
                    slot = ops.OutSlot(buf, c_up, c_up + f[i])
                # the block's last pass also emits the pooled tensor: fp32 flow, and the c8 TRAINING flow (one autograd
                # node for both uses of the block output); the c8 no-grad flow pools in its own small pass
                fuse_pool = ((not flow or torch.is_grad_enabled()) and _is_avgpool2(self.downsampling[i])
                             and isinstance(self.down_blocks[i], Block3d))
                x = self.down_blocks[i](x, out=slot, pool=True) if fuse_pool else self.down_blocks[i](x, out=slot)
                if isinstance(x, tuple):
                    x_skip, x = x      # AvgPool3d(2, 2) came out of the block's last norm + activation pass
                elif (isinstance(self.downsampling[i], nn.AvgPool3d) and isinstance(x, (torch.Tensor, ops.Act16))
                        and x.requires_grad and torch.is_grad_enabled()):
                    _run_downsample(self.downsampling[i], None)          # validates the module's geometry
                    x_skip, x = ops.avgpool3d_2x_with_skip(x)            # one fused gradient for both uses
                else:
                    x_skip = x
                    x = _run_downsample(self.downsampling[i], x)
                skips.append((x_skip, buf, c_up))
                spatial = tuple(x.shape[2:])
            else:
                x = self.down_blocks[i](x, c8_out=bool(flow))

        for i in reversed(range(self.depth - 1)):
            x_skip, buf, c_up = skips[i]
            slot = ops.OutSlot(None, 0, c_up, buf16=buf) if flow else ops.OutSlot(buf, 0, c_up)
            x_up = _run_upsample(self.upsampling[i], x, out=slot)
            x = self.up_blocks[i](ops.Concat(buf, [x_up, x_skip]), c8_out=bool(flow))

        if isinstance(self.hypothesis, nn.Softmax) and self.hypothesis.dim == 1 and isinstance(self.out_conv, nn.Conv3d):
            # out conv + Softmax(dim=1) (:99-100) as one op: the softmax runs in the conv epilogue
            return run_conv(self.out_conv, x, softmax=True)
        x = run_conv(self.out_conv, x)
        return _run_hypothesis(self.hypothesis, ops.as_f32(x))
