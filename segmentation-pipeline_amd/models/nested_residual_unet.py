"""NestedResUNet (UNet++-style, 4 levels, 10 blocks) executed by HIP kernels.

Mirror of segmentation_pipeline/models/nested_residual_unet.py:6-106 (reference):
same constructor, same attribute names (conv{r}_{c}.{res_conv,conv1,bn1,conv2,bn2},
out_conv) and creation order.  The 2- and 3-way torch.cat calls of the reference
forward (:92-101) become channel slots of pre-allocated buffers that the producing
block / pool / upsample kernels write directly.
"""
from typing import Dict, Optional

import torch
from torch import nn

from .. import ops
from .components import run_conv, run_norm_act, wants_batch_stats
from .modular_unet import _run_hypothesis


class NestedResUNet(nn.Module):
    class Block(nn.Module):
        """conv-bn-relu x2 with an optional biased residual conv (reference :7-47)."""

        def __init__(self, in_ch, out_ch, residual=False, dropout_p=0.0):
            super().__init__()
            self.residual = residual
            self.out_ch = out_ch
            conv_params = dict(kernel_size=3, padding=1)
            if self.residual:
                self.res_conv = nn.Conv3d(in_ch, out_ch, **conv_params)
            self.conv1 = nn.Conv3d(in_ch, out_ch, bias=False, **conv_params)
            self.bn1 = nn.BatchNorm3d(out_ch)
            self.activation1 = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv3d(out_ch, out_ch, bias=False, **conv_params)
            self.bn2 = nn.BatchNorm3d(out_ch)
            self.activation2 = nn.ReLU(inplace=True)
            self.dropout = None
            if dropout_p != 0.0:
                self.dropout = nn.Dropout3d(p=dropout_p)

        def forward(self, x, out: Optional[ops.OutSlot] = None, c8: int = 0):
            """`c8`: 16-bit compute code -> `x`, every intermediate and the result are c8 activations (ops.Act16)"""
            res = run_conv(self.res_conv, x, c8_out=bool(c8)) if self.residual else None
            drop = self.dropout is not None and self.training and self.dropout.p > 0.0
            s1 = {} if wants_batch_stats(self.bn1) else None
            h = run_norm_act(self.bn1, self.activation1, run_conv(self.conv1, x, stats=s1, c8_out=bool(c8)), stats=s1, c8=c8)
            s2 = {} if wants_batch_stats(self.bn2) else None
            h = run_norm_act(self.bn2, self.activation2, run_conv(self.conv2, h, stats=s2, c8_out=bool(c8)), stats=s2,
                             add=res, out=None if drop else out, c8=c8)
            if drop:
                # Dropout3d: whole channels zeroed with probability p, survivors scaled by 1/(1-p); written into the slot
                p = self.dropout.p
                noise = torch.empty(h.shape[0] * h.shape[1], device=h.device).bernoulli_(1.0 - p).div_(1.0 - p)
                h = ops.channel_scale(h, noise, out=out)
            return h

    def __init__(
            self,
            input_channels: int,
            output_channels: int,
            filters: int,
            dropout_p: float = 0.0,
            hypothesis_class: nn.Module = nn.Softmax,
            hypothesis_params: Optional[Dict] = None,
    ):
        super().__init__()
        if hypothesis_params is None:
            hypothesis_params = {"dim": 1}

        self.dropout = None
        if dropout_p != 0.0:
            self.dropout = nn.Dropout3d(p=dropout_p)

        self.down = nn.AvgPool3d(kernel_size=2, stride=2, count_include_pad=False)
        self.up = nn.Upsample(scale_factor=2, mode='trilinear', align_corners=True)
        self._filters = filters

        bp = dict(dropout_p=dropout_p)
        # creation order of the reference (:72-85), which fixes the seeded initialisation
        self.conv0_0 = self.Block(input_channels, filters, **bp, residual=True)
        self.conv1_0 = self.Block(filters, filters, **bp)
        self.conv0_1 = self.Block(filters * 2, filters, **bp, residual=True)

        self.conv2_0 = self.Block(filters, filters, **bp)
        self.conv1_1 = self.Block(filters * 3, filters, **bp)
        self.conv0_2 = self.Block(filters * 2, filters, **bp, residual=True)

        self.conv3_0 = self.Block(filters, filters, **bp)
        self.conv2_1 = self.Block(filters * 3, filters, **bp)
        self.conv1_2 = self.Block(filters * 3, filters, **bp)
        self.conv0_3 = self.Block(filters * 2, filters, **bp, residual=True)

        self.out_conv = nn.Conv3d(filters, output_channels, kernel_size=3, padding=1)
        self.hypothesis = hypothesis_class(**hypothesis_params)

    def forward(self, x):
        with ops.grad_scale_scope():   # (one loss-scale cell per forward pass of the fp16 training flow)
            return self._forward(x)

    def _forward(self, x):
        F = self._filters
        N = x.shape[0]
        sp = [tuple(s >> lvl for s in x.shape[2:]) for lvl in range(4)]

        # 16-bit precision modes: activations (and, while training, their gradients) live only in the c8 layout the conv
        # kernels read (ops.Act16); the concat buffers are c8 buffers whose slots start at multiples of 8 channels
        flow = ops.h16_flow() if F % 8 == 0 else 0
        if flow:
            x = ops.pack_act16(x, flow)

        def buf(parts, lvl):
            if flow:
                return ops.Act16.empty(N, F * parts, sp[lvl], flow, x.device)
            return torch.empty((N, F * parts) + sp[lvl], dtype=x.dtype, device=x.device)

        def slot(b, i):
            if flow:
                return ops.OutSlot(None, F * i, F * (i + 1), buf16=b)
            return ops.OutSlot(b, F * i, F * (i + 1))

        up = ops.upsample_trilinear2x

        def down(t, out=None):
            """-> (t as it continues into its other consumers, AvgPool3d(2, 2)(t)).  c8 training flow: one autograd node
            for the pair, whose backward adds the un-pooled gradient to the gradient of the other consumers in the pool
            backward pass (instead of a separate sum of two full-resolution c8 gradients)"""
            if flow and torch.is_grad_enabled() and isinstance(t, ops.Act16) and t.requires_grad:
                return ops.avgpool3d_2x_with_skip(t, out=out)
            return t, ops.avgpool3d_2x(t, out=out)

        # each buffer is the input of one nested block: [own-row predecessor | up | down]
        b01, b02, b03 = buf(2, 0), buf(2, 0), buf(2, 0)
        b11, b12 = buf(3, 1), buf(3, 1)
        b21 = buf(3, 2)

        x0_0 = self.conv0_0(x, out=slot(b01, 0), c8=flow)
        x0_0, p0_0 = down(x0_0)
        x1_0 = self.conv1_0(p0_0, out=slot(b11, 0), c8=flow)
        x1_0, p1_0 = down(x1_0)
        x0_1 = self.conv0_1(ops.Concat(b01, [x0_0, up(x1_0, out=slot(b01, 1))]), out=slot(b02, 0), c8=flow)
        x0_1, p0_1 = down(x0_1, out=slot(b11, 2))

        x2_0 = self.conv2_0(p1_0, out=slot(b21, 0), c8=flow)
        x2_0, p2_0 = down(x2_0)
        x1_1 = self.conv1_1(ops.Concat(b11, [x1_0, up(x2_0, out=slot(b11, 1)), p0_1]), out=slot(b12, 0), c8=flow)
        x1_1, p1_1 = down(x1_1, out=slot(b21, 2))
        x0_2 = self.conv0_2(ops.Concat(b02, [x0_1, up(x1_1, out=slot(b02, 1))]), out=slot(b03, 0), c8=flow)
        x0_2, p0_2 = down(x0_2, out=slot(b12, 2))

        x3_0 = self.conv3_0(p2_0, c8=flow)
        x2_1 = self.conv2_1(ops.Concat(b21, [x2_0, up(x3_0, out=slot(b21, 1)), p1_1]), c8=flow)
        x1_2 = self.conv1_2(ops.Concat(b12, [x1_1, up(x2_1, out=slot(b12, 1)), p0_2]), c8=flow)
        x0_3 = self.conv0_3(ops.Concat(b03, [x0_2, up(x1_2, out=slot(b03, 1))]), c8=flow)

        if isinstance(self.hypothesis, nn.Softmax) and self.hypothesis.dim == 1:
            # out conv + Softmax(dim=1) (:103-104) as one op: the softmax runs in the conv epilogue where the variant has one
            return run_conv(self.out_conv, x0_3, softmax=True)
        return _run_hypothesis(self.hypothesis, ops.as_f32(run_conv(self.out_conv, x0_3)))
