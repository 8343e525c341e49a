"""Thin functional wrapper over the C ABI used by the tests.

`RawOps("hip")` drives libm355seg.so with CUDA tensors; `RawOps("oracle")` drives
oracle/libm355_oracle.so (same signatures, m355o_ prefix) with CPU tensors.  The
parity tests run the same call on both and compare.
"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from segmentation_pipeline_amd import _lib  # noqa: E402
from segmentation_pipeline_amd._lib import ConvDesc, NormDesc  # noqa: E402

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.environ.get("M355_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libm355_oracle.so")  # override: the ASan build (oracle/Makefile)


def build_oracle():
    src = os.path.join(ORACLE_DIR, "m355_oracle.c")
    if (not os.path.exists(ORACLE_LIB)) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return ORACLE_LIB


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class RawOps:
    def __init__(self, backend):
        self.backend = backend
        if backend == "hip":
            self.lib, self.prefix, self.device = _lib.lib(), "m355_", "cuda"
        elif backend == "oracle":
            self.lib = _lib.bind(C.CDLL(build_oracle()), prefix="m355o_")
            self.prefix, self.device = "m355o_", "cpu"
        else:
            raise ValueError(backend)

    # ------------------------------------------------------------- plumbing
    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream) if self.device == "cuda" else None

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.m355_last_error().decode() if self.backend == "hip" else ""
            raise RuntimeError(f"{what} -> {rc} {msg}")

    def to(self, t):
        return None if t is None else t.to(self.device).contiguous()

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _ws(self, qname, desc):
        n = 0
        if self.backend == "hip":
            n = getattr(self.lib, "m355_" + qname)(C.byref(desc))
        return torch.empty(max(int(n), 16), dtype=torch.uint8, device=self.device)

    # ------------------------------------------------------------------ conv
    @staticmethod
    def conv_desc(x_shape, Cout, k, stride, pad, out_pad=0, xbs=0, ybs=0, compute=0):
        N, Cin, D, H, W = x_shape
        return ConvDesc(N, Cin, Cout, D, H, W, k, stride, pad, out_pad, xbs, ybs, compute, 0)

    def pack_weights(self, w, x_shape, which, compute=0):
        """m355_conv3d_pack: (packed buffer, descriptor flags) -- HIP library only"""
        w = self.to(w)
        d = self.conv_desc(x_shape, w.shape[0], w.shape[2], 1, 1, compute=compute)
        n = self.lib.m355_conv3d_packed_bytes(C.byref(d), which)
        assert n > 0
        buf = torch.full((int(n),), 0xA5, dtype=torch.uint8, device=self.device)   # (bytes a layout leaves unwritten compare equal)
        self._chk(self.lib.m355_conv3d_pack(C.byref(d), which, _p(w), _p(buf), self._stream()), "conv3d_pack")
        return buf

    def pack_weights_batch(self, cases):
        """m355_conv3d_pack_batch over cases [(w, x_shape, which, compute)] -> list of packed buffers"""
        items, bufs, keep = [], [], []
        for w, x_shape, which, compute in cases:
            w = self.to(w)
            d = self.conv_desc(x_shape, w.shape[0], w.shape[2], 1, 1, compute=compute)
            n = self.lib.m355_conv3d_packed_bytes(C.byref(d), which)
            assert n > 0
            buf = torch.full((int(n),), 0xA5, dtype=torch.uint8, device=self.device)
            items.append(_lib.PackItem(d, which, w.data_ptr(), buf.data_ptr()))
            bufs.append(buf)
            keep.append(w)
        arr = (_lib.PackItem * len(items))(*items)
        self._chk(self.lib.m355_conv3d_pack_batch(C.cast(arr, C.c_void_p), len(items), self._stream()), "conv3d_pack_batch")
        torch.cuda.synchronize()
        return bufs

    def conv3d_fwd(self, x, w, bias=None, add=None, stride=1, pad=1, compute=0, packed=None, softmax=False):
        """packed: buffer from pack_weights(w, x.shape, 0) -> the call uses M355_CONV_W_PACKED;
        softmax: M355_CONV_SOFTMAX (HIP library, descriptors with m355_conv3d_fuses_softmax != 0)"""
        x, w, bias, add = map(self.to, (x, w, bias, add))
        k = w.shape[2]
        d = self.conv_desc(x.shape, w.shape[0], k, stride, pad, compute=compute)
        if softmax:
            assert self.lib.m355_conv3d_fuses_softmax(C.byref(d)) == 1
            d.flags |= _lib.CONV_SOFTMAX
        if packed is not None:
            d.flags = _lib.CONV_W_PACKED
            od = lambda n: (n + 2 * pad - k) // stride + 1
            y = self.empty(x.shape[0], w.shape[0], od(x.shape[2]), od(x.shape[3]), od(x.shape[4]))
            ws = self._ws("conv3d_fwd_workspace", d)
            self._chk(self.fn("conv3d_fwd")(C.byref(d), _p(x), _p(packed), _p(bias), _p(add), _p(y), _p(ws), ws.numel(),
                                            self._stream()), "conv3d_fwd(packed)")
            return y
        od = lambda n: (n + 2 * pad - k) // stride + 1
        y = self.empty(x.shape[0], w.shape[0], od(x.shape[2]), od(x.shape[3]), od(x.shape[4]))
        ws = self._ws("conv3d_fwd_workspace", d)
        self._chk(self.fn("conv3d_fwd")(C.byref(d), _p(x), _p(w), _p(bias), _p(add), _p(y), _p(ws), ws.numel(),
                                        self._stream()), "conv3d_fwd")
        return y

    # ---- c8 tensors of the 16-bit compute modes (HIP library only; dtype: 1 = bf16, 2 = fp16) ----
    def act16_pack(self, x, compute, pad_batch=0):
        """fp32 [N,C,D,H,W] -> c8 tensor (torch 16-bit dtype, shape [N, CB (+pad), S, 8]); pad_batch > 0 leaves
        that many unused channel blocks per sample (a non-dense batch stride)."""
        x = self.to(x)
        N, Cc = x.shape[:2]
        S = x[0, 0].numel()
        CB = (Cc + 7) // 8
        dt = torch.bfloat16 if compute == 1 else torch.float16
        x16 = torch.full((N, CB + pad_batch, S, 8), 7.0, dtype=dt, device=self.device)
        self._chk(self.fn("act16_pack")(_p(x), _p(x16), N, Cc, S, 0, (CB + pad_batch) * S * 8, compute, self._stream()),
                  "act16_pack")
        return x16

    def act16_unpack(self, x16, Cc, spatial, compute):
        N, CBp, S, _ = x16.shape
        x = self.empty(N, Cc, *spatial)
        self._chk(self.fn("act16_unpack")(_p(x16), _p(x), N, Cc, S, CBp * S * 8, 0, compute, self._stream()), "act16_unpack")
        return x

    def conv_plan(self, x_shape, Cout, compute=0, which=0):
        """(kernel family, NTW, GX, split-K) of the 3x3x3 kernel the library picks (m355_conv3d_plan)"""
        d = self.conv_desc(x_shape, Cout, 3, 1, 1, compute=compute)
        out = (C.c_int32 * 4)()
        self._chk(self.lib.m355_conv3d_plan(C.byref(d), which, out), "conv3d_plan")
        return tuple(out)

    def conv3d_fwd_h16(self, x16, Cin, spatial, w, bias=None, add=None, compute=1, groups=None, eps=1e-5, softmax=False):
        """forward on a c8 input; groups != None also returns the fused statistics (mean, rstd); softmax: the
        M355_CONV_SOFTMAX epilogue (asserts that the library offers it for this descriptor)"""
        w, bias, add = map(self.to, (w, bias, add))
        N, CBp, S, _ = x16.shape
        Cout = w.shape[0]
        d = self.conv_desc((N, Cin) + tuple(spatial), Cout, 3, 1, 1, compute=compute)
        if softmax:
            assert self.lib.m355_conv3d_fuses_softmax(C.byref(d)) == 1
            d.flags |= 2
        y = self.empty(N, Cout, *spatial)
        n = self.lib.m355_conv3d_h16_workspace(C.byref(d), 0)
        ws = torch.empty(max(int(n), 16), dtype=torch.uint8, device=self.device)
        part = None
        if groups is not None:
            slots = self.fn("conv3d_stats_slots")(C.byref(d))
            assert slots > 0
            part = self.empty(N, slots, Cout, 2)
        self._chk(self.fn("conv3d_fwd_h16")(C.byref(d), _p(x16), CBp * S * 8, _p(w), _p(bias), _p(add), _p(y), _p(part),
                                            _p(ws), ws.numel(), self._stream()), "conv3d_fwd_h16")
        if groups is None:
            return y
        nd = NormDesc(N, Cout, S, groups, 0, eps, 0.01, 0, 0, 0)
        ns = self.fn("norm_num_stats")(C.byref(nd))
        mean, rstd = self.empty(ns), self.empty(ns)
        nws = self._ws("norm_workspace", nd)
        self._chk(self.fn("norm_stats_from_partials")(C.byref(nd), _p(part), slots, _p(mean), _p(rstd), None, None,
                                                      0.1, _p(nws), nws.numel(), self._stream()),
                  "norm_stats_from_partials")
        return y, mean, rstd

    def conv3d_fwd_h16_c8(self, x16, Cin, spatial, w, bias=None, compute=1, with_stats=False):
        """forward with c8 input AND c8 output; with_stats -> (y16, partials [N, P, Cout, 2])"""
        w, bias = self.to(w), self.to(bias)
        N, CBp, S, _ = x16.shape
        Cout = w.shape[0]
        d = self.conv_desc((N, Cin) + tuple(spatial), Cout, 3, 1, 1, compute=compute)
        y16 = torch.zeros((N, (Cout + 7) // 8, S, 8), dtype=x16.dtype, device=self.device)
        n = self.lib.m355_conv3d_h16_workspace(C.byref(d), 0)
        ws = torch.empty(max(int(n), 16), dtype=torch.uint8, device=self.device)
        part = None
        if with_stats:
            slots = self.fn("conv3d_stats_slots_c8")(C.byref(d))
            assert slots > 0
            part = self.empty(N, slots, Cout, 2)
        self._chk(self.fn("conv3d_fwd_h16_c8")(C.byref(d), _p(x16), CBp * S * 8, _p(w), _p(bias), _p(y16), 0, _p(part),
                                               _p(ws), ws.numel(), self._stream()), "conv3d_fwd_h16_c8")
        return (y16, part) if with_stats else y16

    def norm_act_fwd_c8(self, x16, Cc, mean, rstd, gamma, beta, groups, act, compute, add16=None, eps=1e-5, slope=0.01):
        mean, rstd, gamma, beta = map(self.to, (mean, rstd, gamma, beta))
        N, CB, S, _ = x16.shape
        d = NormDesc(N, Cc, S, groups, act, eps, slope, 0, 0, 0)
        y16 = torch.empty_like(x16)
        self._chk(self.fn("norm_act_fwd_c8")(C.byref(d), _p(x16), 0, _p(mean), _p(rstd), _p(gamma), _p(beta), _p(add16), 0,
                                             _p(y16), 0, compute, self._stream()), "norm_act_fwd_c8")
        return y16

    def act16_channel_partials(self, x16, Cc, compute):
        N, CB, S, _ = x16.shape
        slots = int(self.lib.m355_act16_partials_slots(S))
        part = self.empty(N, slots, Cc, 2)
        self._chk(self.fn("act16_channel_partials")(_p(x16), 0, N, Cc, S, compute, _p(part), self._stream()),
                  "act16_channel_partials")
        return part

    def conv3d_bwd_data_h16(self, dy16, Cout, w, x_shape, compute=1):
        w = self.to(w)
        N, CBp, S, _ = dy16.shape
        d = self.conv_desc(x_shape, Cout, 3, 1, 1, compute=compute)
        dx = self.empty(*x_shape)
        n = self.lib.m355_conv3d_h16_workspace(C.byref(d), 1)
        ws = torch.empty(max(int(n), 16), dtype=torch.uint8, device=self.device)
        self._chk(self.fn("conv3d_bwd_data_h16")(C.byref(d), _p(dy16), CBp * S * 8, _p(w), _p(dx), _p(ws), ws.numel(),
                                                 self._stream()), "conv3d_bwd_data_h16")
        return dx

    def conv3d_bwd_weight_h16(self, x16, dy16, dy, Cin, Cout, spatial, compute=1, with_bias=True):
        """weight (and bias) gradient of the 3x3x3 conv from c8 operands"""
        N = x16.shape[0]
        d = self.conv_desc((N, Cin) + tuple(spatial), Cout, 3, 1, 1, compute=compute)
        dw, db = self.empty(Cout, Cin, 3, 3, 3), (self.empty(Cout) if with_bias else None)
        n = self.lib.m355_conv3d_bwd_weight_h16_workspace(C.byref(d))
        ws = torch.empty(max(int(n), 16), dtype=torch.uint8, device=self.device)
        self._chk(self.fn("conv3d_bwd_weight_h16")(C.byref(d), _p(x16), 0, _p(dy16), 0, _p(self.to(dy)) if with_bias else None,
                                                   _p(dw), _p(db), _p(ws), ws.numel(), self._stream()),
                  "conv3d_bwd_weight_h16")
        return dw, db

    def conv_transpose3d_fwd_h16(self, x16, Cin, spatial, w, bias, compute):
        """k2 s2 conv-transpose c8 -> c8; returns the c8 output [N, CBout, 8S, 8]"""
        w, bias = self.to(w), self.to(bias)
        N, CB, S, _ = x16.shape
        Cout = w.shape[1]
        d = self.conv_desc((N, Cin) + tuple(spatial), Cout, 2, 2, 0)
        y16 = torch.empty((N, (Cout + 7) // 8, 8 * S, 8), dtype=x16.dtype, device=self.device)
        self._chk(self.fn("conv_transpose3d_fwd_h16")(C.byref(d), _p(x16), 0, _p(w), _p(bias), _p(y16), 0, compute,
                                                      self._stream()), "conv_transpose3d_fwd_h16")
        return y16

    # ---- c8-only training flow (round 3) ----
    def act16_pack_scaled(self, x, compute, scale):
        x = self.to(x)
        N, Cc = x.shape[:2]
        S = x[0, 0].numel()
        dt = torch.bfloat16 if compute == 1 else torch.float16
        x16 = torch.full((N, (Cc + 7) // 8, S, 8), 7.0, dtype=dt, device=self.device)
        self._chk(self.fn("act16_pack_scaled")(_p(x), _p(x16), N, Cc, S, 0, 0, compute, float(scale), self._stream()),
                  "act16_pack_scaled")
        return x16

    def act16_unpack_scaled(self, x16, Cc, spatial, compute, scale):
        N, CBp, S, _ = x16.shape
        x = self.empty(N, Cc, *spatial)
        self._chk(self.fn("act16_unpack_scaled")(_p(x16), _p(x), N, Cc, S, CBp * S * 8, 0, compute, float(scale),
                                                 self._stream()), "act16_unpack_scaled")
        return x

    def conv3d_bwd_data_h16_c8(self, dy16, Cout, w, x_shape, compute=1):
        w = self.to(w)
        N, CBp, S, _ = dy16.shape
        d = self.conv_desc(x_shape, Cout, 3, 1, 1, compute=compute)
        dx16 = torch.full((N, (x_shape[1] + 7) // 8, S, 8), 7.0, dtype=dy16.dtype, device=self.device)
        n = self.lib.m355_conv3d_h16_workspace(C.byref(d), 1)
        ws = torch.empty(max(int(n), 16), dtype=torch.uint8, device=self.device)
        self._chk(self.fn("conv3d_bwd_data_h16_c8")(C.byref(d), _p(dy16), CBp * S * 8, _p(w), _p(dx16), 0, _p(ws), ws.numel(),
                                                    self._stream()), "conv3d_bwd_data_h16_c8")
        return dx16

    def conv3d_bwd_weight_c8(self, x16, dy16, Cin, Cout, spatial, compute=1, with_bias=True, unscale=1.0):
        N = x16.shape[0]
        d = self.conv_desc((N, Cin) + tuple(spatial), Cout, 3, 1, 1, compute=compute)
        dw, db = self.empty(Cout, Cin, 3, 3, 3), (self.empty(Cout) if with_bias else None)
        n = self.lib.m355_conv3d_bwd_weight_c8_workspace(C.byref(d))
        ws = torch.empty(max(int(n), 16), dtype=torch.uint8, device=self.device)
        self._chk(self.fn("conv3d_bwd_weight_c8")(C.byref(d), _p(x16), 0, _p(dy16), 0, _p(dw), _p(db), float(unscale), _p(ws),
                                                  ws.numel(), self._stream()), "conv3d_bwd_weight_c8")
        return dw, db

    def norm_act_bwd_c8(self, x16, dy16, dpool16, Cc, spatial, mean, rstd, gamma, beta, groups, act, compute, training=1,
                        unscale=1.0, eps=1e-5, slope=0.01):
        """-> (dx16, dgamma, dbeta); dy16 / dpool16: c8 gradients (either may be None, not both)"""
        mean, rstd, gamma, beta = map(self.to, (mean, rstd, gamma, beta))
        N, S = x16.shape[0], x16.shape[2]
        D, H, W = spatial
        d = NormDesc(N, Cc, S, groups, act, eps, slope, 0, 0, 0)
        dx16 = torch.full_like(x16, 7.0)
        dg = self.empty(Cc) if gamma is not None else None
        db = self.empty(Cc) if gamma is not None else None
        ws = self._ws("norm_workspace", d)
        self._chk(self.fn("norm_act_bwd_c8")(C.byref(d), _p(x16), 0, _p(dy16), 0, _p(dpool16), 0, D, H, W, _p(mean), _p(rstd),
                                             _p(gamma), _p(beta), _p(dx16), 0, _p(dg), _p(db), training, float(unscale),
                                             compute, _p(ws), ws.numel(), self._stream()), "norm_act_bwd_c8")
        return dx16, dg, db

    def avgpool_bwd_h16(self, dpool16, dskip16, Cc, spatial, compute):
        D, H, W = spatial
        N = dpool16.shape[0]
        dx16 = torch.full((N, (Cc + 7) // 8, D * H * W, 8), 7.0, dtype=dpool16.dtype, device=self.device)
        self._chk(self.fn("avgpool3d_2x_bwd_h16")(_p(dpool16), _p(dskip16), _p(dx16), N, Cc, D, H, W, 0, 0, 0, compute,
                                                  self._stream()), "avgpool3d_2x_bwd_h16")
        return dx16

    def upsample_trilinear2x_fwd_h16(self, x16, Cc, spatial, compute, pad_batch=0):
        """c8 -> c8 at twice the resolution; pad_batch: unused channel blocks per sample of the destination"""
        D, H, W = spatial
        N, CBp = x16.shape[:2]
        CB = (Cc + 7) // 8
        y16 = torch.full((N, CB + pad_batch, 8 * D * H * W, 8), 7.0, dtype=x16.dtype, device=self.device)
        self._chk(self.fn("upsample_trilinear2x_fwd_h16")(_p(x16), _p(y16), N, Cc, D, H, W, CBp * D * H * W * 8,
                                                          (CB + pad_batch) * D * H * W * 64, compute, self._stream()),
                  "upsample_trilinear2x_fwd_h16")
        return y16

    def upsample_trilinear2x_bwd_h16(self, dy16, Cc, spatial, compute):
        """`spatial`: the LOW-resolution size"""
        D, H, W = spatial
        N, CBp = dy16.shape[:2]
        dx16 = torch.full((N, (Cc + 7) // 8, D * H * W, 8), 7.0, dtype=dy16.dtype, device=self.device)
        self._chk(self.fn("upsample_trilinear2x_bwd_h16")(_p(dy16), _p(dx16), N, Cc, D, H, W, CBp * D * H * W * 64, 0, compute,
                                                          self._stream()), "upsample_trilinear2x_bwd_h16")
        return dx16

    def act16_channel_scale(self, x16, scale, Cc, compute):
        scale = self.to(scale)
        N, CBp, S, _ = x16.shape
        y16 = torch.full((N, (Cc + 7) // 8, S, 8), 7.0, dtype=x16.dtype, device=self.device)
        self._chk(self.fn("act16_channel_scale")(_p(x16), _p(scale), _p(y16), N, Cc, S, CBp * S * 8, 0, compute,
                                                 self._stream()), "act16_channel_scale")
        return y16

    def s2d_h16(self, x16, full_shape, compute, to_depth, pad_batch=0):
        """space-to-depth (to_depth) / depth-to-space by 2, c8 -> c8; `full_shape` = (N, C, D, H, W) of the full-resolution
        tensor; pad_batch: unused channel blocks per sample of the destination"""
        N, Cc, D, H, W = full_shape
        S = D * H * W
        CBp = x16.shape[1]
        if to_depth:
            y16 = torch.full((N, Cc + pad_batch, S // 8, 8), 7.0, dtype=x16.dtype, device=self.device)
            xbs, ybs = CBp * S * 8, (Cc + pad_batch) * (S // 8) * 8
            fn = self.fn("space_to_depth2_h16")
        else:
            CB = (Cc + 7) // 8
            y16 = torch.full((N, CB + pad_batch, S, 8), 7.0, dtype=x16.dtype, device=self.device)
            xbs, ybs = CBp * (S // 8) * 8, (CB + pad_batch) * S * 8
            fn = self.fn("depth_to_space2_h16")
        self._chk(fn(_p(x16), _p(y16), N, Cc, D, H, W, xbs, ybs, compute, self._stream()), "s2d_h16")
        return y16

    def convt_h16_bwd_supported(self, x_shape, Cout):
        d = self.conv_desc(x_shape, Cout, 2, 2, 0)
        return bool(self.lib.m355_conv_transpose3d_h16_bwd_supported(C.byref(d)))

    def convt_bwd_data_h16(self, dy16, w, x_shape, compute):
        w = self.to(w)
        N, Cin = x_shape[:2]
        S = x_shape[2] * x_shape[3] * x_shape[4]
        d = self.conv_desc(x_shape, w.shape[1], 2, 2, 0)
        dx16 = torch.full((N, (Cin + 7) // 8, S, 8), 7.0, dtype=dy16.dtype, device=self.device)
        self._chk(self.fn("conv_transpose3d_bwd_data_h16")(C.byref(d), _p(dy16), 0, _p(w), _p(dx16), 0, compute,
                                                           self._stream()), "conv_transpose3d_bwd_data_h16")
        return dx16

    def convt_bwd_weight_h16(self, x16, dy16, x_shape, Cout, compute, with_bias=True, unscale=1.0):
        d = self.conv_desc(x_shape, Cout, 2, 2, 0)
        dw = self.empty(x_shape[1], Cout, 2, 2, 2)
        db = self.empty(Cout) if with_bias else None
        n = self.lib.m355_conv_transpose3d_h16_bwd_workspace(C.byref(d))
        ws = torch.empty(max(int(n), 16), dtype=torch.uint8, device=self.device)
        self._chk(self.fn("conv_transpose3d_bwd_weight_h16")(C.byref(d), _p(x16), 0, _p(dy16), 0, _p(dw), _p(db), float(unscale),
                                                             compute, _p(ws), ws.numel(), self._stream()),
                  "conv_transpose3d_bwd_weight_h16")
        return dw, db

    def conv3d_fwd_stats(self, x, w, bias=None, groups=0, eps=1e-5):
        """fused conv + statistics: returns (y, mean, rstd) of the normalisation that follows the conv, or
        None when this backend has no fused statistics for the shape"""
        x, w = self.to(x), self.to(w)
        bias = self.to(bias) if bias is not None else None
        N, Cin, D, H, W = x.shape
        Cout = w.shape[0]
        d = self.conv_desc(x.shape, Cout, 3, 1, 1)
        slots = self.fn("conv3d_stats_slots")(C.byref(d))
        if slots <= 0:
            return None
        y = self.empty(N, Cout, D, H, W)
        part = self.empty(N, slots, Cout, 2)
        ws = self._ws("conv3d_fwd_workspace", d)
        self._chk(self.fn("conv3d_fwd_stats")(C.byref(d), _p(x), _p(w), _p(bias), None, _p(y), _p(part), _p(ws),
                                              ws.numel(), self._stream()), "conv3d_fwd_stats")
        nd = NormDesc(N, Cout, D * H * W, groups, 0, eps, 0.01, 0, 0, 0)
        ns = self.fn("norm_num_stats")(C.byref(nd))
        mean, rstd = self.empty(ns), self.empty(ns)
        nws = self._ws("norm_workspace", nd)
        self._chk(self.fn("norm_stats_from_partials")(C.byref(nd), _p(part), slots, _p(mean), _p(rstd), None, None,
                                                      0.1, _p(nws), nws.numel(), self._stream()),
                  "norm_stats_from_partials")
        return y, mean, rstd

    def conv3d_bwd_data(self, dy, w, x_shape, stride=1, pad=1, compute=0, packed=None):
        dy, w = self.to(dy), self.to(w)
        d = self.conv_desc(x_shape, w.shape[0], w.shape[2], stride, pad, compute=compute)
        dx = self.empty(*x_shape)
        ws = self._ws("conv3d_bwd_data_workspace", d)
        if packed is not None:
            d.flags, w = _lib.CONV_W_PACKED, packed
        self._chk(self.fn("conv3d_bwd_data")(C.byref(d), _p(dy), _p(w), _p(dx), _p(ws), ws.numel(), self._stream()),
                  "conv3d_bwd_data")
        return dx

    def conv3d_bwd_weight(self, x, dy, k, stride=1, pad=1, with_bias=True, compute=0):
        x, dy = self.to(x), self.to(dy)
        Cout = dy.shape[1]
        d = self.conv_desc(x.shape, Cout, k, stride, pad, compute=compute)
        dw = self.empty(Cout, x.shape[1], k, k, k)
        db = self.empty(Cout) if with_bias else None
        ws = self._ws("conv3d_bwd_weight_workspace", d)
        self._chk(self.fn("conv3d_bwd_weight")(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(ws), ws.numel(),
                                               self._stream()), "conv3d_bwd_weight")
        return dw, db

    def convt_fwd(self, x, w, bias=None, stride=2, pad=0, out_pad=0, compute=0):
        x, w, bias = map(self.to, (x, w, bias))
        k = w.shape[2]
        d = self.conv_desc(x.shape, w.shape[1], k, stride, pad, out_pad, compute=compute)
        od = lambda n: (n - 1) * stride - 2 * pad + k + out_pad
        y = self.empty(x.shape[0], w.shape[1], od(x.shape[2]), od(x.shape[3]), od(x.shape[4]))
        ws = self._ws("conv_transpose3d_workspace", d)
        self._chk(self.fn("conv_transpose3d_fwd")(C.byref(d), _p(x), _p(w), _p(bias), _p(y), _p(ws), ws.numel(),
                                                  self._stream()), "convt_fwd")
        return y

    def convt_bwd_data(self, dy, w, x_shape, stride=2, pad=0, out_pad=0):
        dy, w = self.to(dy), self.to(w)
        d = self.conv_desc(x_shape, w.shape[1], w.shape[2], stride, pad, out_pad)
        dx = self.empty(*x_shape)
        ws = self._ws("conv_transpose3d_workspace", d)
        self._chk(self.fn("conv_transpose3d_bwd_data")(C.byref(d), _p(dy), _p(w), _p(dx), _p(ws), ws.numel(),
                                                       self._stream()), "convt_bwd_data")
        return dx

    def convt_bwd_weight(self, x, dy, k, stride=2, pad=0, out_pad=0, with_bias=True):
        x, dy = self.to(x), self.to(dy)
        Cout = dy.shape[1]
        d = self.conv_desc(x.shape, Cout, k, stride, pad, out_pad)
        dw = self.empty(x.shape[1], Cout, k, k, k)
        db = self.empty(Cout) if with_bias else None
        ws = self._ws("conv_transpose3d_workspace", d)
        self._chk(self.fn("conv_transpose3d_bwd_weight")(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(ws),
                                                         ws.numel(), self._stream()), "convt_bwd_weight")
        return dw, db

    # ------------------------------------------------------------------ norm
    @staticmethod
    def norm_desc(x, groups, act=0, eps=1e-5, slope=0.01):
        N, Cc = x.shape[:2]
        S = x.numel() // (N * Cc)
        return NormDesc(N, Cc, S, groups, act, eps, slope, 0, 0, 0)

    def norm_stats(self, x, groups, eps=1e-5, running=None, momentum=0.1):
        x = self.to(x)
        d = self.norm_desc(x, groups, eps=eps)
        ns = self.fn("norm_num_stats")(C.byref(d))
        mean, rstd = self.empty(ns), self.empty(ns)
        rm = rv = None
        if running is not None:
            rm, rv = self.to(running[0]).clone(), self.to(running[1]).clone()
        ws = self._ws("norm_workspace", d)
        self._chk(self.fn("norm_stats")(C.byref(d), _p(x), _p(mean), _p(rstd), _p(rm), _p(rv), momentum, _p(ws),
                                        ws.numel(), self._stream()), "norm_stats")
        return mean, rstd, rm, rv

    def norm_act_fwd(self, x, mean, rstd, gamma, beta, groups, act, add=None, eps=1e-5, slope=0.01):
        x, mean, rstd, gamma, beta, add = map(self.to, (x, mean, rstd, gamma, beta, add))
        d = self.norm_desc(x, groups, act, eps, slope)
        y = torch.empty_like(x)
        self._chk(self.fn("norm_act_fwd")(C.byref(d), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(add),
                                          _p(y), self._stream()), "norm_act_fwd")
        return y

    def norm_act_fwd_h16(self, x, mean, rstd, gamma, beta, groups, act, compute, add=None, want_f32=False, eps=1e-5,
                         slope=0.01):
        """c8 output (torch 16-bit tensor [N, CB, S, 8]) and optionally the fp32 NCDHW output as well"""
        x, mean, rstd, gamma, beta, add = map(self.to, (x, mean, rstd, gamma, beta, add))
        d = self.norm_desc(x, groups, act, eps, slope)
        N, Cc = x.shape[:2]
        S = x[0, 0].numel()
        y16 = torch.empty((N, (Cc + 7) // 8, S, 8), dtype=torch.bfloat16 if compute == 1 else torch.float16,
                          device=self.device)
        y = torch.empty_like(x) if want_f32 else None
        self._chk(self.fn("norm_act_fwd_h16")(C.byref(d), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(add), _p(y),
                                              _p(y16), 0, compute, self._stream()), "norm_act_fwd_h16")
        return y16, y

    def avgpool_fwd_h16(self, x16, Cc, spatial, compute):
        N, CB, S, _ = x16.shape
        D, H, W = spatial
        y16 = torch.empty((N, CB, S // 8, 8), dtype=x16.dtype, device=self.device)
        self._chk(self.fn("avgpool3d_2x_fwd_h16")(_p(x16), _p(y16), N, Cc, D, H, W, 0, 0, compute, self._stream()),
                  "avgpool3d_2x_fwd_h16")
        return y16

    def norm_act_bwd(self, x, dy, mean, rstd, gamma, beta, groups, act, training=1, eps=1e-5, slope=0.01):
        x, dy, mean, rstd, gamma, beta = map(self.to, (x, dy, mean, rstd, gamma, beta))
        d = self.norm_desc(x, groups, act, eps, slope)
        dx = torch.empty_like(x)
        dg = self.empty(x.shape[1]) if gamma is not None else None
        db = self.empty(x.shape[1]) if gamma is not None else None
        ws = self._ws("norm_workspace", d)
        self._chk(self.fn("norm_act_bwd")(C.byref(d), _p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(beta),
                                          _p(dx), _p(dg), _p(db), training, _p(ws), ws.numel(), self._stream()),
                  "norm_act_bwd")
        return dx, dg, db

    def norm_act_pool_fwd(self, x, mean, rstd, gamma, beta, groups, act, eps=1e-5, slope=0.01):
        """norm + activation with AvgPool3d(2, 2) of the result as second output -> (y, pooled)"""
        x, mean, rstd, gamma, beta = map(self.to, (x, mean, rstd, gamma, beta))
        d = self.norm_desc(x, groups, act, eps, slope)
        N, Cc, D, H, W = x.shape
        y, pooled = torch.empty_like(x), self.empty(N, Cc, D // 2, H // 2, W // 2)
        self._chk(self.fn("norm_act_pool_fwd")(C.byref(d), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(y), _p(pooled),
                                               0, D, H, W, self._stream()), "norm_act_pool_fwd")
        return y, pooled

    def norm_act_bwd_h16(self, x, dy, mean, rstd, gamma, beta, groups, act, compute, training=1, eps=1e-5, slope=0.01):
        """norm backward that also emits dx as c8 -> (dx, dgamma, dbeta, dx16 [N, CB, S, 8])"""
        x, dy, mean, rstd, gamma, beta = map(self.to, (x, dy, mean, rstd, gamma, beta))
        d = self.norm_desc(x, groups, act, eps, slope)
        N, Cc = x.shape[:2]
        S = x[0, 0].numel()
        dx = torch.empty_like(x)
        dg = self.empty(Cc) if gamma is not None else None
        db = self.empty(Cc) if gamma is not None else None
        dx16 = torch.empty((N, (Cc + 7) // 8, S, 8), dtype=torch.bfloat16 if compute == 1 else torch.float16, device=self.device)
        ws = self._ws("norm_workspace", d)
        self._chk(self.fn("norm_act_bwd_h16")(C.byref(d), _p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dx),
                                              _p(dg), _p(db), training, _p(dx16), 0, compute, _p(ws), ws.numel(),
                                              self._stream()), "norm_act_bwd_h16")
        return dx, dg, db, dx16

    # -------------------------------------------------- pool / upsample / softmax
    def avgpool_fwd(self, x):
        x = self.to(x)
        N, Cc, D, H, W = x.shape
        y = self.empty(N, Cc, D // 2, H // 2, W // 2)
        self._chk(self.fn("avgpool3d_2x_fwd")(_p(x), _p(y), N, Cc, D, H, W, 0, 0, self._stream()), "avgpool_fwd")
        return y

    def avgpool_bwd(self, dy, x_shape):
        dy = self.to(dy)
        N, Cc, D, H, W = x_shape
        dx = self.empty(*x_shape)
        self._chk(self.fn("avgpool3d_2x_bwd")(_p(dy), _p(dx), N, Cc, D, H, W, 0, 0, self._stream()), "avgpool_bwd")
        return dx

    def avgpool_bwd_add(self, dy, add, x_shape):
        dy, add = self.to(dy), self.to(add)
        N, Cc, D, H, W = x_shape
        dx = self.empty(*x_shape)
        self._chk(self.fn("avgpool3d_2x_bwd_add")(_p(dy), _p(add), _p(dx), N, Cc, D, H, W, 0, 0, 0, self._stream()),
                  "avgpool_bwd_add")
        return dx

    def upsample_fwd(self, x):
        x = self.to(x)
        N, Cc, D, H, W = x.shape
        y = self.empty(N, Cc, 2 * D, 2 * H, 2 * W)
        self._chk(self.fn("upsample_trilinear2x_fwd")(_p(x), _p(y), N, Cc, D, H, W, 0, 0, self._stream()),
                  "upsample_fwd")
        return y

    def upsample_bwd(self, dy, x_shape):
        dy = self.to(dy)
        N, Cc, D, H, W = x_shape
        dx = self.empty(*x_shape)
        self._chk(self.fn("upsample_trilinear2x_bwd")(_p(dy), _p(dx), N, Cc, D, H, W, 0, 0, self._stream()),
                  "upsample_bwd")
        return dx

    def space_to_depth(self, x):
        x = self.to(x)
        N, Cc, D, H, W = x.shape
        y = self.empty(N, Cc * 8, D // 2, H // 2, W // 2)
        self._chk(self.fn("space_to_depth2")(_p(x), _p(y), N, Cc, D, H, W, 0, 0, self._stream()), "space_to_depth2")
        return y

    def depth_to_space(self, x):
        x = self.to(x)
        N, C8, D, H, W = x.shape
        y = self.empty(N, C8 // 8, 2 * D, 2 * H, 2 * W)
        self._chk(self.fn("depth_to_space2")(_p(x), _p(y), N, C8 // 8, 2 * D, 2 * H, 2 * W, 0, 0, self._stream()),
                  "depth_to_space2")
        return y

    def blur_weight_fwd(self, w, scale, standardize, transposed):
        w, scale = self.to(w), self.to(scale)
        A, B = w.shape[:2]
        wexp = self.empty(8 * B, A, 3, 3, 3) if transposed else self.empty(A, 8 * B, 3, 3, 3)
        ms = self.empty(A, 2)
        self._chk(self.fn("blur_weight_fwd")(_p(w), _p(scale), _p(wexp), _p(ms), A, B, int(standardize),
                                             int(transposed), self._stream()), "blur_weight_fwd")
        return wexp, ms

    def blur_weight_bwd(self, dwexp, w, scale, ms, standardize, transposed):
        dwexp, w, scale, ms = map(self.to, (dwexp, w, scale, ms))
        A, B = w.shape[:2]
        dw = torch.empty_like(w)
        self._chk(self.fn("blur_weight_bwd")(_p(dwexp), _p(w), _p(scale), _p(ms), _p(dw), A, B, int(standardize),
                                             int(transposed), self._stream()), "blur_weight_bwd")
        return dw

    def weight_standardize_fwd(self, w):
        w = self.to(w)
        A, n = w.shape[0], w[0].numel()
        wn, ms = torch.empty_like(w), self.empty(A, 2)
        self._chk(self.fn("weight_standardize_fwd")(_p(w), _p(wn), _p(ms), A, n, self._stream()), "weight_standardize_fwd")
        return wn, ms

    def weight_standardize_bwd(self, dwn, w, ms):
        dwn, w, ms = map(self.to, (dwn, w, ms))
        dw = torch.empty_like(w)
        self._chk(self.fn("weight_standardize_bwd")(_p(dwn), _p(w), _p(ms), _p(dw), w.shape[0], w[0].numel(),
                                                    self._stream()), "weight_standardize_bwd")
        return dw

    def patch_gather_padded(self, vol, loc, ps, border, mode, value=0.0):
        vol, loc = self.to(vol), self.to(loc.to(torch.int32))
        Cc, V0, V1, V2 = vol.shape
        P = loc.shape[0]
        out = self.empty(P, Cc, *ps)
        self._chk(self.fn("patch_gather_padded")(_p(vol), _p(loc), _p(out), P, Cc, V0, V1, V2, ps[0], ps[1], ps[2],
                                                 border[0], border[1], border[2], mode, float(value), self._stream()),
                  "patch_gather_padded")
        return out

    def patch_aggregate_grid(self, tiles, axes, vshape, border=(0, 0, 0)):
        tiles = self.to(tiles)
        P, Cc, ps0, ps1, ps2 = tiles.shape
        starts = torch.tensor([v for a in axes for v in a], dtype=torch.int32, device=self.device)
        out = self.empty(Cc, *vshape)
        self._chk(self.fn("patch_aggregate_grid")(_p(tiles), _p(starts), len(axes[0]), len(axes[1]), len(axes[2]), _p(out), Cc,
                                                  *vshape, ps0, ps1, ps2, *border, self._stream()), "patch_aggregate_grid")
        return out

    def patch_finalize_crop(self, accum, count, border):
        accum, count = self.to(accum), self.to(count)
        Cc, P0, P1, P2 = accum.shape
        out = self.empty(Cc, P0 - 2 * border[0], P1 - 2 * border[1], P2 - 2 * border[2])
        self._chk(self.fn("patch_finalize_crop")(_p(accum), _p(count), _p(out), Cc, P0, P1, P2, border[0], border[1],
                                                 border[2], self._stream()), "patch_finalize_crop")
        return out

    # ---------------------------------------------------------------- ensembles
    @staticmethod
    def _i3(v):
        return (C.c_int32 * 3)(*[int(a) for a in v])

    def flip_permute(self, x, perm, flip_mask):
        x = self.to(x)
        N, Cc = x.shape[:2]
        sp = x.shape[2:]
        y = self.empty(N, Cc, *[sp[p] for p in perm])
        self._chk(self.fn("flip_permute")(_p(x), _p(y), N, Cc, self._i3(sp), self._i3(perm), flip_mask, self._stream()),
                  "flip_permute")
        return y

    def ensemble(self, preds, transforms, canonical_spatial, strategy):
        """preds[e]: member prediction in member orientation; transforms[e] = (perm, flip_mask) -> (result, votes)"""
        mode = 0 if strategy == "mean" else 1
        N, Cc = preds[0].shape[:2]
        shape = (N, Cc) + tuple(canonical_spatial)
        acc = self.empty(*shape) if mode == 0 else self.empty(*shape, dtype=torch.int32)
        for e, (p, (perm, fm)) in enumerate(zip(preds, transforms)):
            p = self.to(p)
            self._chk(self.fn("ensemble_accumulate")(_p(p), _p(acc) if mode == 0 else None, _p(acc) if mode == 1 else None,
                                                     N, Cc, self._i3(canonical_spatial), self._i3(perm), fm, mode,
                                                     1 if e == 0 else 0, self._stream()), "ensemble_accumulate")
        S = shape[2] * shape[3] * shape[4]
        out = self.empty(*shape) if mode == 0 else self.empty(*shape, dtype=torch.int64)
        self._chk(self.fn("ensemble_finalize")(_p(acc) if mode == 0 else None, _p(acc) if mode == 1 else None,
                                               _p(out) if mode == 0 else None, _p(out) if mode == 1 else None, N, Cc, S,
                                               len(preds), mode, self._stream()), "ensemble_finalize")
        return out, acc

    def softmax_fwd(self, x, inner=1, diag_bias=0.0):
        x = self.to(x)
        N, Ct = x.shape[:2]
        S = x.numel() // (N * Ct)
        y = torch.empty_like(x)
        self._chk(self.fn("softmax_fwd")(_p(x), _p(y), N, Ct // inner, inner, S, diag_bias, self._stream()),
                  "softmax_fwd")
        return y

    def softmax_bwd(self, y, dy, inner=1):
        y, dy = self.to(y), self.to(dy)
        N, Ct = y.shape[:2]
        S = y.numel() // (N * Ct)
        dx = torch.empty_like(y)
        self._chk(self.fn("softmax_bwd")(_p(y), _p(dy), _p(dx), N, Ct // inner, inner, S, self._stream()),
                  "softmax_bwd")
        return dx

    # ------------------------------------------------------------------ loss
    def loss_fwd(self, p, t, dice_weight=0.5, cw=None, square=True):
        p, t, cw = map(self.to, (p, t, cw))
        N, Cc = p.shape[:2]
        S = p.numel() // (N * Cc)
        out3, sums = self.empty(3), self.empty(N * Cc * 4)
        nws = self.lib.m355_hybrid_loss_workspace(N, Cc, S) if self.backend == "hip" else 16
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=self.device)
        self._chk(self.fn("hybrid_loss_fwd")(_p(p), _p(t), N, Cc, S, dice_weight, _p(cw), int(square), _p(out3),
                                             _p(sums), _p(ws), ws.numel(), self._stream()), "loss_fwd")
        return out3, sums

    def loss_bwd(self, p, t, sums, dloss=1.0, dice_weight=0.5, cw=None, square=True):
        p, t, sums, cw = map(self.to, (p, t, sums, cw))
        N, Cc = p.shape[:2]
        S = p.numel() // (N * Cc)
        g = torch.tensor([dloss], dtype=torch.float32, device=self.device)
        dp = torch.empty_like(p)
        self._chk(self.fn("hybrid_loss_bwd")(_p(p), _p(t), _p(sums), _p(g), N, Cc, S, dice_weight, _p(cw),
                                             int(square), _p(dp), self._stream()), "loss_bwd")
        return dp

    # ------------------------------------------------------ patches / evaluation
    def patch_gather(self, vol, loc, ps):
        vol, loc = self.to(vol), self.to(loc.to(torch.int32))
        Cc, V0, V1, V2 = vol.shape
        P = loc.shape[0]
        out = self.empty(P, Cc, *ps)
        self._chk(self.fn("patch_gather")(_p(vol), _p(loc), _p(out), P, Cc, V0, V1, V2, *ps, self._stream()),
                  "patch_gather")
        return out

    def patch_aggregate(self, patches, loc, vshape):
        patches, loc = self.to(patches), self.to(loc.to(torch.int32))
        P, Cc = patches.shape[:2]
        ps = patches.shape[2:]
        accum = torch.zeros((Cc,) + tuple(vshape), device=self.device)
        count = torch.zeros(tuple(vshape), device=self.device)
        self._chk(self.fn("patch_accumulate")(_p(patches), _p(loc), _p(accum), _p(count), P, Cc, *vshape, *ps,
                                              self._stream()), "patch_accumulate")
        out = torch.empty_like(accum)
        self._chk(self.fn("patch_finalize")(_p(accum), _p(count), _p(out), Cc, count.numel(), self._stream()),
                  "patch_finalize")
        return out, count

    def argmax_confusion(self, prob, target):
        prob, target = self.to(prob), self.to(target.to(torch.int32))
        N, Cc = prob.shape[:2]
        S = prob.numel() // (N * Cc)
        am = torch.empty(target.shape, dtype=torch.int32, device=self.device)
        counts = torch.empty((N, Cc, 4), dtype=torch.int64, device=self.device)
        self._chk(self.fn("argmax_confusion")(_p(prob), _p(target), _p(am), _p(counts), N, Cc, S, self._stream()),
                  "argmax_confusion")
        return am, counts
