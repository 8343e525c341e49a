#!/usr/bin/env python3
"""Per-layer timing of the conv kernels through the C ABI (HIP events on the launch stream).
usage: python tools/conv_bench.py [fwd] [bwd_data] [bwd_weight] [--layers cfg2|quick]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raw_ops import RawOps  # noqa: E402

CFG2 = [  # name, Cin, Cout, spatial
    ("d0.c0", 4, 32, 128), ("d0.c1", 32, 32, 128), ("u0.c0", 96, 32, 128), ("out", 32, 3, 128),
    ("d1.c0", 32, 64, 64), ("d1.c1", 64, 64, 64), ("u1.c0", 192, 64, 64),
    ("d2.c0", 64, 128, 32), ("d2.c1", 128, 128, 32), ("u2.c0", 384, 128, 32),
    ("d3.c0", 128, 256, 16), ("d3.c1", 256, 256, 16), ("u3.c0", 576, 256, 16),
    ("d4.c0", 256, 320, 8), ("d4.c1", 320, 320, 8),
]
QUICK = [("d0.c1", 32, 32, 128), ("u0.c0", 96, 32, 128), ("d1.c1", 64, 64, 64), ("u3.c0", 576, 256, 16)]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ops = [a for a in sys.argv[1:] if not a.startswith("--")] or ["fwd", "bwd_data", "bwd_weight"]
    layers = QUICK if "--quick" in sys.argv else CFG2
    for a in sys.argv[1:]:
        if a.startswith("--only="):   # --only=out,d0.c0
            layers = [l for l in CFG2 if l[0] in a[7:].split(",")]
    hip = RawOps("hip")
    compute = 1 if "--bf16" in sys.argv else (2 if "--fp16" in sys.argv else 0)
    h16 = compute != 0
    if "--f32x3" in sys.argv:   # fp32 tensors, bf16 matrix pipe on an exact 3-way split (M355_COMPUTE_F32X3)
        compute = 3
    tot = {o: [0.0, 0.0] for o in ops}
    print(f"{'layer':8s} {'Cin':>4s} {'Cout':>4s} {'S':>4s} " + " ".join(f"{o + ' ms':>14s} {'TF':>6s}" for o in ops))
    for name, ci, co, sp in layers:
        x = torch.randn(1, ci, sp, sp, sp, device="cuda")
        w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
        dy = torch.randn(1, co, sp, sp, sp, device="cuda")
        # operand statistics (power probe: the MFMA clock depends on how many operand bits toggle)
        if "--zeros" in sys.argv:
            x, w, dy = torch.zeros_like(x), torch.zeros_like(w), torch.zeros_like(dy)
        elif "--relu" in sys.argv:       # what the convs of the network actually read: post-ReLU activations
            x = torch.relu(x)
        elif "--const" in sys.argv:
            x, w, dy = torch.ones_like(x), torch.full_like(w, 0.05), torch.ones_like(dy)
        flops = 2.0 * 27 * ci * co * sp ** 3
        row = f"{name:8s} {ci:4d} {co:4d} {sp:4d} "
        if h16:   # 16-bit modes: the model path hands over c8 tensors (packed outside the timing)
            x16, dy16 = hip.act16_pack(x, compute), hip.act16_pack(dy, compute)
        for o in ops:
            if o == "fwd" and h16 and ci > 4:   # c8 in, c8 out (the inference flow); --f32out: fp32 NCDHW output
                if "--f32out" in sys.argv:
                    ms = timeit(lambda: hip.conv3d_fwd_h16(x16, ci, (sp, sp, sp), w, compute=compute))
                else:
                    ms = timeit(lambda: hip.conv3d_fwd_h16_c8(x16, ci, (sp, sp, sp), w, compute=compute))
            elif o == "bwd_data" and h16 and ci > 4:
                if "--f32out" in sys.argv:
                    ms = timeit(lambda: hip.conv3d_bwd_data_h16(dy16, co, w, x.shape, compute=compute))
                else:   # c8 in, c8 out (the c8 training flow)
                    ms = timeit(lambda: hip.conv3d_bwd_data_h16_c8(dy16, co, w, tuple(x.shape), compute))
            elif o == "fwd":
                ms = timeit(lambda: hip.conv3d_fwd(x, w, compute=compute))
            elif o == "bwd_data":
                ms = timeit(lambda: hip.conv3d_bwd_data(dy, w, x.shape, compute=compute))
            elif h16 and ci > 4 and co > 4 and "--f32in" not in sys.argv:   # c8 operands (the 16-bit training flow)
                ms = timeit(lambda: hip.conv3d_bwd_weight_h16(x16, dy16, dy, ci, co, (sp, sp, sp), compute, with_bias=False))
            elif h16 and "--f32in" not in sys.argv:   # the edge layers of the c8 training flow (conv3_bww_c8_small_kernel)
                ms = timeit(lambda: hip.conv3d_bwd_weight_c8(x16, dy16, ci, co, (sp, sp, sp), compute, with_bias=False))
            else:
                ms = timeit(lambda: hip.conv3d_bwd_weight(x, dy, 3, with_bias=False, compute=compute))
            tot[o][0] += ms
            tot[o][1] += flops
            row += f"{ms:14.3f} {flops / ms / 1e9:6.1f} "
        print(row, flush=True)
    print("total    " + " " * 15 + " ".join(f"{tot[o][0]:14.3f} {tot[o][1] / tot[o][0] / 1e9:6.1f}" for o in ops))


if __name__ == "__main__":
    main()
