"""What HBM rates do plain streaming kernels reach on this box?  (the bound the normalisation passes are priced against)"""
import torch
def t(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for mb in (134, 536):
    n = mb * 1024 * 1024 // 2
    a = torch.randn(n, device="cuda").bfloat16(); b = torch.randn(n, device="cuda").bfloat16(); c = torch.empty_like(a)
    s = t(lambda: torch.add(a, b, out=c));  print(f"{mb:4d} MB bf16 tensors: c = a + b      {3 * n * 2 / s / 1e12:5.2f} TB/s")
    s = t(lambda: c.copy_(a));              print(f"{mb:4d} MB bf16 tensors: copy           {2 * n * 2 / s / 1e12:5.2f} TB/s")
    s = t(lambda: torch.relu_(c));          print(f"{mb:4d} MB bf16 tensors: relu_ in place {2 * n * 2 / s / 1e12:5.2f} TB/s")
    af = a.float()
    s = t(lambda: af.sum());                print(f"{2*mb:4d} MB fp32 tensor : sum (read)     {n * 4 / s / 1e12:5.2f} TB/s")
    s = t(lambda: a.sum());                 print(f"{mb:4d} MB bf16 tensor : sum (read)     {n * 2 / s / 1e12:5.2f} TB/s")
    del af
