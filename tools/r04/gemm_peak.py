"""What does the vendor bf16 GEMM reach on this box (random vs zero operands)?  Calibrates the practical MFMA ceiling
under the power limit: the 16-bit conv kernels are priced against the nominal 2.5 PFLOP/s."""
import torch, time
def bench(n, kind, dtype=torch.bfloat16, iters=20):
    a = torch.randn(n, n, device="cuda", dtype=dtype) if kind == "randn" else torch.zeros(n, n, device="cuda", dtype=dtype)
    b = torch.randn(n, n, device="cuda", dtype=dtype) if kind == "randn" else torch.zeros(n, n, device="cuda", dtype=dtype)
    if kind == "relu": a = torch.relu(torch.randn(n, n, device="cuda", dtype=dtype)); b = torch.randn(n, n, device="cuda", dtype=dtype)
    for _ in range(3): torch.matmul(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): torch.matmul(a, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2 * n ** 3 / ms / 1e9
for n in (4096, 8192):
    for kind in ("randn", "relu", "zeros"):
        print(f"bf16 GEMM {n}^3 {kind:6s}: {bench(n, kind):7.1f} TFLOP/s")
for kind in ("randn", "zeros"):
    print(f"fp32 GEMM 8192^3 {kind:6s}: {bench(8192, kind, torch.float32, 5):7.1f} TFLOP/s")
