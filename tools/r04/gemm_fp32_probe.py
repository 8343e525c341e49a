import torch
n = 8192
a = torch.randn(n, n, device="cuda"); b = torch.randn(n, n, device="cuda")
for _ in range(5): c = torch.matmul(a, b)
torch.cuda.synchronize()
# exactness probe: compare one output row with float64
ref = (a[:4].double() @ b.double())
print("max rel err vs fp64 of the fp32 GEMM:", ((c[:4].double() - ref).abs().max() / ref.abs().max()).item())
ab = torch.randn(n, n, device="cuda", dtype=torch.bfloat16); bb = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
for _ in range(5): torch.matmul(ab, bb)
torch.cuda.synchronize()
