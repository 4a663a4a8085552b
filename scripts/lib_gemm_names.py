"""Which vendor kernels torch.matmul (strict fp32) picks for the step's main GEMM shapes: run under rocprofv3 --kernel-trace."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(16384, 384, 1536, 0, 0), (16384, 384, 1536, 0, 1), (16448, 384, 1536, 0, 0), (16384, 1536, 384, 0, 0),
          (16384, 384, 384, 0, 0), (12544, 384, 1536, 0, 0), (50176, 192, 768, 0, 0), (3136, 768, 3072, 0, 0)]
for M, N, K, al, bl in shapes:
    a = torch.randn(M, K, device="cuda")
    b = torch.randn((N, K) if bl == 0 else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(3):
        torch.matmul(a, b.t() if bl == 0 else b, out=out)
torch.cuda.synchronize()
