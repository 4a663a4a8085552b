import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
M, N, K = 16384, 1536, 384
a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
out = torch.empty(M, N, device="cuda")
for v in (2, 3, 1):
    for _ in range(3): ops.gemm(a, b, out=out, variant=v)
torch.cuda.synchronize()
