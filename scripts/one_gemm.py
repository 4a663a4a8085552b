"""A few launches of chosen GEMM variants on one shape (run under rocprofv3 --pmc ... to read the SQ counters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16384, 384, 1536)))
variants = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [44, 2]
a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
out = torch.empty(M, N, device="cuda")
for v in variants:
    for _ in range(3): ops.gemm(a, b, out=out, variant=v)
torch.cuda.synchronize()
if os.environ.get("LIB"):
    torch.backends.cuda.matmul.allow_tf32 = False
    for _ in range(3): torch.matmul(a, b.t(), out=out)
    torch.cuda.synchronize()
