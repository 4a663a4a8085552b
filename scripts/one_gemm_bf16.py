"""A few launches of the bf16 GEMM (and the vendor library's) on one shape, for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (12544, 384, 1536)))
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(4): ops.gemm_bf16(a, w, out=out)
torch.cuda.synchronize()
for _ in range(4): torch.matmul(a, w.t(), out=out)
torch.cuda.synchronize()
