import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
g = torch.Generator().manual_seed(0)
for M, N, K in ((514, 1152, 384), (514, 384, 1536)):
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.02
    ref16 = (a.bfloat16().double() @ b.bfloat16().double().t())
    ref32 = a.double() @ b.double().t()
    trunc = lambda t: (t.view(torch.int32) & -65536).view(torch.float32)
    reft = trunc(a).double() @ trunc(b).double().t()
    out = ops.gemm(a.cuda(), b.cuda(), precision=ops.PREC_BF16).cpu().double()
    rel = lambda x, y: ((x - y).norm() / y.norm()).item()
    print(M, N, K, "vs RNE-bf16 %.2e  vs truncated-bf16 %.2e  vs fp32 %.2e   (RNE vs fp32 %.2e)" % (rel(out, ref16), rel(out, reft), rel(out, ref32), rel(ref16, ref32)))
