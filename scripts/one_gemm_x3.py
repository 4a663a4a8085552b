"""A few launches of chosen bf16x3 variants on one shape (run under rocprofv3 --pmc ... to read the SQ counters).
    python scripts/one_gemm_x3.py M N K a_layout b_layout 31,40,41"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
M, N, K, al, bl = (int(v) for v in sys.argv[1:6])
variants = [int(v) for v in sys.argv[6].split(",")]
a = torch.randn((M, K) if al == 0 else (K, M), device="cuda")
b = torch.randn((N, K) if bl == 0 else (K, N), device="cuda")
out = torch.empty(M, N, device="cuda")
for v in variants:
    for _ in range(3):
        ops.gemm(a, b, a_layout=al, b_layout=bl, out=out, variant=v, precision=ops.PREC_BF16X3 if v else ops.PREC_F32)
torch.cuda.synchronize()
