"""Timing-only knock-outs of the LDS-DMA GEMM (results are wrong on purpose): run once per DKD_GEMM_KNOCKOUT in
{0, 2 (no DMA), 4 (no DMA of B), 8 (no barrier)} and compare."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dinov2_distillation_amd import ops
for (M, N, K) in [(16384, 384, 1536), (12544, 1536, 384), (3136, 768, 3072)]:
    a = torch.randn(M, K).cuda(); b = torch.randn(N, K).cuda(); out = torch.empty(M, N, device="cuda")
    for _ in range(3): ops.gemm(a, b, out=out, variant=101)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ops.gemm(a, b, out=out, variant=101)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 3)
    us = sorted(ts)[3]
    print("DBG=%s %d %d %d: %.1f us %.0f TF" % (os.environ.get("DKD_GEMM_KNOCKOUT", "0"), M, N, K, us, 2.0 * M * N * K / us / 1e6))
