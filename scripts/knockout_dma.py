"""Timing-only knock-outs of the LDS-DMA GEMM (results are wrong on purpose): variant 101 = the production 64x64 BK32 kernel,
130 = no DMA at all, 131 = no DMA of B, 132 = no barrier (compile-time switches: as run-time tests they cost the kernel 6 %)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
vs = [101, 130, 131, 132]
for (M, N, K) in [(16384, 384, 1536), (12544, 1536, 384), (3136, 768, 3072)]:
    a = torch.randn(M, K).cuda(); b = torch.randn(N, K).cuda(); out = torch.empty(M, N, device="cuda")
    for v in vs:
        for _ in range(2): ops.gemm(a, b, out=out, variant=v)
    torch.cuda.synchronize()
    samples = {v: [] for v in vs}
    for rnd in range(7):
        for v in (vs if rnd % 2 == 0 else vs[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.gemm(a, b, out=out, variant=v)
            e1.record(); torch.cuda.synchronize()
            samples[v].append(e0.elapsed_time(e1) * 1e3 / 3)
    print(M, N, K, " ".join("v%d %.1fus %.0fTF" % (v, sorted(samples[v])[3], 2.0 * M * N * K / sorted(samples[v])[3] / 1e6) for v in vs))
