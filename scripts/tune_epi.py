import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
for (M, N, K) in [(16384, 1536, 384), (16384, 384, 1536), (200704, 384, 96), (200704, 96, 384), (16384, 384, 384), (50176, 768, 192)]:
    a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    out = torch.empty(M, N, device="cuda")
    r = []
    for alpha in (1.0, 12345.0):
        ops.gemm(a, b, out=out, alpha=alpha)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.gemm(a, b, out=out, alpha=alpha)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 3)
        r.append(best * 1e3)
    print("M=%6d N=%5d K=%5d full %.1f us  no-epilogue %.1f us  (epilogue share %.0f%%)  mfma-peak-time %.1f us" % (M, N, K, r[0], r[1], 100 * (r[0] - r[1]) / r[0], 2.0*M*N*K/157.3e6))
