import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
for (M, N, K) in [(16384, 1536, 384), (16384, 1536, 1536), (8192, 8192, 4096)]:
    for fill in ("randn", "zeros"):
        a = getattr(torch, fill)(M, K, device="cuda"); b = getattr(torch, fill)(N, K, device="cuda")
        out = torch.empty(M, N, device="cuda")
        for v in (2, 5, 9, 12):
            ops.gemm(a, b, out=out, variant=v)
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3): ops.gemm(a, b, out=out, variant=v)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 3)
            print(M, N, K, fill, "v%d %.1f TF/s" % (v, 2.0 * M * N * K / best / 1e9))
