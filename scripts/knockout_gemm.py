import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
for (M, N, K) in [(16384, 384, 1536), (16384, 1536, 384), (12544, 384, 1536)]:
    a = torch.randn(M, K).cuda(); b = torch.randn(N, K).cuda(); out = torch.empty(M, N, device="cuda")
    res = {}
    vs = [44, 98, 92, 96]
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
