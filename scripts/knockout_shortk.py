import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dinov2_distillation_amd import ops
vs = [44, 101, 130, 133, 134]
for (M, N, K) in [(200704, 384, 96), (200704, 96, 384), (16384, 384, 384)]:
    a = torch.randn(M, K).cuda(); b = torch.randn(N, K).cuda(); out = torch.empty(M, N, device="cuda"); bias = torch.randn(N).cuda()
    for v in vs:
        for _ in range(2): ops.gemm(a, b, out=out, bias=bias, variant=v)
    torch.cuda.synchronize()
    samples = {v: [] for v in vs}
    for rnd in range(7):
        for v in (vs if rnd % 2 == 0 else vs[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.gemm(a, b, out=out, bias=bias, variant=v)
            e1.record(); torch.cuda.synchronize()
            samples[v].append(e0.elapsed_time(e1) * 1e3 / 3)
    print(M, N, K, " ".join("v%d %.1fus" % (v, sorted(samples[v])[3]) for v in vs))
