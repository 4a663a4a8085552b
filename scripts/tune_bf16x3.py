import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
shapes = [(16384, 1536, 384), (16384, 384, 1536), (16448, 1152, 384), (16384, 384, 384), (200704, 384, 96), (200704, 96, 384),
          (50176, 768, 192), (3136, 3072, 768), (8192, 8192, 4096), (771, 388, 100)]
for (M, N, K) in shapes:
    a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    bias = torch.randn(N, device="cuda")
    ref64 = (a.double() @ b.double().t() + bias.double())
    out = torch.empty(M, N, device="cuda")
    res = []
    for prec, var in ((0, 0), (1, 0), (1, 21), (1, 22), (1, 23)):
        o = ops.gemm(a, b, bias=bias, out=out, precision=prec, variant=var)
        err = ((o.double() - ref64).abs().max() / ref64.abs().max()).item()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.gemm(a, b, bias=bias, out=out, precision=prec, variant=var)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 3)
        res.append("p%d/v%-2d %6.1f TF err %.1e" % (prec, var, 2.0 * M * N * K / best / 1e9, err))
    print("M=%6d N=%5d K=%5d | " % (M, N, K) + " | ".join(res))
