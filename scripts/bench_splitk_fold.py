"""Weight-gradient GEMMs with many split-K slabs: whole call (GEMM + fold) per shape.  Run twice, with and without
DKD_SPLITK_SERIAL_FOLD=1 (the one-thread-per-quad fold), on the same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops, _lib
for (rows, n_out, k_in) in [(200704, 96, 48), (200704, 96, 384), (200704, 384, 96), (50176, 192, 768), (50176, 768, 192), (16384, 384, 384),
                            (12544, 384, 1536), (16384, 384, 1536), (3136, 768, 3072)]:
    dy, x = torch.randn(rows, n_out).cuda(), torch.randn(rows, k_in).cuda()
    for _ in range(3): ops.linear_wgrad(dy, x, True)
    torch.cuda.synchronize()
    s = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.linear_wgrad(dy, x, True)
        e1.record(); torch.cuda.synchronize()
        s.append(e0.elapsed_time(e1) * 1e3 / 5)
    print(f"dW[{n_out},{k_in}] K={rows}: splits {_lib.load().dkd_gemm_f32_splits(n_out, k_in, rows)}  {sorted(s)[3]:.1f} us", flush=True)
