"""bf16 GEMM yardstick: ours (csrc/gemm_bf16.hip) beside the vendor library (torch.matmul on bf16 tensors) on the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops


def t(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[2]


for (M, N, K) in [(16448, 1152, 384), (16448, 1536, 384), (16448, 384, 1536), (200704, 384, 96), (200704, 96, 384), (50176, 768, 192),
                  (50176, 192, 768), (12544, 1536, 384), (12544, 384, 1536), (3136, 3072, 768), (3136, 768, 3072), (16384, 384, 384),
                  (16384, 1536, 384), (16384, 384, 1536)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    u = t(lambda: ops.gemm_bf16(a, w, out=out))
    v = t(lambda: torch.matmul(a, w.t(), out=out))
    by = (M * K + N * K + M * N) * 2
    print(f"{M:7d} x {N:5d} x {K:5d}: ours {u:6.1f} us {fl / u * 1e-6:5.0f} TF/s | vendor {v:6.1f} us {fl / v * 1e-6:5.0f} TF/s | "
          f"HBM floor {by / 6.0e6:5.1f} us", flush=True)
