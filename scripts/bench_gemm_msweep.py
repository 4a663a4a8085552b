"""Why do the teacher's M = 64 * 257 GEMMs run 20 % below the M = 64 * 256 ones of the same N, K (gemm_shapes_final.log)?
Sweep M around 16384 for the two dominant (N, K) pairs, automatic configuration and pinned variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops


def t(M, N, K, variant=0, reps=20):
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    f = lambda: ops.gemm(a, b, out=out, variant=variant)
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


for (N, K) in [(384, 1536), (1536, 384), (1152, 384), (384, 384)]:
    for M in (15360, 16320, 16384, 16448, 16512, 16640, 17408, 20480):
        row = []
        for v in (0, 44, 101):
            try:
                us = t(M, N, K, v)
                row.append(f"v{v}: {us:7.1f} us {2.0 * M * N * K / us * 1e-6:6.1f} TF/s")
            except Exception as e:  # noqa: BLE001
                row.append(f"v{v}: {type(e).__name__}")
        print(f"M={M:6d} ({M / 64:6.1f} row tiles, {M // 64 * (N // 64)} tiles) N={N} K={K}: " + " | ".join(row), flush=True)
