"""Timing-only knock-outs of the bf16 ring GEMM (DKD_TUNING build): which part of the k-loop costs what.
    DKD_TUNING=1 python3 -m dinov2_distillation_amd.build --force && python3 scripts/knockout_bf16_ring.py [cfgs...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops

cfgs = [int(c) for c in sys.argv[1:]] or [2, 1, 7]


def t(f, reps=20):
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


KN = [(0, "as built"), (16, "no stores"), (1, "no DMA in loop"), (2, "frags once per tile"), (4, "no MFMA"), (8, "no barrier"),
      (3, "no DMA, frags once"), (5, "no DMA, no MFMA"), (6, "frags once, no MFMA"), (7, "MFMA-free skeleton: barrier + waits only"),
      (17, "no DMA, no stores"), (19, "no DMA, frags once, no stores (MFMA + barrier)"), (23, "skeleton, no stores"), (31, "nothing")]
for (M, N, K) in [(12544, 384, 1536), (3136, 3072, 768), (16448, 1536, 384), (200704, 384, 384)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    print(f"--- {M} x {N} x {K}  (MFMA floor {fl / 2.5e9:.1f} us)")
    for c in cfgs:
        row = []
        for kn, name in KN:
            try:
                u = t(lambda: ops.gemm_bf16(a, b, out=out, variant=10 + c + 100 * kn))
                row.append(f"{name}: {u:.1f}")
            except Exception as e:
                row.append(f"{name}: n/a")
        print(f"cfg {c}: " + " | ".join(row), flush=True)
