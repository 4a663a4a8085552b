"""backward attention (delta + dQ + dK/dV kernels) at the step's shapes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
tot = 0.0
for (B, H, N, hd, cnt, name) in [(64, 6, 256, 64, 4, "A7 block reuse"), (64, 16, 256, 24, 2, "projector res4"), (64, 24, 256, 16, 2, "projector res5")]:
    D = H * hd
    q, k, v, do = (torch.randn(B, N, D, device="cuda") for _ in range(4))
    st = (N * D, D)
    sc = hd ** -0.5 * (5.0 if hd < 64 else 1.0)
    o, lse = ops.attention_forward(q, k, v, H, hd, sc, B, N, N, st, st, st)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    f = lambda: ops.attention_backward(do, q, k, v, o, lse, H, hd, sc, B, N, N, st, st, st, dq, dk, dv, st, st, st)
    f()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    us = sorted(ts)[3]
    fl = 10.0 * N * N * hd * B * H
    tot += cnt * us
    print("%-16s hd=%d: %.1f us  %.1f TFLOP/s" % (name, hd, us, fl / us / 1e6))
print("per step: %.0f us" % tot)
