"""Where the N = 257 forward loses time: leftover query row and leftover key measured apart."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
B, H, hd = 64, 6, 64
D = H * hd
for (Nq, Nk) in [(256, 256), (257, 257), (256, 288), (1024, 256)]:
    q = torch.randn(B, Nq, D, device="cuda")
    kv = torch.randn(B, Nk, 2 * D, device="cuda")
    f = lambda: ops.attention_forward(q, kv, kv[..., D:], H, hd, hd ** -0.5, B, Nq, Nk, (Nq * D, D), (Nk * 2 * D, 2 * D), (Nk * 2 * D, 2 * D))
    f()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts.sort()
    fl = 4.0 * Nq * Nk * hd * B * H
    print("Nq=%d Nk=%d: %.1f us  %.1f TFLOP/s" % (Nq, Nk, ts[2] * 1e3, fl / ts[2] / 1e9))
