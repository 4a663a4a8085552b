import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
B, H, hd = 64, 6, 64
D = H * hd
for (Nq, Nk) in [(32, 257), (33, 257), (36, 257), (256, 256), (257, 257)]:
    q, k, v = torch.randn(B, Nq, D, device="cuda"), torch.randn(B, Nk, D, device="cuda"), torch.randn(B, Nk, D, device="cuda")
    f = lambda: ops.attention_forward(q, k, v, H, hd, hd ** -0.5, B, Nq, Nk, (Nq * D, D), (Nk * D, D), (Nk * D, D))
    f()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print("Nq=%d Nk=%d: %.1f us  %.1f TF/s" % (Nq, Nk, best * 1e3, 4.0 * Nq * Nk * hd * B * H / best / 1e9))
