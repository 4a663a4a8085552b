"""Teacher attention under the policy: the bf16-MFMA flash kernel against the fp32-MFMA kernel's round_bf16 mode (us per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops


def timed(f, n=20):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]


for (B, N, H) in [(64, 257, 6), (128, 257, 12), (16, 1370, 16), (32, 257, 24)]:
    hd = 64; D = H * hd
    qkv = (torch.randn(B * N, 3 * D, device="cuda") * 1.5).bfloat16()
    q32 = qkv.float()
    st = (N * 3 * D, 3 * D)
    t16 = timed(lambda: ops.attention16_flash(qkv, B, H, N, hd, 0.125))
    t32 = timed(lambda: ops.attention_forward(q32, q32[:, D:], q32[:, 2 * D:], H, hd, 0.125, B, N, N, st, st, st, round_bf16=True))
    fl = 4.0 * B * H * N * N * hd
    print(f"B {B:3d} N {N:4d} H {H:2d}: bf16 flash {t16:7.1f} us ({fl / t16 / 1e6:6.1f} TF/s)   fp32-MFMA rounded {t32:7.1f} us ({fl / t32 / 1e6:6.1f} TF/s)")
