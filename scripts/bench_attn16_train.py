"""The policy attention with gradients (hd 64): flash form (csrc/attention16_train.hip) against the materialised path, us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops


def timed(f, n=10):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]


for (B, N, H) in [(64, 256, 6), (16, 1369, 16), (128, 256, 12)]:
    hd = 64; D = H * hd
    qkv = (torch.randn(B * N, 3 * D, device="cuda")).bfloat16()
    do = torch.randn(B * N, D, device="cuda").bfloat16()
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    dqkv = torch.empty_like(qkv)
    out = (dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
    res = {}
    for flash in (True, False):
        ops.ATTN16_FLASH = flash
        O, S, P = ops.attention16_forward(q, k, v, B, H, N, N, hd, 0.125, ld=3 * D)
        tf = timed(lambda: ops.attention16_forward(q, k, v, B, H, N, N, hd, 0.125, ld=3 * D))
        tb = timed(lambda: ops.attention16_backward(do, q, k, v, S, P, B, H, N, N, hd, 0.125, ld=3 * D, out=out))
        res[flash] = (tf, tb)
        del O, S, P
    ops.ATTN16_FLASH = True
    print(f"B {B:3d} N {N:4d} H {H:2d}: flash fwd {res[True][0]:8.1f} bwd {res[True][1]:8.1f} us | materialised fwd {res[False][0]:8.1f} bwd {res[False][1]:8.1f} us")
