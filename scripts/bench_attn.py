import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
for (B, H, N, hd, name) in [(64, 6, 257, 64, "teacher vits14"), (64, 6, 256, 64, "A7 block reuse"), (64, 16, 256, 24, "projector res4"), (64, 24, 256, 16, "projector res5"), (16, 16, 1370, 64, "vitl14 @518")]:
    D = H * hd
    qkv = torch.randn(B, N, 3 * D, device="cuda")
    st = (N * 3 * D, 3 * D)
    sc = hd ** -0.5
    f = lambda: ops.attention_forward(qkv, qkv[..., D:], qkv[..., 2 * D:], H, hd, sc, B, N, N, st, st, st)
    o, lse = f()
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(B, N, H, hd).transpose(1, 2) for i in range(3))
    if B * H * N * N < 3e8:
        ref = (torch.softmax(q @ k.transpose(-1, -2) * sc, -1) @ v).transpose(1, 2).reshape(B, N, D)
        err = ((o - ref).abs().max() / ref.abs().max()).item()
    else:
        err = float("nan")
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    fl = 4.0 * N * N * hd * B * H
    print("%-16s B=%d H=%d N=%d hd=%d: %.1f us  %.1f TFLOP/s (%.1f%% of 157.3)  err %.1e" % (name, B, H, N, hd, best * 1e3, fl / best / 1e9, 100 * fl / best / 1e9 / 157.3, err))
