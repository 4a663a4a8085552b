"""Window-attention kernel timing at the four Swin-T stage shapes of config [3] (518x518, bs 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
B = 16
for name, nW, H in (("stage0", 361, 3), ("stage1", 100, 6), ("stage2", 25, 12), ("stage3", 9, 24)):
    N, hd = 49, 32
    C, Bw = H * hd, B * nW
    qkv = torch.randn(Bw * N, 3 * C, device="cuda")
    bias = torch.randn(N * N, H, device="cuda")
    lab = torch.randint(0, 4, (nW, N), device="cuda", dtype=torch.int32)
    do = torch.randn(Bw * N, C, device="cuda")
    for labels in (None, lab):
        o = ops.winattn_forward(qkv, bias, labels, Bw, nW, H, N, hd, hd ** -0.5)
        ops.winattn_backward(do, qkv, o, bias, labels, Bw, nW, H, N, hd, hd ** -0.5)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(5): o = ops.winattn_forward(qkv, bias, labels, Bw, nW, H, N, hd, hd ** -0.5)
        e[1].record()
        for _ in range(5): ops.winattn_backward(do, qkv, o, bias, labels, Bw, nW, H, N, hd, hd ** -0.5)
        e[2].record(); torch.cuda.synchronize()
        fl = 4.0 * N * N * hd * Bw * H
        tf, tb = e[0].elapsed_time(e[1]) / 5 * 1e3, e[1].elapsed_time(e[2]) / 5 * 1e3
        print(f"{name} Bw={Bw} H={H} mask={labels is not None}: fwd {tf:.0f} us ({fl/tf/1e6:.1f} TF/s)  bwd {tb:.0f} us ({3.5*fl/tb/1e6:.1f} TF/s)")
