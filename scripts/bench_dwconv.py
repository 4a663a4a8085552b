"""dwconv7 forward / dgrad / wgrad at the four ConvNeXt-T stage shapes (bs 64), with the HBM-roofline time beside it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "wgrad16": 0.0}
for (H, C, nblk) in [(56, 96, 3), (28, 192, 3), (14, 384, 9), (7, 768, 3)]:
    B = 64
    x = torch.randn(B, H, H, C, device="cuda")
    d = torch.randn(B, H, H, C, device="cuda")
    wt = torch.randn(49, C, device="cuda")
    bias = torch.randn(C, device="cuda")
    x16, d16 = x.bfloat16(), d.bfloat16()
    fns = {"fwd": lambda: ops.dwconv7(x, wt, bias), "dgrad": lambda: ops.dwconv7(d, wt, None, add=x, flip=True),
           "wgrad": lambda: ops.dwconv7_wgrad(x, d), "wgrad16": lambda: ops.dwconv7_wgrad16(x16, d16)}
    line = "H=%2d C=%3d:" % (H, C)
    for name, f in fns.items():
        f()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        us = sorted(ts)[2]
        nbytes = x.numel() * (2 if name == "wgrad16" else 4) * (2 if name != "dgrad" else 3)
        line += "  %s %6.1f us (%4.2f TB/s)" % (name, us, nbytes / us / 1e6)
        tot[name] += nblk * us
    print(line)
print("per step (us):", {k: round(v) for k, v in tot.items()})
