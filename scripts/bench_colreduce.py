"""GRN reductions of the ConvNeXt stages (bs 64): energies from the pre-activation (RED_GELUSQ) and the backward's two-plane
dot / sum with gelu recomputed (RED_DOTGELU_SUM), plus a plain column sum.  DKD_HIP_LIB selects another build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
B = 64
for (hw, C) in [(3136, 384), (784, 768), (196, 1536), (49, 3072)]:
    h = torch.randn(B * hw, C, device="cuda"); dy = torch.randn(B * hw, C, device="cuda")
    fns = {"gelusq": lambda: ops.colreduce(h, ops.RED_GELUSQ, groups=B),
           "dotgelu_sum": lambda: ops.colreduce(dy, ops.RED_DOTGELU_SUM, y=h, groups=B),
           "sum": lambda: ops.colreduce(h, ops.RED_SUM)}
    out = []
    for k, f in fns.items():
        for _ in range(5): f()
        torch.cuda.synchronize()
        s = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): f()
            e1.record(); torch.cuda.synchronize()
            s.append(e0.elapsed_time(e1) * 250)
        out.append("%s %.1f us" % (k, sorted(s)[4]))
    print(hw, C, " | ".join(out), flush=True)
for (rows, C) in [(200704, 96), (50176, 192), (16384, 384)]:   # plain column sums: depthwise bias gradients, BatchNorm statistics
    x = torch.randn(rows, C, device="cuda")
    f = lambda: ops.colreduce(x, ops.RED_SUM)
    for _ in range(5): f()
    torch.cuda.synchronize()
    s = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): f()
        e1.record(); torch.cuda.synchronize()
        s.append(e0.elapsed_time(e1) * 250)
    print(rows, C, "sum %.1f us" % sorted(s)[4], flush=True)
