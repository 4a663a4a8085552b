"""Soak of the step under the autocast policy: two steppers built alike run 150 steps each on the same batch (default stream
schedule: teacher / frequency chain / weight gradients on side streams); their loss curves must be finite, falling, and IDENTICAL
step by step (bit reproducibility of every kernel under co-scheduling), allocated memory must not grow."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd.trainer import DistillationStepper

cfgs = {"convnext_tiny": dict(teacher="dinov2_vits14", student="convnext_tiny", bs=64, crop=224, heads=(16, 24)),
        "swin_tiny": dict(teacher="dinov2_vits14", student="swin_tiny", bs=16, crop=224, heads=(16, 24)),
        "resnet_50": dict(teacher="dinov2_vitb14", student="resnet_50", bs=32, crop=224, heads=(16, 24))}
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else list(cfgs)):
    c = dict(cfgs[name]); bs = c.pop("bs")
    x = torch.randn(bs, 3, c["crop"], c["crop"], generator=torch.Generator().manual_seed(1)).cuda()
    curves, mems = [], []
    for run in range(2):
        from dinov2_distillation_amd import functional as Fn
        Fn._DROP_CALLS[0] = 0
        st = DistillationStepper(device="cuda:0", lr=1e-4, precision="bf16_autocast", **c)
        ls = []
        for i in range(steps):
            ls.append(st.step(x))
            if i == 10 or i == steps - 1:
                mems.append(torch.cuda.memory_allocated() / 2 ** 20)
        curves.append([float(v.detach()) for v in ls])
        del st
        torch.cuda.empty_cache()
    same = curves[0] == curves[1]
    first_diff = next((i for i, (a, b) in enumerate(zip(*curves)) if a != b), None)
    print(f"{name}: loss {curves[0][0]:.3f} -> {curves[0][-1]:.3f}; identical curves: {same} (first difference at step {first_diff}); "
          f"allocated MiB {mems[0]:.0f} -> {mems[1]:.0f}", flush=True)
    assert all(v == v for v in curves[0]) and curves[0][-1] < curves[0][0] and same and mems[1] <= mems[0] * 1.01
print("SOAK OK")
