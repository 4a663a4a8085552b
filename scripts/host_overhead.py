"""How far ahead of the GPU is the host? enqueue time of K steps (no sync) vs their GPU completion time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd.trainer import DistillationStepper
st = DistillationStepper(device="cuda:0")
x = torch.randn(64, 3, 224, 224).cuda()
for _ in range(3): st.step(x)
torch.cuda.synchronize()
for K in (1, 5, 10):
    t0 = time.perf_counter()
    for _ in range(K): st.step(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: host enqueue {1e3*(t1-t0)/K:.1f} ms/step, wall {1e3*(t2-t0)/K:.1f} ms/step")
