"""Throughput of the GPU input pipeline (csrc/augment.hip + host planning) on a batch of 64 decoded 500 x 375 RGB images -> 224 x 224:
images/s end to end (host planning + packing + H2D + ~30 launches) and a cProfile of the host side (DKD_AUG_PROFILE=1)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dinov2_distillation_amd.data import DataAugmentationDINOGpu
rng = np.random.default_rng(0)
imgs = [torch.from_numpy(rng.integers(0, 256, size=(375, 500, 3), dtype=np.uint8)) for _ in range(64)]
aug = DataAugmentationDINOGpu()
gen = torch.Generator().manual_seed(0)
for _ in range(3): aug(imgs, gen)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 10
for _ in range(n): out = aug(imgs, gen)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
plans = aug.last_plans
t1 = time.perf_counter()
for _ in range(n): aug.plan([(375, 500)] * 64, gen)
tp = (time.perf_counter() - t1) / n
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(n): aug(imgs, plans=plans)
e1.record(); torch.cuda.synchronize()
print(f"batch of 64: {dt * 1e3:.1f} ms end to end = {64 / dt:.0f} images/s; random planning alone {tp * 1e3:.1f} ms; with fixed plans "
      f"{e0.elapsed_time(e1) / n:.1f} ms per batch between HIP events")
if os.environ.get("DKD_AUG_PROFILE"):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): aug(imgs, gen)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
