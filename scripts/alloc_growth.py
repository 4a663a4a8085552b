"""does the caching allocator reach a steady state under the un-synchronised step loop?  reserved memory / device allocations every 10 steps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd.trainer import DistillationStepper
cfgs = {"1": dict(teacher="dinov2_vits14", student="convnext_tiny", batch=64), "2": dict(teacher="dinov2_vitb14", student="resnet_50", batch=128)}
c = dict(cfgs[sys.argv[1] if len(sys.argv) > 1 else "1"])
B = c.pop("batch")
for inflight in (3, 2, 0):
    st = DistillationStepper(device="cuda:0", precision="bf16_autocast", max_inflight_steps=inflight, **c)
    x = torch.randn(B, 3, 224, 224).cuda()
    out = []
    t0 = time.perf_counter()
    for i in range(80):
        st.step(x)
        if i % 10 == 9:
            s = torch.cuda.memory_stats()
            t1 = time.perf_counter()   # (host time: lags the GPU by the steps in flight, the same number in every window)
            out.append(f"{i + 1}: {s['reserved_bytes.all.current'] / 2 ** 30:.1f} GiB / {s.get('num_device_alloc', 0)} allocs / {(t1 - t0) / 10 * 1e3:.1f} ms")
            t0 = t1
    torch.cuda.synchronize()
    print(f"max_inflight_steps={inflight}: " + "; ".join(out) + f"; peak allocated {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB", flush=True)
    del st
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
