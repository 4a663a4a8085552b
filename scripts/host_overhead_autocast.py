"""Is the mixed-precision step host-bound?  Enqueue time vs wall time of the eager step, and hipGraph replay of the same step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd.trainer import DistillationStepper
x = torch.randn(64, 3, 224, 224).cuda()
for prec in (None, "bf16_autocast"):
    st = DistillationStepper(device="cuda:0", precision=prec)
    for _ in range(3): st.step(x)
    torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    for _ in range(K): st.step(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"precision={prec}: host enqueue {1e3*(t1-t0)/K:.1f} ms/step, wall {1e3*(t2-t0)/K:.1f} ms/step", flush=True)
    try:
        st.capture(x)
        for _ in range(3): st.step_graph(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): st.step_graph(x)
        torch.cuda.synchronize()
        print(f"precision={prec}: hipGraph replay {1e3*(time.perf_counter()-t0)/K:.1f} ms/step", flush=True)
    except Exception as e:
        print("graph capture failed:", type(e).__name__, str(e)[:300], flush=True)
    del st
