"""Eager vs hipGraph replay of the whole step: where the host enqueue time matters (small batches) and where it does not."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd.trainer import DistillationStepper
PREC = sys.argv[1] if len(sys.argv) > 1 else None   # e.g. bf16_autocast
for student, bs in (("stdc_2", 2), ("convnext_tiny", 2), ("convnext_tiny", 8), ("convnext_tiny", 64)):
    st = DistillationStepper(student=student, device="cuda:0", precision=PREC)
    x = torch.randn(bs, 3, 224, 224).cuda()
    for _ in range(3): st.step(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): st.step(x)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10
    st.capture(x)
    for _ in range(3): st.step_graph(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): st.step_graph(x)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
    print(f"{PREC or 'fp32'} {student} bs={bs}: eager {te*1e3:.2f} ms/step ({bs/te:.0f} img/s)   graph {tg*1e3:.2f} ms/step ({bs/tg:.0f} img/s)", flush=True)
    del st
    torch.cuda.empty_cache()
