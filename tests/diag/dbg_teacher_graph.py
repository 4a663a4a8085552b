"""Diagnostic: per-step wall time of a step with the autocast teacher while its forward goes eager -> captured -> replayed
(models/dinov2.py, VisionTransformer._forward_tokens_autocast_graph): steps 0-1 eager, step 2 captures, later steps replay.
python3 tests/diag/dbg_teacher_graph.py [config3]   (DKD_TEACHER_GRAPH=0 for the eager A/B)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinov2_distillation_amd.trainer import DistillationStepper
if len(sys.argv) > 1 and sys.argv[1] == "config3":
    st = DistillationStepper(device="cuda:0", precision="bf16_autocast", teacher="dinov2_vitl14", student="swin_tiny", crop=518, heads=(16, 32))
    x = torch.randn(16, 3, 518, 518).cuda()
elif len(sys.argv) > 1 and sys.argv[1] == "config2":
    st = DistillationStepper(device="cuda:0", precision="bf16_autocast", teacher="dinov2_vitb14", student="resnet_50")
    x = torch.randn(128, 3, 224, 224).cuda()
else:
    st = DistillationStepper(device="cuda:0", teacher_precision="bf16_autocast")
    x = torch.randn(64, 3, 224, 224).cuda()
vt = st.teacher.model
for i in range(int(os.environ.get("STEPS", "14"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.step(x)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(i, f"{dt*1e3:.1f} ms", [(e["calls"], "graph" in e) for e in vt._tgraph.values()],
          f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
