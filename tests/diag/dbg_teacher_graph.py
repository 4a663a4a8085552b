"""Diagnostic: per-step wall time of the step with the autocast teacher while its forward goes eager -> captured -> replayed
(models/dinov2.py, VisionTransformer._forward_tokens_autocast_graph): steps 0-1 eager, step 2 captures, later steps replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinov2_distillation_amd.trainer import DistillationStepper
st = DistillationStepper(device="cuda:0", teacher_precision="bf16_autocast")
x = torch.randn(64, 3, 224, 224).cuda()
vt = st.teacher.model
for i in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.step(x)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ents = list(vt._tgraph.values())
    print(i, f"{dt*1e3:.1f} ms", [(e["calls"], "graph" in e, sum(e["stamp"])) for e in ents], vt.blocks[0].attn.qkv.weight._version, flush=True)
