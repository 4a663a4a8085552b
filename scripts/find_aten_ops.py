"""Which ATen kernels still run inside a step, from where? (torch.profiler, shapes + python stack of every aten:: op
that launches a GPU kernel)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from dinov2_distillation_amd.trainer import DistillationStepper
st = DistillationStepper(device="cuda:0")
x = torch.randn(64, 3, 224, 224).cuda()
for _ in range(3): st.step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    st.step(x)
    torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::copy_", "aten::div_", "aten::zero_", "aten::fill_", "aten::sum", "aten::cat", "aten::div", "aten::sub"):
        stack = [s for s in (e.stack or []) if "dinov2_distillation_amd" in s or "bench" in s or "autograd" in s][:2]
        agg[(e.name, str(e.input_shapes)[:60], " <- ".join(s.split("/")[-1] for s in stack)[:150])] += 1
for (k, n) in agg.most_common(40):
    print(n, *k, sep=" | ")
