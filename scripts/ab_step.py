"""A/B a module-level switch on the full config-[1] step inside ONE process (boxes differ by a few %, runs drift):
python scripts/ab_step.py functional.CONVNEXT_KEEP_GELU_OUT   -> interleaved medians with the switch False / True"""
import os, sys, time, importlib, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd.trainer import DistillationStepper
modname, attr = sys.argv[1].rsplit(".", 1)
mod = importlib.import_module("dinov2_distillation_amd." + modname)
st = DistillationStepper(device="cuda:0")
x = torch.randn(64, 3, 224, 224).cuda()
for _ in range(3): st.step(x)
res = {False: [], True: []}
for rnd in range(5):
    for val in (False, True):
        setattr(mod, attr, val)
        st.step(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8): st.step(x)
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t0) / 8 * 1e3)
for val in (False, True):
    print(f"{attr}={val}: median {statistics.median(res[val]):.2f} ms/step  ({' '.join('%.2f' % t for t in res[val])})")
