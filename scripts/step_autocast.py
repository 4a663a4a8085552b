"""A few steps of config [1] with teacher AND student under the autocast policy (profiling target):
    rocprofv3 --kernel-trace --stats -- python3 scripts/step_autocast.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from dinov2_distillation_amd.trainer import DistillationStepper

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
st = DistillationStepper(teacher_precision="bf16_autocast", student_precision=os.environ.get("STUDENT", "bf16_autocast"),
                         projector_precision=os.environ.get("PROJECTOR", "bf16_autocast"))
x = torch.randn(64, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
st.step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    st.step(x)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step")
