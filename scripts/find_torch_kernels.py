"""Which aten ops still launch kernels inside one steady-state step (they show in rocprof as at::native::* kernels)?
    python scripts/find_torch_kernels.py [precision]
torch.profiler over one step (all threads: the autograd engine's worker included), aten ops that launched something grouped by input shapes."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from dinov2_distillation_amd.trainer import DistillationStepper  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
stepper = DistillationStepper(device="cuda", **({} if prec == "fp32" else {"precision": prec}))
batch = torch.randn(64, 3, 224, 224).cuda()
for _ in range(3):
    stepper.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    stepper.step(batch)
    torch.cuda.synchronize()
VIEWS = ("view", "reshape", "detach", "alias", "t", "transpose", "permute", "slice", "select", "expand", "as_strided", "unsqueeze", "squeeze",
         "empty", "empty_like", "empty_strided", "unbind", "split", "narrow", "_unsafe_view", "result_type", "to", "_to_copy", "contiguous",
         "chunk", "split_with_sizes", "flatten", "unflatten", "numpy_T", "item", "_local_scalar_dense", "is_nonzero", "record_stream",
         "resolve_conj", "resolve_neg", "lift_fresh", "new_empty", "zeros", "ones_like", "zeros_like", "new_zeros", "clone", "cat")
seen = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name[6:] not in VIEWS:
        seen[(ev.name, str(ev.input_shapes)[:120])] += 1
byname = collections.Counter()
for (n, _), c in seen.items():
    byname[n] += c
print(dict(byname))
print(f"{'aten op':28s} {'calls':>5s}  input shapes")
for key, n in seen.most_common(60):
    print(f"{key[0]:28s} {n:5d}  {key[1]}")
