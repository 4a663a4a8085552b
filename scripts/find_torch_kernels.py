"""Which aten ops still launch kernels inside one steady-state step (they show in rocprof as at::native::* kernels / copyBuffer)?
    python scripts/find_torch_kernels.py [precision]      -> aten op, call count, first input shapes, the package frame that issued it
The autograd engine is put on the calling thread (set_multithreading_enabled(False)) so that the dispatch spy sees the backward pass too."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from dinov2_distillation_amd.trainer import DistillationStepper  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
torch.autograd.set_multithreading_enabled(False)
stepper = DistillationStepper(device="cuda", **({} if prec == "fp32" else {"precision": prec}))
batch = torch.randn(64, 3, 224, 224).cuda()
for _ in range(3):
    stepper.step(batch)
torch.cuda.synchronize()

SKIP = ("aten.view", "aten.reshape", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.permute", "aten.slice",
        "aten.select", "aten.expand", "aten.as_strided", "aten.unsqueeze", "aten.squeeze", "aten.empty", "aten.unbind", "aten.split",
        "aten.narrow", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.lift_fresh", "aten.record_stream", "aten._local_scalar_dense",
        "aten.new_empty", "aten.unflatten", "aten.flatten", "aten.chunk", "profiler.")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seen = collections.Counter()
shapes = {}


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "(autograd engine / C++)"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if fr.filename.startswith(ROOT) and "find_torch_kernels" not in fr.filename:
                    where = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
                    break
            key = (name, where)
            seen[key] += 1
            shapes.setdefault(key, [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2])
        return func(*args, **(kwargs or {}))


with Spy():
    stepper.step(batch)
torch.cuda.synchronize()
print(f"{'aten op':34s} {'calls':>5s}  where / first shapes")
for (name, where), n in seen.most_common(70):
    print(f"{name:34s} {n:5d}  {where}  {shapes[(name, where)]}")
print("total:", sum(seen.values()))
