"""Where do the launches of one step come from?  Counts, per Python call site inside the package, (i) every C-ABI launcher call
(ops.check) and (ii) every ATen op that touches a CUDA tensor and is not a pure view (TorchDispatchMode), for one eager step.
    python3 scripts/launch_census.py [precision|none] [top]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from dinov2_distillation_amd import _lib, ops
from dinov2_distillation_amd.trainer import DistillationStepper

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16_autocast"
prec = None if prec == "none" else prec
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
PKG = os.sep + "dinov2_distillation_amd" + os.sep
VIEWS = ("view", "reshape", "as_strided", "empty", "detach", "slice", "permute", "expand", "select", "unsqueeze", "squeeze", "alias",
         "transpose", "t.default", "_unsafe_view", "unbind", "split", "lift_fresh", "is_", "size", "stride", "numel", "sym_",
         "_local_scalar_dense", "record_stream", "_to_copy", "narrow", "unfold", "new_empty", "set_", "is_pinned", "resize_")


def site(skip_files=("ops.py", "_lib.py")):
    f = sys._getframe(1)
    out = []
    while f is not None and len(out) < 2:
        fn = f.f_code.co_filename
        if PKG in fn and os.path.basename(fn) not in skip_files:
            out.append(f"{os.path.basename(fn)}:{f.f_lineno}")
        f = f.f_back
    return " <- ".join(out) or "?"


ours, aten = collections.Counter(), collections.Counter()
inner_check = _lib.check
counting = [False]


def check(code, what, detail=None):
    if counting[0]:
        ours[(what.split("[")[0], site())] += 1
    return inner_check(code, what, detail)


ops.check = _lib.check = check
for m in list(sys.modules.values()):
    if m is not None and getattr(m, "__name__", "").startswith("dinov2_distillation_amd") and getattr(m, "check", None) is inner_check:
        m.check = check


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if counting[0] and not any(v in name for v in VIEWS):
            flat = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
            if any(t.is_cuda for t in flat):
                aten[(name.replace("aten.", ""), site(skip_files=()))] += 1
        return func(*args, **(kwargs or {}))


st = DistillationStepper(device="cuda:0", precision=prec)
x = torch.randn(64, 3, 224, 224).cuda()
for _ in range(3):
    st.step(x)
torch.cuda.synchronize()
counting[0] = True
with Census():
    st.step(x)
torch.cuda.synchronize()
counting[0] = False
print(f"precision={prec}: {sum(ours.values())} C-ABI launcher calls, {sum(aten.values())} non-view ATen ops on CUDA tensors (autograd-thread "
      "ops are not seen by the dispatch mode of the forward thread)")
print("--- C-ABI launcher calls by site")
for (k, n) in ours.most_common(top):
    print(f"{n:5d}  {k[0]:34s} {k[1]}")
byop = collections.Counter()
for (k, n) in ours.items():
    byop[k[0]] += n
print("--- C-ABI launcher calls by op")
for k, n in byop.most_common(50):
    print(f"{n:5d}  {k}")
print("--- ATen ops by site")
for (k, n) in aten.most_common(top):
    print(f"{n:5d}  {k[0]:34s} {k[1]}")
