"""Localise run-to-run differences of the step at a BASELINE config's full size: repeats the teacher forward, the student
forward, the whole training_step and its backward from the same state and reports which tensors are not bit-identical.

    python tests/diag/diag_determinism.py 3|4|2|1 [passes]
    DKD_TEACHER_STREAM=0 / DKD_CHAIN_STREAMS=0 / DKD_ASYNC_WGRAD=0   switch the side streams off one by one
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from dinov2_distillation_amd import functional as Fn, ops
from dinov2_distillation_amd.trainer import DistillationStepper

CONFIGS = {
    "1": dict(teacher="dinov2_vits14", student="convnext_tiny", crop=224, batch=64, heads=(16, 24)),
    "2": dict(teacher="dinov2_vitb14", student="resnet_50", crop=224, batch=128, heads=(16, 24)),
    "3": dict(teacher="dinov2_vitl14", student="swin_tiny", crop=518, batch=16, heads=(16, 32)),
    "4": dict(teacher="dinov2_vitg14", student="convnext_base", crop=224, batch=32, heads=(16, 24), teacher_precision="bf16"),
    "4f32": dict(teacher="dinov2_vitg14", student="convnext_base", crop=224, batch=32, heads=(16, 24)),
}
name = sys.argv[1] if len(sys.argv) > 1 else "3"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = CONFIGS[name]
print(f"config {name}: {c}; env " + " ".join(f"{k}={os.environ[k]}" for k in ("DKD_TEACHER_STREAM", "DKD_CHAIN_STREAMS", "DKD_ASYNC_WGRAD")
                                              if k in os.environ), flush=True)
st = DistillationStepper(teacher=c["teacher"], student=c["student"], crop=c["crop"], heads=c["heads"], device="cuda:0",
                         teacher_precision=c.get("teacher_precision"))
x = torch.randn(c["batch"], 3, c["crop"], c["crop"], generator=torch.Generator().manual_seed(1234)).cuda()


def same(a, b):
    return torch.equal(a, b)


with torch.no_grad():
    st.module.train()
    t = [st.teacher(x)["feature_map"].clone() for _ in range(passes)]
    print("teacher forward bit-identical:", [same(t[0], ti) for ti in t[1:]], flush=True)
    s = []
    for _ in range(passes):
        Fn._DROP_CALLS[0] = 0
        s.append({k: v.clone() for k, v in st.student(x).items()})
    print("student forward bit-identical:", [{k: same(s[0][k], si[k]) for k in si} for si in s[1:]], flush=True)
POISON = os.environ.get("DKD_DIAG_POISON")  # "nan" / "big": fill the allocator's free memory with NaNs / 1e30 before every pass


def poison():
    """an uninitialised read shows up as NaN (or a wildly different loss): every stream's pool gets a freed block of poison"""
    from dinov2_distillation_amd.train import distillation_module as dm
    val = float("nan") if POISON == "nan" else 1e30
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    streams = [torch.cuda.current_stream()] + list(dm._TEACHER_STREAMS.values()) + list(dm._CHAIN_STREAMS.values()) + \
        list(ops._side_streams.values())
    for s_ in streams:
        with torch.cuda.stream(s_):
            blocks = [torch.full((int(float(os.environ.get("DKD_DIAG_POISON_GB", "6")) * 2 ** 28),), val, device="cuda") for _ in range(1)]
            small = [torch.full((n,), val, device="cuda") for n in (1 << 8, 1 << 12, 1 << 16, 1 << 18) for _ in range(64)]
            del blocks, small
    torch.cuda.synchronize()


runs = []
for i in range(passes):
    if POISON:
        poison()
    Fn._DROP_CALLS[0] = 0
    st.optimizer.zero_grad()
    loss = st.module.training_step(x, 0)
    logged = {k: v.detach().clone() for k, v in st.module.logged.items()}
    loss.backward()
    ops.wgrad_join()
    torch.cuda.synchronize()
    runs.append((loss.detach().clone(), logged, st.optimizer.flat_g.clone()))
if POISON:
    for i, (l, lg, g) in enumerate(runs):
        print(f"pass {i} (poisoned free memory): loss {float(l)!r}; non-finite logged: {[k for k, v in lg.items() if not torch.isfinite(v)]}; "
              f"non-finite gradient entries {int((~torch.isfinite(g)).sum())}")
for i in range(1, passes):
    l0, lg0, g0 = runs[0]
    l, lg, g = runs[i]
    bad_scalars = [k for k in lg if not same(lg[k], lg0[k])]
    print(f"pass {i} vs 0: loss equal {same(l, l0)} ({float(l0):.7f} / {float(l):.7f}); differing logged scalars: {bad_scalars}")
    if not same(g, g0):
        names = {id(p): n for n, p in st.module.named_parameters()}
        bad = []
        for p, off, n in st.optimizer.layout:
            if not same(g[off:off + n], g0[off:off + n]):
                d = (g[off:off + n] - g0[off:off + n]).abs().max().item() / (g0[off:off + n].abs().max().item() + 1e-30)
                bad.append((names.get(id(p), "?"), d))
        print(f"   gradients differ in {len(bad)} of {len(st.optimizer.layout)} tensors; first: {bad[:12]}")
    else:
        print("   gradients bit-identical")
