"""Where does a run-to-run difference of the FORWARD pass first appear?  Records a bit-exact copy of the output of every coarse
node of the step (teacher features, student features, every projector / frozen-block / loss node, in host call order) over
several passes from the same state and reports, per pass, the first node whose bits differ from pass 0.
    python tests/diag/diag_trace_forward.py 4 [passes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dinov2_distillation_amd import functional as Fn, ops
from dinov2_distillation_amd.trainer import DistillationStepper
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {
    "1": dict(teacher="dinov2_vits14", student="convnext_tiny", crop=224, batch=64, heads=(16, 24)),
    "4": dict(teacher="dinov2_vitg14", student="convnext_base", crop=224, batch=32, heads=(16, 24), teacher_precision="bf16"),
    "4s": dict(teacher="dinov2_vits14", student="convnext_tiny", crop=224, batch=64, heads=(16, 24), teacher_precision="bf16"),
}
name = sys.argv[1] if len(sys.argv) > 1 else "4"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
c = CONFIGS[name]
st = DistillationStepper(teacher=c["teacher"], student=c["student"], crop=c["crop"], heads=c["heads"], device="cuda:0",
                         teacher_precision=c.get("teacher_precision"))
x = torch.randn(c["batch"], 3, c["crop"], c["crop"], generator=torch.Generator().manual_seed(1234)).cuda()
trace = []


def tap(label, out):
    outs = out if isinstance(out, (tuple, list)) else (out,)
    trace.append((label, [o.detach().clone() for o in outs if torch.is_tensor(o)], torch.cuda.current_stream().cuda_stream))
    return out


def wrap_fn(cls, label):
    inner = cls.apply
    counter = [0]

    def apply(*a, **k):
        counter[0] += 1
        return tap(f"{label}#{counter[0]}", inner(*a, **k))
    cls.apply = staticmethod(apply)
    return counter


counters = [wrap_fn(Fn.AttentionProjectorFn, "projector"), wrap_fn(Fn.ViTBlockFrozenFn, "frozen_block"), wrap_fn(Fn.KDLossFn, "kd_loss")]
t_fwd, s_fwd = st.teacher.forward, st.student.forward
st.teacher.forward = lambda b: {k: tap("teacher:" + k, v) for k, v in t_fwd(b).items()}
st.student.forward = lambda b: {k: tap("student:" + k, v) for k, v in s_fwd(b).items()}
runs = []
for p in range(passes):
    trace.clear()
    for cnt in counters:
        cnt[0] = 0
    st.optimizer.zero_grad()
    loss = st.module.training_step(x, 0)
    torch.cuda.synchronize()
    runs.append((float(loss), list(trace)))
    if os.environ.get("DKD_TRACE_BACKWARD"):
        loss.backward(); ops.wgrad_join(); torch.cuda.synchronize()
    del loss
print(f"config {name}: {len(runs[0][1])} traced nodes per pass; losses {[r[0] for r in runs]}")
streams = sorted({s for _, _, s in runs[0][1]})
for p in range(1, passes):
    first = None
    ndiff = 0
    for (la, ta, sa), (lb, tb, sb) in zip(runs[0][1], runs[p][1]):
        assert la == lb
        if not all(torch.equal(u, v) for u, v in zip(ta, tb)):
            ndiff += 1
            if first is None:
                d = max(((u - v).abs().max() / (v.abs().max() + 1e-30)).item() for u, v in zip(ta, tb))
                nel = sum(int((u != v).sum()) for u, v in zip(ta, tb))
                first = f"{la} (stream #{streams.index(sa)}), {nel} elements differ, max rel {d:.2e}"
    print(f"pass {p} vs 0: {ndiff} nodes differ; first: {first}")
