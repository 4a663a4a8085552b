"""Diagnostic: run the same ScaleKD fwd+bwd twice in a fresh process; log a checksum of every ops.* output; report the
first op whose result differs between run 1 and run 2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops, functional
from dinov2_distillation_amd.losses import ScaleKD

log = []
def wrap(name, fn):
    def f(*a, **k):
        r = fn(*a, **k)
        outs = r if isinstance(r, (tuple, list)) else (r,)
        flat = []
        for o in outs:
            if torch.is_tensor(o): flat.append(o)
            elif isinstance(o, (tuple, list)): flat += [t for t in o if torch.is_tensor(t)]
        log.append((name, [ (tuple(t.shape), t.double().sum().item(), t.double().abs().sum().item()) for t in flat]))
        return r
    return f
for n in dir(ops):
    fn = getattr(ops, n)
    if callable(fn) and not n.startswith("_") and n not in ("workspace", "check") and getattr(fn, "__module__", "") == ops.__name__:
        setattr(ops, n, wrap(n, fn))

torch.manual_seed(0)
B, Cs, D, P, heads = 2, 768, 384, 16, 24
m = ScaleKD(name="scalekd_res5", alpha=[0.08, 0.06], student_dims=Cs, teacher_dims=D, query_hw=[P, P], pos_hw=[P, P],
            pos_dims=D, self_query=False, softmax_scale=[5.0, 5.0], num_heads=heads).cuda().train()
x = torch.randn(B, P, P, Cs).permute(0, 3, 1, 2).cuda()
t = torch.randn(B, P, P, D).permute(0, 3, 1, 2).cuda()
qs, qf = torch.randn(B, P * P, D).cuda(), torch.randn(B, P * P, D).cuda()
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
logs, grads = [], []
for it in range(2):
    m.load_state_dict(sd0)
    m.zero_grad(set_to_none=True)
    log.clear()
    xg, q1, q2 = x.clone().requires_grad_(True), qs.clone().requires_grad_(True), qf.clone().requires_grad_(True)
    out = m(xg, t, query_s=q1, query_f=q2)
    out["loss"].backward()
    torch.cuda.synchronize()
    logs.append(list(log))
    grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
print("ops per run:", len(logs[0]), len(logs[1]))
nd = 0
for i, (a, b) in enumerate(zip(logs[0], logs[1])):
    if a != b:
        print("DIFF at op", i, a[0], a[1], "vs", b[1])
        nd += 1
        if nd > 6: break
for k in grads[0]:
    d = (grads[0][k] - grads[1][k]).abs().max().item()
    if d > 0: print("grad differs", k, d, grads[0][k].abs().max().item())
