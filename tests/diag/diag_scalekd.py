"""Diagnostic (GPU box): ScaleKD at real sizes vs fp64 oracle, per-parameter errors; also pieces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinov2_distillation_amd.losses import ScaleKD
from dinov2_distillation_amd import ops
from oracle import scalekd_oracle as skd

torch.manual_seed(0)
B, Cs, D, P, heads = 2, 768, 384, 16, 24
for selfq in (False, False, True, False):
    m = ScaleKD(name="scalekd_res5", alpha=[0.08, 0.06], student_dims=Cs, teacher_dims=D, query_hw=[P, P], pos_hw=[P, P],
                pos_dims=D, self_query=selfq, softmax_scale=[5.0, 5.0], num_heads=heads)
    x = torch.randn(B, P, P, Cs).permute(0, 3, 1, 2)
    t = torch.randn(B, P, P, D).permute(0, 3, 1, 2)
    qs = qf = None
    if not selfq:
        qs, qf = torch.randn(B, P * P, D), torch.randn(B, P * P, D)
    res = {}
    for dt in (torch.float32, torch.float64):
        p = {"L." + k: (v.detach().clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in m.state_dict().items()}
        for k, v in p.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_(True)
        xx = x.to(dt).clone().requires_grad_(True)
        q1 = None if qs is None else qs.to(dt).clone().requires_grad_(True)
        q2 = None if qf is None else qf.to(dt).clone().requires_grad_(True)
        cfg = dict(alpha=[0.08, 0.06], hw=(P, P), num_heads=heads, softmax_scale=[5.0, 5.0])
        out = skd.scalekd_forward(p, "L.", xx, t.to(dt), cfg, q1, q2)
        out["loss"].backward()
        g = {k[2:]: v.grad for k, v in p.items() if v.dtype.is_floating_point and v.requires_grad}
        g["__x"] = xx.grad
        if q1 is not None:
            g["__qs"], g["__qf"] = q1.grad, q2.grad
        res[dt] = (out, g)
    m.cuda().train()
    xg = x.cuda().requires_grad_(True)
    q1 = None if qs is None else qs.cuda().requires_grad_(True)
    q2 = None if qf is None else qf.cuda().requires_grad_(True)
    out = m(xg, t.cuda(), query_s=q1, query_f=q2)
    out["loss"].backward()
    ours = {k: p.grad.cpu() for k, p in m.named_parameters()}
    ours["__x"] = xg.grad.cpu()
    if q1 is not None:
        ours["__qs"], ours["__qf"] = q1.grad.cpu(), q2.grad.cpu()
    o64, g64 = res[torch.float64]
    o32, g32 = res[torch.float32]
    print("selfq", selfq, {k: (float(out[k]), float(o64[k])) for k in out})
    rows = []
    for k, r in g64.items():
        sc = r.abs().max().item() + 1e-300
        rows.append(((ours[k].double() - r).abs().max().item() / sc, (g32[k].double() - r).abs().max().item() / sc, sc, k))
    rows.sort(reverse=True)
    for r in rows[:14]:
        print("  %.2e %.2e (scale %.2e) %s" % r)
