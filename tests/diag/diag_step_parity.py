"""Diagnostic (GPU box): config-2 step at B=2 - ours vs fp32 CPU oracle vs fp64 CPU oracle, per-parameter errors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinov2_distillation_amd.config import make_config, set_student_dims
from dinov2_distillation_amd.models import DINOv2ViT, ModelWrapper
from dinov2_distillation_amd.train import DistillationModule
from oracle import convnext_oracle as cno, step_oracle as so, vit_oracle as vo

scale_w = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
torch.manual_seed(1)
cfg = make_config()
student = ModelWrapper("convnext_tiny", cfg.teacher.n_patches, cfg.student.student_keys)
set_student_dims(cfg, student)
with torch.no_grad():
    for n_, p_ in student.named_parameters():
        if "grn" in n_:
            p_.add_(0.1 * torch.randn_like(p_))
        elif p_.dim() > 1:
            p_.mul_(scale_w)
teacher = DINOv2ViT("dinov2_vits14", seed=0)
mod = DistillationModule(student, teacher, cfg)
x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1234))


def oracle(dtype):
    sp = {k[len("model.model."):]: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in mod.student.state_dict().items()}
    lp = {k: (v.detach().cpu().to(dtype) if v.dtype.is_floating_point else v.detach().cpu().clone()) for k, v in mod.losses.state_dict().items()}
    for k, v in lp.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    tp = {k: v.detach().cpu().to(dtype) for k, v in mod.teacher.model.state_dict().items()}
    tcfg = vo.VIT_CONFIGS[cfg.teacher.model_name]
    with torch.no_grad():
        feat_t = vo.teacher_feature_map(tp, x.to(dtype), tcfg)
    feats = cno.student_features(sp, x.to(dtype), mod.student.model.model.depths, cfg.teacher.n_patches, cfg.student.student_keys)
    cfgs, w = {}, {}
    for spec in cfg.loss.losses:
        kw = spec.kwargs
        cfgs[kw.name] = dict(alpha=kw.alpha, hw=tuple(kw.pos_hw), num_heads=kw.num_heads, softmax_scale=kw.softmax_scale)
        w[kw.name] = spec.weight
    out = so.compute_losses(lp, cfgs, w, feats, feat_t, tp, tcfg["depth"], tcfg["num_heads"])
    out["loss"].backward()
    g = {"student." + k: v.grad for k, v in sp.items()}
    g.update({"losses." + k: v.grad for k, v in lp.items() if v.dtype.is_floating_point and v.requires_grad})
    return out, g

o32, g32 = oracle(torch.float32)
o64, g64 = oracle(torch.float64)
mod.cuda().train()
loss = mod.training_step(x.cuda(), 0)
loss.backward()
ours = {"student." + k[len("model.model."):]: p.grad.cpu() for k, p in mod.student.named_parameters()}
ours.update({"losses." + k: p.grad.cpu() for k, p in mod.losses.named_parameters()})
print("loss ours %.8f  o32 %.8f  o64 %.8f" % (loss.item(), o32["loss"].item(), o64["loss"].item()))
rows = []
for k, r64 in g64.items():
    r64 = r64.double()
    sc = r64.abs().max().item() + 1e-300
    e_ours = (ours[k].double() - r64).abs().max().item() / sc
    e_o32 = (g32[k].double() - r64).abs().max().item() / sc
    l2_ours = ((ours[k].double() - r64).norm() / (r64.norm() + 1e-300)).item()
    l2_o32 = ((g32[k].double() - r64).norm() / (r64.norm() + 1e-300)).item()
    rows.append((e_ours, e_o32, l2_ours, l2_o32, k))
rows.sort(reverse=True)
print("worst 25 by max-norm rel err vs fp64:  ours | cpu-fp32 | L2 ours | L2 cpu-fp32")
for r in rows[:25]:
    print("%.2e %.2e %.2e %.2e %s" % r)
import statistics
print("median ours %.2e cpu32 %.2e" % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows)))
