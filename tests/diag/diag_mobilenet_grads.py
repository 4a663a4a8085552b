"""Per-parameter gradient error of the MI355X MobileNetV2 vs the fp64 oracle, beside the reference's own fp32 error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import param_recipe as pr
from oracle import mobilenet_v2_oracle as mo
from dinov2_distillation_amd.models.mobilenet_v2 import D2MobileNetV2
z = np.load(os.path.join(ROOT, "tests", "golden", "mobilenet_v2_small.npz"))
g = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
net = D2MobileNetV2()
shapes = {k: v.shape for k, v in net.state_dict().items()}
net.load_state_dict(pr.seeded_state(shapes, 77)); net.cuda().train()
feats = net(g["in/x"].cuda())
sum((feats[k] * g["in/cotangent_" + k].cuda()).sum() for k in ("res4", "res5")).backward()
p64 = {k: (v.double().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.double() if v.dtype.is_floating_point else v)) for k, v in pr.seeded_state(shapes, 77).items()}
f64 = mo.forward(p64, g["in/x"].double())
sum((f64[k] * g["in/cotangent_" + k].double()).sum() for k in ("res4", "res5")).backward()
sd = dict(net.named_parameters())
for name in sd:
    if not name.endswith("weight") or "norm" in name: continue
    a, t = sd[name].grad.cpu().double(), p64[name].grad
    v = g["pgrad_digest/" + name].double()
    scale = t.abs().max().item()
    e_full = (a - t).abs().max().item() / scale
    e_ref_norm = abs(v[0].item() - t.norm().item()) / t.norm().item()
    e_our_norm = abs(a.norm().item() - t.norm().item()) / t.norm().item()
    print("%-28s full-err/scale %.2e   norm err ours %.2e ref %.2e" % (name, e_full, e_our_norm, e_ref_norm))
