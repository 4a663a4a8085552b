"""Where does the PResNet-50 far-end gradient drift come from?  (VERDICT r01, weak #4.)

Runs the presnet50 golden case (tests/test_modules_gpu.py::test_presnet_against_reference_golden[50]) with every
ops.colreduce call checked against an fp64 evaluation of the SAME device input, and prints
  * per reduction op: the worst error of the kernel relative to the column's |result| and to sum|terms| (the conditioning),
    next to what a plain torch fp32 sum of the same input gives;
  * the end-to-end statistic of the test: worst (ours - fp64) / scale over all parameter-gradient digests, and the
    reference's own worst, so that two builds (DKD_HIP_LIB=...) can be compared on one box.

    python tests/diag/diag_colreduce_accuracy.py            # the in-tree library
    DKD_HIP_LIB=ab_libs/libdkd_r01.so python tests/diag/diag_colreduce_accuracy.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import param_recipe as pr
from dinov2_distillation_amd import ops
from dinov2_distillation_amd.models.presnet import BLOCKS, D2Presnet
from oracle import presnet_oracle as po

depth = 50
z = np.load(os.path.join(ROOT, "tests", "golden", f"presnet{depth}_small.npz"))
g = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
stats = {}
inner = ops.colreduce


def checked(x, op=ops.RED_SUM, y=None, aux=None, groups=1, scale=1.0, out=None):
    res = inner(x, op, y, aux, groups, scale, out)
    if op not in (ops.RED_SUM, ops.RED_DOT, ops.RED_SQDIFF, ops.RED_SUMSQ):
        return res
    N = x.shape[-1]
    xd = x.reshape(groups, -1, N).double()
    if op == ops.RED_DOT:
        t = xd * y.reshape(groups, -1, N).double()
    elif op == ops.RED_SQDIFF:
        t = (xd - aux.double().reshape(1, 1, N)) ** 2
    elif op == ops.RED_SUMSQ:
        t = xd * xd
    else:
        t = xd
    ref = t.sum(1) * scale
    mag = t.abs().sum(1) * abs(scale)
    t32 = (t.float().sum(1) * scale).double()
    r = res.reshape(groups, N).double()
    e = (r - ref).abs()
    e32 = (t32 - ref).abs()
    s = stats.setdefault(op, dict(calls=0, rel_res=0.0, rel_mag=0.0, t_rel_res=0.0, t_rel_mag=0.0, rows=0))
    s["calls"] += 1
    s["rows"] = max(s["rows"], xd.shape[1])
    s["rel_res"] = max(s["rel_res"], (e.max() / ref.abs().max()).item())
    s["rel_mag"] = max(s["rel_mag"], (e / mag.clamp_min(1e-30)).max().item())
    s["t_rel_res"] = max(s["t_rel_res"], (e32.max() / ref.abs().max()).item())
    s["t_rel_mag"] = max(s["t_rel_mag"], (e32 / mag.clamp_min(1e-30)).max().item())
    return res


ops.colreduce = checked
net = D2Presnet(depth)
state = pr.seeded_state({k: v.shape for k, v in net.state_dict().items()}, int(g["meta/seed"][0]))
net.load_state_dict(state, strict=True)
net.cuda().train()
feats = net(g["in/x"].cuda())
loss = sum((feats[k] * g["in/cotangent_" + k].cuda()).sum() for k in ("res4", "res5"))
loss.backward()
torch.cuda.synchronize()
names = {ops.RED_SUM: "SUM", ops.RED_DOT: "DOT", ops.RED_SQDIFF: "SQDIFF", ops.RED_SUMSQ: "SUMSQ"}
print(f"library: {os.environ.get('DKD_HIP_LIB', 'in-tree')}")
for op, s in stats.items():
    print(f"colreduce {names[op]:7s} {s['calls']:4d} calls, up to {s['rows']} rows/group: worst err / max|result| kernel "
          f"{s['rel_res']:.2e} (torch fp32 sum {s['t_rel_res']:.2e}); worst err / sum|terms| kernel {s['rel_mag']:.2e} "
          f"(torch {s['t_rel_mag']:.2e})")
p64 = {k: (v.double().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else
           (v.double() if v.dtype.is_floating_point else v)) for k, v in pr.seeded_state(
               {k: v.shape for k, v in net.state_dict().items()}, int(g["meta/seed"][0])).items()}
f64 = po.forward(p64, g["in/x"].double(), BLOCKS[depth], bottleneck=True)
sum((f64[k] * g["in/cotangent_" + k].double()).sum() for k in ("res4", "res5")).backward()
sd = dict(net.named_parameters())
rows = []
for k, v in g.items():
    if k.startswith("pgrad_digest/"):
        name = k[13:]
        d = pr.digest(sd[name].grad.cpu()).double()
        t = pr.digest(p64[name].grad).double()
        v = v.double()
        for sl, what in ((slice(0, 1), "norm"), (slice(2, None), "values")):
            scale = t[sl].abs().max().item() + 1e-30
            rows.append(((d[sl] - t[sl]).abs().max().item() / scale, (v[sl] - t[sl]).abs().max().item() / scale, name, what))
ours = sorted(rows, reverse=True)[:5]
print("worst five (ours/scale, reference/scale, tensor):")
for r in ours:
    print(f"   {r[0]:.2e}  {r[1]:.2e}  {r[2]} ({r[3]})")
print(f"reference's own worst: {max(r[1] for r in rows):.2e}; median ours {sorted(r[0] for r in rows)[len(rows) // 2]:.2e}, "
      f"median reference {sorted(r[1] for r in rows)[len(rows) // 2]:.2e}")
bad = [r for r in rows if r[0] > max(3e-3, 4 * r[1])]
print(f"tensors beyond the ORIGINAL bound max(3e-3 scale, 4 e_ref): {len(bad)}")
