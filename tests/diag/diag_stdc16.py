"""stage-by-stage comparison of a stride-2 STDC CatBottleneck under the policy with torch.autocast on the oracle (same inputs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from dinov2_distillation_amd import functional16 as Fn16, functional as Fn
from dinov2_distillation_amd.models.stdc import CatBottleneck
from oracle import stdc_oracle as so

rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).norm() / (b.float().cpu().norm() + 1e-30)).item()
r16 = lambda t: t.bfloat16().float()
g = torch.Generator().manual_seed(13)
B, H, cin, cout = 4, 16, 64, 256
blk = CatBottleneck(cin, cout, 4, 2)
for m in blk.modules():
    if isinstance(m, torch.nn.Conv2d):
        torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", generator=g)
p = {k: v.detach().clone() for k, v in blk.state_dict().items()}
for k, v in p.items():
    if k.endswith("bn.weight") or k.endswith("avd_layer.1.weight"):
        v.copy_(1 + 0.2 * torch.randn(v.shape, generator=g))
    elif k.endswith("bn.bias") or k.endswith("avd_layer.1.bias"):
        v.copy_(0.2 * torch.randn(v.shape, generator=g))
blk.load_state_dict(p)
for m in blk.modules():
    m.autocast16 = True
blk.cuda().train()
x = r16(torch.randn(B, cin, H, H, generator=g).relu())
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
nchw = lambda t: t.permute(0, 3, 1, 2)
with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
    q = {k: v.clone() for k, v in p.items()}
    o1 = so.convx(q, "conv_list.0.", x.bfloat16())
    a = F.conv2d(o1, q["avd_layer.0.weight"], None, stride=2, padding=1, groups=o1.shape[1])
    abn = so._bn(q, "avd_layer.1.", a)
    o2 = so.convx(q, "conv_list.1.", abn)
    ap = F.avg_pool2d(o1, 3, 2, 1)
with torch.no_grad():
    m1 = blk.conv_list[0](nhwc(x))
    print("conv_list.0 (1x1)", rel(nchw(m1), o1))
    bn = blk.avd_layer[1]
    C = o1.shape[1]
    wt = Fn16._prepared(blk.avd_layer[0].weight, ("dwtaps16", 3), 1, C, 9, 1, (9, C))
    print("taps", rel(wt, blk.avd_layer[0].weight.reshape(C, 9).t().bfloat16().float()))
    from dinov2_distillation_amd import ops
    d = ops.round_bf16_(ops.dwconv(nhwc(o1.float()), wt, 3, 2))
    print("depthwise on the oracle's input", rel(nchw(d), a))
    mbn = Fn16.DepthwiseBN16Fn.apply(nhwc(o1.float()), blk.avd_layer[0].weight, bn.weight, bn.bias, (bn.running_mean, bn.running_var), 2, True)
    print("depthwise + bn on the oracle's input", rel(nchw(mbn), abn))
    m2 = blk.conv_list[1](nhwc(abn.float()))
    print("conv_list.1 on the oracle's input", rel(nchw(m2), o2))
    mp = Fn16.AvgPool3_16Fn.apply(nhwc(o1.float()), 2)
    print("avgpool on the oracle's input", rel(nchw(mp), ap))
with torch.no_grad():
    with torch.autocast("cpu", dtype=torch.bfloat16):
        q = {k: v.clone() for k, v in p.items()}
        full = so.cat_bottleneck(q, "", x.bfloat16(), 2)
        q = {k: v.clone() for k, v in p.items()}
    q32 = {k: v.clone() for k, v in p.items()}
    full32 = so.cat_bottleneck(q32, "", x, 2)
    ours = nchw(blk(nhwc(x)))
    for name, sl in (("avgpool(out1)", slice(0, 128)), ("out2", slice(128, 192)), ("out3", slice(192, 224)), ("out4", slice(224, 256))):
        print(f"{name}: ours vs autocast {rel(ours[:, sl], full[:, sl]):.2e}; autocast vs fp32 {rel(full[:, sl], full32[:, sl]):.2e}")
