"""Where does the mixed-precision ConvNeXt leave the autocast oracle?  Feeds the SAME stage input to the oracle's and the build's
downsample layer / blocks one at a time and prints the distance after each, beside the autocast-vs-fp32 distance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle import convnext_oracle as co
from dinov2_distillation_amd.models.convnext import ConvNeXtV2
from dinov2_distillation_amd import functional16 as Fn16

depths, dims = [2, 2, 3, 2], [96, 192, 384, 768]
p = co.init_convnext_params(depths, dims, seed=7)
g = torch.Generator().manual_seed(7)
for k, v in p.items():
    if "grn" in k: v.copy_(0.5 * torch.randn(v.shape, generator=g))
    elif k.endswith("bias"): v.copy_(0.1 * torch.randn(v.shape, generator=g))
x = torch.randn(4, 3, 128, 128, generator=g)
net = ConvNeXtV2(depths, dims).set_precision("bf16_autocast")
net.load_state_dict(p, strict=True)
net.cuda().train()
rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm()).item()
with torch.no_grad():
    cur = None
    for i in range(4):
        ds = net.downsample_layers[i]
        with torch.autocast("cpu", dtype=torch.bfloat16):
            if i == 0:
                o = F.conv2d(x, p["downsample_layers.0.0.weight"], p["downsample_layers.0.0.bias"], stride=4)
                o_conv = o
                o = co._ln_channels_first_autocast(o, p["downsample_layers.0.1.weight"], p["downsample_layers.0.1.bias"])
            else:
                xin = cur.permute(0, 3, 1, 2).cpu()
                o_ln = co._ln_channels_first_autocast(xin, p[f"downsample_layers.{i}.0.weight"], p[f"downsample_layers.{i}.0.bias"])
                o = F.conv2d(o_ln, p[f"downsample_layers.{i}.1.weight"], p[f"downsample_layers.{i}.1.bias"], stride=2)
        if i == 0:
            m = Fn16.PatchConv16Fn.apply(x.cuda(), ds[0].weight, ds[0].bias, 4, True)
            print(f"stage {i} stem conv: ours vs oracle {rel(m.permute(0, 3, 1, 2), o_conv):.2e}  dtypes {m.dtype} {o_conv.dtype}")
            m = Fn16.LayerNorm16Fn.apply(m, ds[1].weight, ds[1].bias, 1e-6, False)
        else:
            mln = Fn16.LayerNorm16Fn.apply(cur, ds[0].weight, ds[0].bias, 1e-6, True)
            print(f"stage {i} downsample LN: ours(bf16) vs rnd(oracle) {rel(mln.permute(0, 3, 1, 2), o_ln.bfloat16()):.2e}  oracle dtype {o_ln.dtype}")
            m = Fn16.PatchConv16Fn.apply(mln, ds[1].weight, ds[1].bias, 2, False)
        print(f"stage {i} after downsample: ours vs oracle {rel(m.permute(0, 3, 1, 2), o):.2e}  dtypes {m.dtype} {o.dtype}")
        for j, blk in enumerate(net.stages[i]):
            with torch.autocast("cpu", dtype=torch.bfloat16):
                o = co.block_autocast(p, f"stages.{i}.{j}.", m.permute(0, 3, 1, 2).cpu())   # the oracle block on OUR input
            m = blk(m, autocast=True)
            print(f"stage {i} block {j}: ours vs oracle on the same input {rel(m.permute(0, 3, 1, 2), o):.2e} (branch {rel(m.permute(0,3,1,2).float().cpu() - 0, o):.1e})  dtypes {m.dtype} {o.dtype}")
        cur = m
