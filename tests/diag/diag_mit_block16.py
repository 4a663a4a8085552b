"""diagnostic: where does a MiT block under the policy leave torch.autocast (CUDA fp32 list forced)?  forward intermediates, several sizes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle import mit_oracle as mo
from dinov2_distillation_amd import ops, functional16 as F16
from dinov2_distillation_amd.models.mit import TransformerBlock

rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).norm() / (b.float().cpu().norm() + 1e-30)).item()
for (C, heads, sr, H) in [(64, 2, 4, 16), (64, 2, 4, 8), (256, 8, 1, 4), (256, 8, 1, 16), (160, 5, 2, 8), (128, 4, 1, 8)]:
    g = torch.Generator().manual_seed(C + sr)
    B, N = 4, H * H
    blk = TransformerBlock(C, heads, 4, sr)
    with torch.no_grad():
        for n, v in blk.named_parameters():
            if v.dim() == 1:
                v.copy_((1.0 if "norm" in n and n.endswith("weight") else 0.0) + 0.2 * torch.randn(v.shape, generator=g))
            elif "dwconv" in n:
                v.copy_(0.3 * torch.randn(v.shape, generator=g))
            else:
                v.copy_(torch.randn(v.shape, generator=g) * (v[0].numel() ** -0.5))
    p = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    x = torch.randn(B, N, C, generator=g)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        t = mo._ln(p, "norm1.", x, policy=True)
        a = mo.attention(p, "attn.", t, H, H, heads, sr, True)
        y = x + a
        z = mo._ln(p, "norm2.", y, policy=True)
        h = F.linear(z, p["mlp.fc1.weight"], p["mlp.fc1.bias"])
        Ch = h.shape[-1]
        d = F.conv2d(h.transpose(1, 2).reshape(B, Ch, H, H), p["mlp.dwconv.dwconv.weight"], p["mlp.dwconv.dwconv.bias"], padding=1, groups=Ch).flatten(2).transpose(1, 2)
        ge = F.gelu(d)
        f2 = F.linear(ge, p["mlp.fc2.weight"], p["mlp.fc2.bias"])
    # ours, fed the ORACLE's own intermediates (teacher-forced per op)
    dev = "cuda"
    prm = {k: v.cuda() for k, v in p.items()}
    z16 = ops.layernorm_fwd_io(y.cuda(), prm["norm2.weight"], prm["norm2.bias"], 1e-5, torch.bfloat16)[0]
    e_ln = rel(z16, z.bfloat16())
    h_o = ops.gemm_bf16(z.bfloat16().cuda().view(-1, C), prm["mlp.fc1.weight"].bfloat16(), bias=prm["mlp.fc1.bias"].bfloat16())
    e_fc1 = rel(h_o.view(B, N, Ch), h)
    wpar = torch.nn.Parameter(prm["mlp.dwconv.dwconv.weight"]); bpar = torch.nn.Parameter(prm["mlp.dwconv.dwconv.bias"])
    g_o = F16.DWConvBiasGelu16Fn.apply(h.cuda().view(B, H, H, Ch), wpar, bpar)
    e_dw = rel(g_o.view(B, N, Ch), ge)
    f_o = ops.gemm_bf16(ge.cuda().view(-1, Ch), prm["mlp.fc2.weight"].bfloat16(), bias=prm["mlp.fc2.bias"].bfloat16())
    e_fc2 = rel(f_o.view(B, N, C), f2)
    # the same ops on the CPU in fp32 from the 16-bit values, rounded once (what a 16-bit op is meant to return)
    h_ref = (z.bfloat16().float() @ p["mlp.fc1.weight"].bfloat16().float().t() + p["mlp.fc1.bias"].bfloat16().float()).bfloat16()
    f_ref = (ge.float() @ p["mlp.fc2.weight"].bfloat16().float().t() + p["mlp.fc2.bias"].bfloat16().float()).bfloat16()
    print(f"C{C} h{heads} sr{sr} H{H}: LN {e_ln:.1e}  fc1 ours/cpu16 {e_fc1:.1e} (cpu16 vs rnd(fp32) {rel(h, h_ref):.1e}, ours vs rnd(fp32) {rel(h_o.view(B, N, Ch), h_ref):.1e})  "
          f"dwconv+gelu {e_dw:.1e}  fc2 ours/cpu16 {e_fc2:.1e} (cpu16 vs rnd(fp32) {rel(f2, f_ref):.1e}, ours vs rnd(fp32) {rel(f_o.view(B, N, C), f_ref):.1e})")
