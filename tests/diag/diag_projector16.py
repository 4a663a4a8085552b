"""Stage-by-stage comparison of the mixed-precision projector forward with torch.autocast on the oracle's ops (same inputs per stage)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from dinov2_distillation_amd import ops
from dinov2_distillation_amd.functional16 import _w16
from oracle.scalekd_oracle import _fp32

g = torch.Generator().manual_seed(1)
B, P, Cs, D, H = 4, 16, 768, 384, 24
N, M, hd = P * P, B * P * P, D // H
rel = lambda a, b: ((a.float().cpu().reshape(b.shape) - b.float()).norm() / b.float().norm()).item()
x = torch.randn(B, Cs, P, P, generator=g).bfloat16().float()
wc, bc = torch.randn(D, Cs, 1, 1, generator=g) * Cs ** -0.5, 0.1 * torch.randn(D, generator=g)
bnw, bnb = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
pos = 0.02 * torch.randn(1, D, P, P, generator=g)
with torch.autocast("cpu", dtype=torch.bfloat16):
    z_o = F.conv2d(x, wc, bc)
    y_o = F.batch_norm(z_o, torch.zeros(D), torch.ones(D), bnw, bnb, True, 0.1, 1e-5)
    t_o = torch.relu(y_o) + pos
tok_o = t_o.flatten(2).transpose(1, 2)
xn = x.permute(0, 2, 3, 1).reshape(M, Cs).cuda()
z16 = ops.gemm_bf16(ops.cast_bf16(xn), ops.cast_bf16(wc.reshape(D, Cs).cuda()), bias=_w16(bc.cuda()))
print("conv:", rel(z16.reshape(B, P, P, D).permute(0, 3, 1, 2), z_o), z_o.dtype)
z = ops.cast_f32(z16)
mean, rstd = ops.bn_stats(z, torch.zeros(D).cuda(), torch.ones(D).cuda(), 0.1, 1e-5)
ybn = ops.bn_act_pos_fwd(z, mean, rstd, bnw.cuda(), bnb.cuda(), None, relu=0 | 16)
print("bn:", rel(ybn.reshape(B, P, P, D).permute(0, 3, 1, 2), y_o), y_o.dtype,
      "frac of elements differing:", (ybn.reshape(B, P, P, D).permute(0, 3, 1, 2).cpu() != y_o.float()).float().mean().item())
pos_t = ops.transpose(pos.reshape(D, N).cuda())
tok = ops.bn_act_pos_fwd(z, mean, rstd, bnw.cuda(), bnb.cuda(), pos_t, relu=1 | 16)
print("tok:", rel(tok.reshape(B, N, D), tok_o), tok_o.dtype)
# attention on the oracle's tok
wq, wk, wv = (torch.randn(D, D, generator=g) * D ** -0.5 for _ in range(3))
bq, bk, bv = (0.1 * torch.randn(D, generator=g) for _ in range(3))
qin = torch.randn(N, D, generator=g)
with torch.autocast("cpu", dtype=torch.bfloat16):
    heads = lambda t, n: t.reshape(B, n, H, hd).permute(0, 2, 1, 3)
    q_o = heads(F.linear(qin.unsqueeze(0).expand(B, -1, -1), wq, bq), N)
    k_o = heads(F.linear(tok_o, wk, bk), N)
    v_o = heads(F.linear(tok_o, wv, bv), N)
    lg = (q_o @ k_o.transpose(-1, -2)) * (hd ** -0.5) * 5.0
    w_o = _fp32(torch.softmax, lg, dim=-1)
    a_o = (w_o @ v_o).transpose(1, 2).reshape(B, N, D)
tokd = tok_o.reshape(M, D).cuda()
tok16 = ops.cast_bf16(tokd)
q16 = ops.gemm_bf16(ops.cast_bf16(qin.cuda()), ops.cast_bf16(wq.cuda()), bias=_w16(bq.cuda()))
k16 = ops.gemm_bf16(tok16, ops.cast_bf16(wk.cuda()), bias=_w16(bk.cuda()))
v16 = ops.gemm_bf16(tok16, ops.cast_bf16(wv.cuda()), bias=_w16(bv.cuda()))
print("k:", rel(k16.reshape(B, N, H, hd).permute(0, 2, 1, 3), k_o), "q:", rel(q16.reshape(1, N, H, hd).permute(0, 2, 1, 3).expand(B, H, N, hd), q_o))
o16, S, Pm = ops.attention16_forward(q16, k16, v16, B, H, N, N, hd, hd ** -0.5, 5.0, q_shared=True)
print("logits S:", rel(S.reshape(B, H, N, N), (q_o @ k_o.transpose(-1, -2))), "P:", rel(Pm.reshape(B, H, N, N), w_o.bfloat16()), "attn out:", rel(o16.reshape(B, N, D), a_o), a_o.dtype)
print("cast equal:", torch.equal(tok16.cpu(), tok_o.reshape(M, D).bfloat16()))
k_man = (tok_o.reshape(M, D).bfloat16().double() @ wk.bfloat16().double().t() + bk.bfloat16().double()).float().bfloat16()
print("k ours vs manual:", rel(k16, k_man), " oracle vs manual:", rel(k_o.permute(0, 2, 1, 3).reshape(M, D), k_man))
with torch.autocast("cpu", dtype=torch.bfloat16):
    k_c = F.linear(tok_o.contiguous(), wk, bk)
print("oracle (contiguous input) vs manual:", rel(k_c.reshape(M, D), k_man))
