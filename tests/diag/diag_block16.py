"""Stage-by-stage: the frozen block under the policy vs torch.autocast on the published op sequence (same inputs per stage)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from dinov2_distillation_amd import ops
from oracle import vit_oracle as vo
from test_autocast_gpu import _teacher

t = _teacher("bf16_autocast")
p = {k: v.clone() for k, v in t.model.state_dict().items()}
blk = t.model.blocks[9]
t.cuda()
pp = blk.prepared16()
B, N, D, H = 4, 256, 384, 6
hd = D // H
M = B * N
g = torch.Generator().manual_seed(12)
tok = torch.randn(B, N, D, generator=g)
rel = lambda a, b: ((a.float().cpu().reshape(b.shape) - b.float().cpu()).norm() / b.float().cpu().norm()).item()
pre = "blocks.9."
with torch.autocast("cpu", dtype=torch.bfloat16):
    y_o = vo._cuda_fp32_op(F.layer_norm, tok, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-6)
    qkv_o = F.linear(y_o, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"])
    q5 = qkv_o.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = q5[0] * hd ** -0.5, q5[1], q5[2]
    s_o = q @ k.transpose(-2, -1)
    a_o = vo._cuda_fp32_op(torch.softmax, s_o, dim=-1)
    c_o = (a_o @ v).transpose(1, 2).reshape(B, N, D)
    pr_o = F.linear(c_o, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])
    x1_o = tok + p[pre + "ls1.gamma"] * pr_o
    y2_o = vo._cuda_fp32_op(F.layer_norm, x1_o, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-6)
    u_o = F.linear(y2_o, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"])
    ge_o = F.gelu(u_o)
    m_o = F.linear(ge_o, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])
x2 = tok.reshape(M, D).cuda()
h1, m1, r1 = ops.layernorm_fwd_io(x2, pp["n1w"], pp["n1b"], 1e-6, torch.bfloat16)
print("ln1 (rounded):", rel(h1, y_o.bfloat16()))
qkv = ops.gemm_bf16(h1, pp["wqkv"], bias=pp["bqkv"])
print("qkv:", rel(qkv, qkv_o), qkv_o.dtype)
o16, S, P = ops.attention16_forward(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, N, hd, hd ** -0.5, ld=3 * D)
print("S*scale:", rel(S.float() * hd ** -0.5, s_o.reshape(B * H, N, N)), "P:", rel(P, a_o.bfloat16().reshape(B * H, N, N)), "ctx:", rel(o16, c_o))
# the oracle's ctx into OUR proj, and our MLP on the oracle's x1
x1 = ops.gemm_bf16(c_o.reshape(M, D).cuda(), pp["wproj"], epi=ops.EPI_F32_RESID, bias=pp["bproj"], colscale=pp["ls1"], residual=x2)
print("x1 from the oracle's ctx:", rel(x1 - x2, x1_o - tok))
h2, m2, r2 = ops.layernorm_fwd_io(x1_o.reshape(M, D).cuda(), pp["n2w"], pp["n2b"], 1e-6, torch.bfloat16)
u = ops.gemm_bf16(h2, pp["w1"], bias=pp["b1"])
print("fc1:", rel(u, u_o), "gelu:", rel(ops.gelu16(u_o.reshape(M, -1).cuda()), ge_o))
m = ops.gemm_bf16(ge_o.reshape(M, -1).cuda(), pp["w2"], bias=pp["b2"])
print("fc2 from the oracle's gelu:", rel(m, m_o))
