"""Diagnostic (GPU box): step through AttentionProjectorFn by hand at real sizes and compare each intermediate with fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dinov2_distillation_amd import ops

torch.manual_seed(0)
B, Cs, D, P, H = 2, 768, 384, 16, 24
N, M, hd = P * P, B * P * P, D // H
scale = hd ** -0.5 * 5.0
g = torch.Generator().manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g)
x2, wc, bc = r(M, Cs), r(D, Cs) * 0.03, r(D) * 0.1
bnw, bnb, pos = 1 + 0.1 * r(D), 0.1 * r(D), 0.02 * r(N, D)
wq, bq, wk, bk, wv, bv, wp, bp = r(D, D) * .05, r(D) * .1, r(D, D) * .05, r(D) * .1, r(D, D) * .05, r(D) * .1, r(D, D) * .05, r(D) * .1
w1, b1, w2, b2 = r(4 * D, D) * .05, r(4 * D) * .1, r(D, 4 * D) * .05, r(D) * .1
n1w, n1b, n2w, n2b = 1 + .1 * r(D), .1 * r(D), 1 + .1 * r(D), .1 * r(D)
qin = r(M, D)
ct = r(M, D)
def cmp(name, a, b):
    b = b.double(); a = a.detach().cpu().double()
    sc = b.abs().max().item() + 1e-300
    print("  %-8s max-rel %.2e  L2-rel %.2e" % (name, (a - b).abs().max().item() / sc, ((a - b).norm() / b.norm()).item()))
dd = lambda t: t.double()
# ---- fp64 reference with autograd
leaves = {k: dd(v).clone().requires_grad_(True) for k, v in dict(x2=x2, wc=wc, bc=bc, bnw=bnw, bnb=bnb, pos=pos, wq=wq, bq=bq, wk=wk, bk=bk, wv=wv, bv=bv, wp=wp, bp=bp, w1=w1, b1=b1, w2=w2, b2=b2, n1w=n1w, n1b=n1b, n2w=n2w, n2b=n2b, qin=qin).items()}
L = leaves
z = L["x2"] @ L["wc"].t() + L["bc"]
mu, var = z.mean(0), z.var(0, unbiased=False)
tok = torch.relu((z - mu) / torch.sqrt(var + 1e-5) * L["bnw"] + L["bnb"]) + L["pos"].repeat(B, 1)
q = L["qin"] @ L["wq"].t() + L["bq"]; k = tok @ L["wk"].t() + L["bk"]; v = tok @ L["wv"].t() + L["bv"]
hh = lambda t: t.reshape(B, N, H, hd).transpose(1, 2)
o = (torch.softmax(hh(q) @ hh(k).transpose(-1, -2) * scale, -1) @ hh(v)).transpose(1, 2).reshape(M, D)
y = o @ L["wp"].t() + L["bp"] + tok
zn = F.layer_norm(y, (D,), L["n1w"], L["n1b"], 1e-5)
f1 = torch.relu(zn @ L["w1"].t() + L["b1"]); f2 = f1 @ L["w2"].t() + L["b2"] + zn
out = F.layer_norm(f2, (D,), L["n2w"], L["n2b"], 1e-5)
inter = dict(z=z, tok=tok, q=q, k=k, v=v, o=o, y=y, zn=zn, f1=f1, f2=f2, out=out)
for t in inter.values(): t.retain_grad()
(out * dd(ct)).sum().backward()
# ---- ours, by hand
c = lambda t: t.cuda().contiguous()
X2, WC, BC, BNW, BNB, POS, WQ, BQ, WK, BK, WV, BV, WP, BP, W1, B1, W2, B2, N1W, N1B, N2W, N2B, QIN, CT = map(c, (x2, wc, bc, bnw, bnb, pos, wq, bq, wk, bk, wv, bv, wp, bp, w1, b1, w2, b2, n1w, n1b, n2w, n2b, qin, ct))
Z = ops.gemm(X2, WC, bias=BC); mean, rstd = ops.bn_stats(Z)
TOK = ops.bn_act_pos_fwd(Z, mean, rstd, BNW, BNB, POS, relu=True)
Q = ops.gemm(QIN, WQ, bias=BQ); K = ops.gemm(TOK, WK, bias=BK); V = ops.gemm(TOK, WV, bias=BV)
st = (N * D, D)
O, lse = ops.attention_forward(Q, K, V, H, hd, scale, B, N, N, st, st, st)
Y = ops.gemm(O.reshape(M, D), WP, bias=BP, residual=TOK)
ZN, m1, r1 = ops.layernorm_fwd(Y, N1W, N1B, 1e-5)
F1 = ops.gemm(ZN, W1, bias=B1, act=ops.ACT_RELU); F2 = ops.gemm(F1, W2, bias=B2, residual=ZN)
OUT, m2, r2 = ops.layernorm_fwd(F2, N2W, N2B, 1e-5)
print("forward:")
for nm, t in dict(z=Z, tok=TOK, q=Q, k=K, v=V, o=O.reshape(M, D), y=Y, zn=ZN, f1=F1, f2=F2, out=OUT).items(): cmp(nm, t, inter[nm])
print("backward:")
DF2, dn2w, dn2b = ops.layernorm_bwd(CT, F2, m2, r2, N2W); cmp("df2", DF2, inter["f2"].grad)
DW2 = ops.linear_wgrad(DF2, F1); cmp("dw2", DW2, L["w2"].grad)
DF1raw = ops.gemm(DF2, W2, b_layout=1)
cmp("df1raw", DF1raw, inter["f2"].grad @ L["w2"].detach())
DF1 = ops.relu_bwd(DF1raw, F1); cmp("df1", DF1, inter["f1"].grad * (inter["f1"] > 0))
DW1 = ops.linear_wgrad(DF1, ZN); cmp("dw1", DW1, L["w1"].grad)
DZN = ops.gemm(DF1, W1, b_layout=1, residual=DF2); cmp("dzn", DZN, inter["zn"].grad)
DY, dn1w, dn1b = ops.layernorm_bwd(DZN, Y, m1, r1, N1W); cmp("dy", DY, inter["y"].grad)
DO = ops.gemm(DY, WP, b_layout=1); cmp("do", DO, inter["o"].grad)
DQ = torch.empty(B, N, D, device="cuda"); DK, DV = torch.empty_like(K), torch.empty_like(V)
ops.attention_backward(DO.reshape(B, N, D), Q, K, V, O, lse, H, hd, scale, B, N, N, st, st, st, DQ, DK, DV, st, st, st)
cmp("dq", DQ.reshape(M, D), inter["q"].grad); cmp("dk", DK, inter["k"].grad); cmp("dv", DV, inter["v"].grad)
DTOK = ops.gemm(DK, WK, b_layout=1, residual=DY); ops.gemm(DV, WV, b_layout=1, out=DTOK, residual=DTOK); cmp("dtok", DTOK, inter["tok"].grad)
DZ, dbnw, dbnb = ops.bn_bwd(DTOK, Z, mean, rstd, BNW, BNB); cmp("dz", DZ, inter["z"].grad)
cmp("dbnw", dbnw, L["bnw"].grad); cmp("dwc", ops.linear_wgrad(DZ, X2), L["wc"].grad)
