"""Oracle (TEST INFRASTRUCTURE): MiT / SegFormer student, CPU fp32 restatement.  PINNED by tests/golden/mit_b0_small.npz
(generated from the reference's own MultiscaleImageTransformer).  Restates /root/reference/models/backbones/mit.py:22-34
(DWConv), :70-78 (Mix-FFN), :128-165 (spatial-reduction attention), :218-221 (block), :262-269 (OverlapPatchEmbed.forward)
and :476-519 (forward_features) functionally; parameter names are the reference's state-dict keys.  All dropout and
drop-path rates are 0 in the wrapper (mit_wrapper.py:76-86)."""
import torch
import torch.nn.functional as F


def _fp32(fn, *args, **kw):
    """an op of CUDA autocast's fp32 list (layer_norm, softmax) when the oracle runs under torch.autocast on the CPU, whose own
    policy leaves those ops in the input dtype: inputs cast up, autocast off inside (as oracle/scalekd_oracle.py, vit_oracle.py)"""
    with torch.autocast("cpu", enabled=False):
        return fn(*[a.float() if torch.is_tensor(a) and a.is_floating_point() else a for a in args], **kw)


def _ln(p, pre, x, eps=1e-5, policy=False):
    if policy:
        return _fp32(F.layer_norm, x, (x.shape[-1],), p[pre + "weight"], p[pre + "bias"], eps)
    return F.layer_norm(x, (x.shape[-1],), p[pre + "weight"], p[pre + "bias"], eps)


def attention(p, pre, x, H, W, heads, sr, policy=False):
    """policy: the caller runs this under torch.autocast("cpu", bfloat16); layer_norm and softmax are then forced to fp32 as the
    CUDA policy the reference trains under does (train.py:263), and the matmul operands are made dense"""
    B, N, C = x.shape
    hd = C // heads
    q = F.linear(x, p[pre + "q.weight"], p[pre + "q.bias"]).reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    if sr > 1:
        x_ = x.permute(0, 2, 1).reshape(B, C, H, W)
        x_ = F.conv2d(x_, p[pre + "sr.weight"], p[pre + "sr.bias"], stride=sr).reshape(B, C, -1).permute(0, 2, 1)
        x = _ln(p, pre + "norm.", x_, policy=policy)
    kv = F.linear(x, p[pre + "kv.weight"], p[pre + "kv.bias"]).reshape(B, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
    if policy:
        logits = (q.contiguous() @ kv[0].transpose(-2, -1).contiguous()) * hd ** -0.5
        o = (_fp32(torch.softmax, logits, dim=-1) @ kv[1].contiguous()).transpose(1, 2).reshape(B, N, C)
    else:
        attn = ((q @ kv[0].transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
        o = (attn @ kv[1]).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, p[pre + "proj.weight"], p[pre + "proj.bias"])


def mix_ffn(p, pre, x, H, W):
    B, N, _ = x.shape
    h = F.linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"])
    C = h.shape[-1]
    h = F.conv2d(h.transpose(1, 2).reshape(B, C, H, W), p[pre + "dwconv.dwconv.weight"], p[pre + "dwconv.dwconv.bias"],
                 padding=1, groups=C).flatten(2).transpose(1, 2)
    return F.linear(F.gelu(h), p[pre + "fc2.weight"], p[pre + "fc2.bias"])


def forward(p, x, depths, heads=(1, 2, 4, 8), sr_ratios=(8, 4, 2, 1), policy=False):
    outs = {}
    B = x.shape[0]
    for i in range(4):
        k, s = (7, 4) if i == 0 else (3, 2)
        x = F.conv2d(x, p[f"patch_embed{i + 1}.proj.weight"], p[f"patch_embed{i + 1}.proj.bias"], stride=s, padding=k // 2)
        H, W = x.shape[2:]
        x = _ln(p, f"patch_embed{i + 1}.norm.", x.flatten(2).transpose(1, 2), policy=policy)
        for j in range(depths[i]):
            pre = f"block{i + 1}.{j}."
            x = x + attention(p, pre + "attn.", _ln(p, pre + "norm1.", x, policy=policy), H, W, heads[i], sr_ratios[i], policy)
            x = x + mix_ffn(p, pre + "mlp.", _ln(p, pre + "norm2.", x, policy=policy), H, W)
        x = _ln(p, f"norm{i + 1}.", x, policy=policy).reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        outs[f"res{i + 2}"] = x
    return outs
