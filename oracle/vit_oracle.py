"""Oracle (TEST INFRASTRUCTURE): DINOv2 ViT teacher, CPU fp32 restatement.  PARITY UNPINNED.

The reference obtains the teacher with ``torch.hub.load('facebookresearch/dinov2', name)``
(/root/reference/models/backbones/dinov2.py:20); that dependency is unpinned and absent from
/root/reference, and fetching it was denied, so this file restates the *published* DINOv2 ViT
architecture (pre-norm blocks, LayerScale, qkv bias, head_dim 64, GELU MLP x4, LN eps 1e-6, patch 14,
learned 37x37 position table bicubically resized).  What IS anchored on the reference:
  * I/O contract of ``DINOv2ViT.forward``  (models/backbones/dinov2.py:27-46): last-layer tokens after the
    final LayerNorm, CLS dropped, returned as a permuted ``[B,D,P,P]`` *view* of token-major memory.
  * block re-use on student tokens (train/distillation_module.py:169-177): ``blocks[i]`` maps
    ``[B,N,D] -> [B,N,D]`` and autograd flows to the input.

Parameter names follow the public DINOv2 checkpoints (``blocks.{i}.attn.qkv.weight`` ...).
"""
import math

import torch
import torch.nn.functional as F

# teacher table: /root/reference/train.py:103-108 (embed dims); depth/heads from the published models.
VIT_CONFIGS = {
    "dinov2_vits14": dict(embed_dim=384, depth=12, num_heads=6, ffn="mlp"),
    "dinov2_vitb14": dict(embed_dim=768, depth=12, num_heads=12, ffn="mlp"),
    "dinov2_vitl14": dict(embed_dim=1024, depth=24, num_heads=16, ffn="mlp"),
    "dinov2_vitg14": dict(embed_dim=1536, depth=40, num_heads=24, ffn="swiglu"),
}
PATCH = 14
POS_GRID = 37  # 518 // 14: the resolution the published position table is stored at


def init_vit_params(cfg, seed=0, depth=None, dtype=torch.float32):
    """Synthetic teacher weights: trunc-normal(0.02) linears, zero bias,
    LayerScale 1.0, LN (1, 0).  No pretrained weights are reachable (no network)."""
    g = torch.Generator().manual_seed(seed)
    D = cfg["embed_dim"]
    depth = cfg["depth"] if depth is None else depth

    def tn(*shape):
        t = torch.empty(*shape, dtype=dtype)
        torch.nn.init.trunc_normal_(t, std=0.02, generator=g)
        return t

    p = {
        "cls_token": tn(1, 1, D),
        "pos_embed": tn(1, 1 + POS_GRID * POS_GRID, D),
        "patch_embed.proj.weight": tn(D, 3, PATCH, PATCH),
        "patch_embed.proj.bias": torch.zeros(D, dtype=dtype),
        "norm.weight": torch.ones(D, dtype=dtype),
        "norm.bias": torch.zeros(D, dtype=dtype),
    }
    hidden = 4 * D
    for i in range(depth):
        b = f"blocks.{i}."
        p[b + "norm1.weight"] = torch.ones(D, dtype=dtype)
        p[b + "norm1.bias"] = torch.zeros(D, dtype=dtype)
        p[b + "attn.qkv.weight"] = tn(3 * D, D)
        p[b + "attn.qkv.bias"] = torch.zeros(3 * D, dtype=dtype)
        p[b + "attn.proj.weight"] = tn(D, D)
        p[b + "attn.proj.bias"] = torch.zeros(D, dtype=dtype)
        p[b + "ls1.gamma"] = torch.ones(D, dtype=dtype)
        p[b + "norm2.weight"] = torch.ones(D, dtype=dtype)
        p[b + "norm2.bias"] = torch.zeros(D, dtype=dtype)
        p[b + "mlp.fc1.weight"] = tn(hidden, D)
        p[b + "mlp.fc1.bias"] = torch.zeros(hidden, dtype=dtype)
        p[b + "mlp.fc2.weight"] = tn(D, hidden)
        p[b + "mlp.fc2.bias"] = torch.zeros(D, dtype=dtype)
        p[b + "ls2.gamma"] = torch.ones(D, dtype=dtype)
    return p


class _Bf16Linear(torch.autograd.Function):
    """y = round_bf16(x) round_bf16(w)^T + b with fp32 accumulation; dx = round_bf16(dy) round_bf16(w) (the frozen weight
    gets no gradient): the arithmetic of a bf16-operand MFMA in both directions."""

    @staticmethod
    def forward(ctx, x, w, b):
        wr = w.bfloat16().to(w.dtype)
        ctx.save_for_backward(wr)
        return F.linear(x.bfloat16().to(x.dtype), wr, b)

    @staticmethod
    def backward(ctx, dy):
        (wr,) = ctx.saved_tensors
        return dy.bfloat16().to(dy.dtype) @ wr, None, None


def _linear(x, w, b, bf16, scale=None):
    """F.linear; bf16=True restates the opt-in "bf16 teacher on MFMA" arithmetic (BASELINE configs[4]): both operands rounded
    to bfloat16 (round-to-nearest-even), products and sums in fp32.  ``scale`` (LayerScale) is folded into the weight BEFORE
    the rounding, as the product folds it into the frozen projection weights."""
    if scale is not None:
        w, b = w * scale[:, None], b * scale
    if bf16:
        return _Bf16Linear.apply(x, w, b)
    return F.linear(x, w, b)


def vit_block(p, prefix, x, num_heads, eps=1e-6, bf16=False):
    """One pre-norm block: x += ls1*proj(MHSA(LN(x))); x += ls2*MLP(LN(x)).
    Call site in the reference: train/distillation_module.py:177 (``blocks[i](feat)``)."""
    B, N, D = x.shape
    hd = D // num_heads
    y = F.layer_norm(x, (D,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], eps)
    qkv = _linear(y, p[prefix + "attn.qkv.weight"], p[prefix + "attn.qkv.bias"], bf16)
    qkv = qkv.view(B, N, 3, num_heads, hd)
    q = qkv[:, :, 0].transpose(1, 2)  # [B,h,N,hd]
    k = qkv[:, :, 1].transpose(1, 2)
    v = qkv[:, :, 2].transpose(1, 2)
    logits = (q * hd ** -0.5) @ k.transpose(-1, -2)
    ctx = torch.softmax(logits, dim=-1) @ v
    ctx = ctx.transpose(1, 2).reshape(B, N, D)
    x = x + _linear(ctx, p[prefix + "attn.proj.weight"], p[prefix + "attn.proj.bias"], bf16, scale=p[prefix + "ls1.gamma"])
    y = F.layer_norm(x, (D,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], eps)
    if (prefix + "mlp.w12.weight") in p:  # ViT-g: SwiGLUFFNFused, w3(silu(x1) * x2) with [x1 | x2] = w12(y)
        x1, x2 = _linear(y, p[prefix + "mlp.w12.weight"], p[prefix + "mlp.w12.bias"], bf16).chunk(2, dim=-1)
        m = _linear(F.silu(x1) * x2, p[prefix + "mlp.w3.weight"], p[prefix + "mlp.w3.bias"], bf16, scale=p[prefix + "ls2.gamma"])
    else:
        m = _linear(y, p[prefix + "mlp.fc1.weight"], p[prefix + "mlp.fc1.bias"], bf16)
        m = F.gelu(m)
        m = _linear(m, p[prefix + "mlp.fc2.weight"], p[prefix + "mlp.fc2.bias"], bf16, scale=p[prefix + "ls2.gamma"])
    return x + m


def resize_pos_table(pos_embed, ph, pw, offset=0.1):
    """Bicubic resize of the stored 37x37 patch position table to (ph, pw) (published recipe:
    scale_factor = (P + 0.1) / 37, antialias off).  Identity when the grid already matches."""
    cls_pos, patch_pos = pos_embed[:, :1], pos_embed[:, 1:]
    n = patch_pos.shape[1]
    m = int(math.isqrt(n))
    if ph * pw == n and ph == pw:
        return pos_embed
    D = patch_pos.shape[-1]
    grid = patch_pos.reshape(1, m, m, D).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=((ph + offset) / m, (pw + offset) / m), mode="bicubic", antialias=False)
    assert grid.shape[-2:] == (ph, pw)
    patch_pos = grid.permute(0, 2, 3, 1).reshape(1, ph * pw, D)
    return torch.cat([cls_pos, patch_pos], dim=1)


def vit_tokens(p, x, cfg, depth=None, bf16=False):
    """Patch-embed + CLS + position + all blocks + final LN; returns [B, 1+P*P, D]."""
    B, _, H, W = x.shape
    ph, pw = H // PATCH, W // PATCH
    depth = cfg["depth"] if depth is None else depth
    if bf16:
        t = F.conv2d(x.bfloat16().to(x.dtype), p["patch_embed.proj.weight"].bfloat16().to(x.dtype), p["patch_embed.proj.bias"],
                     stride=PATCH)
    else:
        t = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=PATCH)
    t = t.flatten(2).transpose(1, 2)  # [B, P*P, D]
    t = torch.cat([p["cls_token"].expand(B, -1, -1), t], dim=1)
    t = t + resize_pos_table(p["pos_embed"], ph, pw)
    for i in range(depth):
        t = vit_block(p, f"blocks.{i}.", t, cfg["num_heads"], bf16=bf16)
    return F.layer_norm(t, (t.shape[-1],), p["norm.weight"], p["norm.bias"], 1e-6)


def teacher_feature_map(p, x, cfg, depth=None, bf16=False):
    """``DINOv2ViT.forward`` contract (models/backbones/dinov2.py:32-40): tokens of the last layer
    (``get_intermediate_layers(n=1, return_class_token=True)`` applies the final norm and strips CLS),
    reshaped [B,P,P,D] and permuted to [B,D,P,P] without copying."""
    B, _, H, W = x.shape
    ph, pw = H // PATCH, W // PATCH
    tok = vit_tokens(p, x, cfg, depth, bf16)[:, 1:]
    return tok.reshape(B, ph, pw, tok.shape[-1]).permute(0, 3, 1, 2)


# ---- the reference's default mixed-precision policy (train.py:263, precision=16 == torch autocast) for the teacher, in bf16 ------
def _r16(t):
    """round to bfloat16 (nearest even), keep the container dtype"""
    return t.bfloat16().to(t.dtype)


def _lin16(x16, w, b):
    """nn.Linear under autocast: input, weight AND bias cast to the 16-bit type, fp32 accumulation, 16-bit result"""
    return _r16(F.linear(x16, _r16(w), _r16(b)))


def vit_block_autocast(p, prefix, x, num_heads, eps=1e-6):
    """One block under torch.autocast(bf16)'s op policy: LayerNorm, softmax and the residual adds in fp32; every Linear on
    bf16 operands with a bf16 result; GELU / SiLU evaluated on the bf16 tensor (fp32 inside, rounded); LayerScale an fp32
    multiply AFTER the projection's rounding.  Attention is the EAGER form (what the hub model runs without xformers, and
    what autocast can see): q * hd^-0.5 and q k^T are 16-bit results, softmax is an fp32 op whose output is cast to 16 bit as
    the operand of P V, the context a 16-bit result.  (test_autocast_oracle_follows_torch_autocast_policy: this emulation
    agrees with torch.autocast itself to 1e-4, twenty times closer than either is to fp32.)"""
    B, N, D = x.shape
    hd = D // num_heads
    h = _r16(F.layer_norm(x, (D,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], eps))
    qkv = _lin16(h, p[prefix + "attn.qkv.weight"], p[prefix + "attn.qkv.bias"]).view(B, N, 3, num_heads, hd)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    ctx = _r16(torch.softmax(_r16(_r16(q * hd ** -0.5) @ k.transpose(-1, -2)), dim=-1)) @ v
    ctx = _r16(ctx.transpose(1, 2).reshape(B, N, D))
    x = x + p[prefix + "ls1.gamma"] * _lin16(ctx, p[prefix + "attn.proj.weight"], p[prefix + "attn.proj.bias"])
    h = _r16(F.layer_norm(x, (D,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], eps))
    if (prefix + "mlp.w12.weight") in p:
        x1, x2 = _lin16(h, p[prefix + "mlp.w12.weight"], p[prefix + "mlp.w12.bias"]).chunk(2, dim=-1)
        m = _lin16(_r16(_r16(F.silu(x1)) * x2), p[prefix + "mlp.w3.weight"], p[prefix + "mlp.w3.bias"])
    else:
        m = _r16(F.gelu(_lin16(h, p[prefix + "mlp.fc1.weight"], p[prefix + "mlp.fc1.bias"])))
        m = _lin16(m, p[prefix + "mlp.fc2.weight"], p[prefix + "mlp.fc2.bias"])
    return x + p[prefix + "ls2.gamma"] * m


def _cuda_fp32_op(fn, *args, **kw):
    """an op of CUDA autocast's fp32 list (layer_norm, softmax) while the block runs under torch.autocast on the CPU, whose policy
    leaves those ops in the input dtype (values agree either way; the BACKWARD does not: fp32 softmax saves fp32 probabilities)"""
    with torch.autocast("cpu", enabled=False):
        return fn(*[a.float() if torch.is_tensor(a) and a.is_floating_point() else a for a in args], **kw)


def vit_block_upstream_ops(p, prefix, x, num_heads, eps=1e-6, cuda_policy=False):
    """The block as the SEQUENCE OF TORCH OPS the published module issues (LayerScale a separate multiply, eager attention) -
    plain fp32 here; run under ``torch.autocast`` it shows where torch's own policy rounds (tests/test_oracle_golden.py).
    cuda_policy (with autocast on): layer_norm and softmax forced to fp32 as CUDA's autocast does - the form whose autograd
    backward the mixed-precision block re-use is compared with."""
    B, N, D = x.shape
    hd = D // num_heads
    ln = (lambda *a: _cuda_fp32_op(F.layer_norm, *a)) if cuda_policy else F.layer_norm
    sm = (lambda t: _cuda_fp32_op(torch.softmax, t, dim=-1)) if cuda_policy else (lambda t: t.softmax(dim=-1))
    y = ln(x, (D,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], eps)
    qkv = F.linear(y, p[prefix + "attn.qkv.weight"], p[prefix + "attn.qkv.bias"]).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    if cuda_policy:  # dense operands: torch's CPU bf16 matmuls lose accuracy on some strided layouts (tests/diag/diag_projector16.py)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    attn = sm(q @ k.transpose(-2, -1))
    ctx = (attn @ v).transpose(1, 2).reshape(B, N, D)
    x = x + p[prefix + "ls1.gamma"] * F.linear(ctx, p[prefix + "attn.proj.weight"], p[prefix + "attn.proj.bias"])
    y = ln(x, (D,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], eps)
    if (prefix + "mlp.w12.weight") in p:
        x1, x2 = F.linear(y, p[prefix + "mlp.w12.weight"], p[prefix + "mlp.w12.bias"]).chunk(2, dim=-1)
        m = F.linear(F.silu(x1) * x2, p[prefix + "mlp.w3.weight"], p[prefix + "mlp.w3.bias"])
    else:
        m = F.linear(F.gelu(F.linear(y, p[prefix + "mlp.fc1.weight"], p[prefix + "mlp.fc1.bias"])),
                     p[prefix + "mlp.fc2.weight"], p[prefix + "mlp.fc2.bias"])
    return x + p[prefix + "ls2.gamma"] * m


def teacher_feature_map_upstream_ops(p, x, cfg, depth=None):
    """teacher_feature_map built from vit_block_upstream_ops (same numbers in fp32; the op structure autocast acts on)"""
    B, _, H, W = x.shape
    ph, pw = H // PATCH, W // PATCH
    depth = cfg["depth"] if depth is None else depth
    t = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=PATCH).flatten(2).transpose(1, 2)
    t = torch.cat([p["cls_token"].expand(B, -1, -1), t], dim=1) + resize_pos_table(p["pos_embed"], ph, pw)
    for i in range(depth):
        t = vit_block_upstream_ops(p, f"blocks.{i}.", t, cfg["num_heads"])
    t = F.layer_norm(t, (t.shape[-1],), p["norm.weight"], p["norm.bias"], 1e-6)[:, 1:]
    return t.reshape(B, ph, pw, t.shape[-1]).permute(0, 3, 1, 2)


def teacher_feature_map_autocast(p, x, cfg, depth=None):
    """``DINOv2ViT.forward`` (models/backbones/dinov2.py:27-46) as the reference's DEFAULT trainer configuration runs it
    (train.py:263 ``precision=16``): patch-embed conv on 16-bit operands with a 16-bit result, the fp32 position table added
    in fp32 (type promotion), blocks as vit_block_autocast, final LayerNorm in fp32."""
    B, _, H, W = x.shape
    ph, pw = H // PATCH, W // PATCH
    depth = cfg["depth"] if depth is None else depth
    t = _r16(F.conv2d(_r16(x), _r16(p["patch_embed.proj.weight"]), _r16(p["patch_embed.proj.bias"]), stride=PATCH))
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat([p["cls_token"].expand(B, -1, -1), t], dim=1) + resize_pos_table(p["pos_embed"], ph, pw)
    for i in range(depth):
        t = vit_block_autocast(p, f"blocks.{i}.", t, cfg["num_heads"])
    t = F.layer_norm(t, (t.shape[-1],), p["norm.weight"], p["norm.bias"], 1e-6)[:, 1:]
    return t.reshape(B, ph, pw, t.shape[-1]).permute(0, 3, 1, 2)
