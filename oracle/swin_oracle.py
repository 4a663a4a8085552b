"""Oracle (TEST INFRASTRUCTURE): Swin-Transformer student, CPU fp32 restatement.
PINNED by tests/golden/swin_small.npz (generated from the reference's own backbone).

Restates /root/reference/models/backbones/swin.py in functional form; parameter names are the reference's
state-dict names of ``D2SwinTransformer`` (``patch_embed.proj.weight``, ``layers.{i}.blocks.{j}.attn.qkv.weight``,
``layers.{i}.downsample.reduction.weight``, ``norm{i}.weight`` ...).  Behaviours reproduced on purpose:
  * the image is zero-padded on the right/bottom to a multiple of the patch size (swin.py:524-529);
  * every block pads its (already LayerNorm-ed) token grid with ZERO rows to a multiple of the window (:283-287);
    padded tokens take part in attention (their q/k/v are the qkv bias);
  * odd blocks roll by -window//2 and add a 0/-100 region mask built on the PADDED grid (:454-481), even when the
    grid is no larger than one window;
  * PatchMerging pads odd grids with zeros and concatenates (even,even),(odd,even),(even,odd),(odd,odd) (:359-371);
  * ``init_weights`` defines a function and never applies it (:680-694): weights keep torch's default initialisers,
    only the relative-position table is trunc-normal(0.02) (:148);
  * ``relative_position_index`` is an (integer, non-trainable) entry of the state dict (:139-141);
  * all four ``norm{i}`` are applied, whatever ``out_features`` asks for (:711-727).
Dropout (drop_rate 0.3 in models/wrappers/swin_wrapper.py:89) is NOT restated: parity runs use p=0 / eval mode
(SURVEY.md section 8d), where it is the identity.
"""
import math

import torch
import torch.nn.functional as F

# models/wrappers/swin_wrapper.py:11-28
SWIN_CONFIGS = {
    "swin_tiny": dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window=7),
    "swin_small": dict(embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], window=7),
}
PATCH = 4
MLP_RATIO = 4


def relative_position_index(ws):
    """[ws*ws, ws*ws] int64 index into the (2ws-1)^2 bias table (swin.py:124-138)."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    y, x = ys.reshape(-1), xs.reshape(-1)
    dy = y[:, None] - y[None, :] + ws - 1
    dx = x[:, None] - x[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


def init_swin_params(cfg, seed=1, in_chans=3, dtype=torch.float32):
    """torch default initialisers (the reference's init_weights is a no-op, swin.py:680-694): Linear / Conv2d
    kaiming_uniform(a=sqrt 5) + bias U(+-1/sqrt(fan_in)); LayerNorm (1, 0); bias table trunc-normal(0.02)."""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def linear(prefix, out_f, in_f, bias=True, fan_in=None):
        fan_in = in_f if fan_in is None else fan_in
        bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
        p[prefix + ".weight"] = (torch.rand(out_f, in_f, generator=g, dtype=dtype) * 2 - 1) * bound
        if bias:
            p[prefix + ".bias"] = (torch.rand(out_f, generator=g, dtype=dtype) * 2 - 1) * bound

    def ln(prefix, d):
        p[prefix + ".weight"] = torch.ones(d, dtype=dtype)
        p[prefix + ".bias"] = torch.zeros(d, dtype=dtype)

    C0, ws = cfg["embed_dim"], cfg["window"]
    linear("patch_embed.proj", C0, in_chans * PATCH * PATCH)
    p["patch_embed.proj.weight"] = p["patch_embed.proj.weight"].view(C0, in_chans, PATCH, PATCH)
    ln("patch_embed.norm", C0)
    n_layers = len(cfg["depths"])
    for i in range(n_layers):
        C, nH = C0 * 2 ** i, cfg["num_heads"][i]
        for j in range(cfg["depths"][i]):
            b = f"layers.{i}.blocks.{j}."
            ln(b + "norm1", C)
            t = torch.empty((2 * ws - 1) ** 2, nH, dtype=dtype)
            torch.nn.init.trunc_normal_(t, std=0.02, generator=g)
            p[b + "attn.relative_position_bias_table"] = t
            p[b + "attn.relative_position_index"] = relative_position_index(ws)
            linear(b + "attn.qkv", 3 * C, C)
            linear(b + "attn.proj", C, C)
            ln(b + "norm2", C)
            linear(b + "mlp.fc1", MLP_RATIO * C, C)
            linear(b + "mlp.fc2", C, MLP_RATIO * C)
        if i < n_layers - 1:
            linear(f"layers.{i}.downsample.reduction", 2 * C, 4 * C, bias=False)
            ln(f"layers.{i}.downsample.norm", 4 * C)
    for i in range(n_layers):
        ln(f"norm{i}", C0 * 2 ** i)
    return p


def to_windows(x, ws):
    """[B,Hp,Wp,C] -> [B*nW, ws*ws, C], windows row-major (swin.py:46-59)."""
    B, Hp, Wp, C = x.shape
    x = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).transpose(2, 3)
    return x.reshape(-1, ws * ws, C)


def from_windows(w, ws, Hp, Wp):
    """inverse of to_windows (swin.py:28-43)."""
    C = w.shape[-1]
    x = w.reshape(-1, Hp // ws, Wp // ws, ws, ws, C).transpose(2, 3)
    return x.reshape(-1, Hp, Wp, C)


def region_labels(Hp, Wp, ws, shift):
    """[Hp,Wp] label image of the 3x3 regions a cyclic shift creates (swin.py:457-472)."""
    lab = torch.zeros(Hp, Wp)
    edges_h = [(0, Hp - ws), (Hp - ws, Hp - shift), (Hp - shift, Hp)]
    edges_w = [(0, Wp - ws), (Wp - ws, Wp - shift), (Wp - shift, Wp)]
    c = 0
    for h0, h1 in edges_h:
        for w0, w1 in edges_w:
            lab[h0:h1, w0:w1] = c
            c += 1
    return lab


def shift_mask(H, W, ws):
    """[nW, ws*ws, ws*ws] additive mask (0 / -100) of a stage, built on the padded grid (swin.py:455-481)."""
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    lab = to_windows(region_labels(Hp, Wp, ws, ws // 2)[None, :, :, None], ws)[..., 0]  # [nW, ws*ws]
    diff = lab[:, None, :] - lab[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


# ---- the same functions under torch.autocast(bfloat16) with CUDA's op policy (the reference's default precision=16, train.py:263).
# The oracle runs on the CPU, whose autocast policy leaves layer_norm and softmax in the input dtype where CUDA's runs them in
# fp32 (torch/csrc/autocast_mode.cpp, the `fp32` cast-policy list): with policy=True those call sites are forced to fp32 with
# explicit casts, and the matmul operands are made dense (torch's CPU bf16 matmuls lose accuracy on some strided layouts,
# tests/diag/diag_projector16.py).  The CALLER opens torch.autocast("cpu", dtype=torch.bfloat16).  PARITY UNPINNED for that policy (no
# CUDA device to run the reference's autocast on); the ops themselves are the pinned fp32 restatement.
def _fp32(fn, *args, **kw):
    with torch.autocast("cpu", enabled=False):
        return fn(*[a.float() if torch.is_tensor(a) and a.is_floating_point() else a for a in args], **kw)


def _ln(x, w, b, policy):
    return _fp32(F.layer_norm, x, (x.shape[-1],), w, b, 1e-5) if policy else F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def window_attention(p, b, xw, nH, mask, policy=False):
    """swin.py:151-199 on [B_, N, C] windows."""
    B_, N, C = xw.shape
    hd = C // nH
    qkv = F.linear(xw, p[b + "qkv.weight"], p[b + "qkv.bias"]).reshape(B_, N, 3, nH, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    if policy:
        q, kt, v = q.contiguous(), k.transpose(-2, -1).contiguous(), v.contiguous()
    else:
        kt = k.transpose(-2, -1)
    logits = q @ kt
    bias = p[b + "relative_position_bias_table"][p[b + "relative_position_index"].reshape(-1)]
    logits = logits + bias.reshape(N, N, nH).permute(2, 0, 1)[None]   # (under autocast: 16-bit + fp32 -> fp32)
    if mask is not None:
        nW = mask.shape[0]
        logits = (logits.reshape(B_ // nW, nW, nH, N, N) + mask[None, :, None]).reshape(B_, nH, N, N)
    w = _fp32(torch.softmax, logits, dim=-1) if policy else torch.softmax(logits, dim=-1)
    ctx = (w @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(ctx, p[b + "proj.weight"], p[b + "proj.bias"])


def swin_block(p, b, x, H, W, nH, ws, shift, mask, policy=False):
    """swin.py:268-334, dropout / drop-path at rate 0."""
    B, L, C = x.shape
    y = _ln(x, p[b + "norm1.weight"], p[b + "norm1.bias"], policy).reshape(B, H, W, C)
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    y = F.pad(y, (0, 0, 0, Wp - W, 0, Hp - H))
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    a = window_attention(p, b + "attn.", to_windows(y, ws), nH, mask if shift > 0 else None, policy)
    y = from_windows(a, ws, Hp, Wp)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    x = x + y[:, :H, :W].reshape(B, L, C)
    z = _ln(x, p[b + "norm2.weight"], p[b + "norm2.bias"], policy)
    z = F.linear(F.gelu(F.linear(z, p[b + "mlp.fc1.weight"], p[b + "mlp.fc1.bias"])), p[b + "mlp.fc2.weight"], p[b + "mlp.fc2.bias"])
    return x + z


def patch_merging(p, b, x, H, W, policy=False):
    """swin.py:350-376."""
    B, L, C = x.shape
    x = F.pad(x.reshape(B, H, W, C), (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], dim=-1)
    x = x.reshape(B, -1, 4 * C)
    x = _ln(x, p[b + "norm.weight"], p[b + "norm.bias"], policy)
    return F.linear(x, p[b + "reduction.weight"])


def forward_features(p, x, cfg, policy=False, tokens=None):
    """SwinTransformer.forward (swin.py:696-727): {'res2'..} NCHW feature maps of every stage.  tokens: an optional dict that
    receives every block's and every PatchMerging's output tokens (teacher-forced per-stage tests)."""
    ws = cfg["window"]
    _, _, H, W = x.shape
    x = F.pad(x, (0, (-W) % PATCH, 0, (-H) % PATCH))
    x = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=PATCH)
    B, C, H, W = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = _ln(x, p["patch_embed.norm.weight"], p["patch_embed.norm.bias"], policy)
    outs = {}
    n_layers = len(cfg["depths"])
    for i in range(n_layers):
        nH = cfg["num_heads"][i]
        mask = shift_mask(H, W, ws).to(torch.float32)
        for j in range(cfg["depths"][i]):
            x = swin_block(p, f"layers.{i}.blocks.{j}.", x, H, W, nH, ws, 0 if j % 2 == 0 else ws // 2, mask, policy)
            if tokens is not None:
                tokens[f"layers.{i}.blocks.{j}"] = x
        C = x.shape[-1]
        o = _ln(x, p[f"norm{i}.weight"], p[f"norm{i}.bias"], policy)
        outs[f"res{i + 2}"] = o.reshape(B, H, W, C).permute(0, 3, 1, 2)
        if i < n_layers - 1:
            x = patch_merging(p, f"layers.{i}.downsample.", x, H, W, policy)
            if tokens is not None:
                tokens[f"layers.{i}.downsample"] = x
            H, W = (H + 1) // 2, (W + 1) // 2
    return outs


def student_features(p, x, cfg, n_patches, keys=("res4", "res5"), policy=False):
    """ModelWrapper.forward (models/model_zoo.py:116-129): bilinear resize of the chosen levels to the teacher grid."""
    feats = forward_features(p, x, cfg, policy)
    return {k: F.interpolate(feats[k], size=tuple(n_patches), mode="bilinear", align_corners=False) for k in keys}
