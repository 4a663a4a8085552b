"""Oracle (TEST INFRASTRUCTURE): ScaleKD loss, CPU fp32 restatement.  PINNED by tests/golden/scalekd_*.npz.

Restates /root/reference/losses/scalekd.py in functional form.  Parameters arrive as a dict keyed by the
reference's state-dict names (``projector_0.pos_attention.q.weight`` ...), so a reference checkpoint maps 1:1.
Every function cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------
# attention projector                                                    losses/scalekd.py:177-245
# --------------------------------------------------------------------------------------------------
def _window_major(t, wh, ww):
    """``separate_tokens`` (losses/scalekd.py:327-335): [B,h,T,hd] -> [B,h*wh*ww,T/(wh*ww),hd]; assumes a
    square token grid (H = W = sqrt(T))."""
    B, h, T, hd = t.shape
    side = int(math.sqrt(T))
    t = t.reshape(B, h, wh, side // wh, ww, side // ww, hd).permute(0, 1, 2, 4, 3, 5, 6)
    return t.reshape(B, h * wh * ww, -1, hd)


def _fp32(fn, *args, **kw):
    """an op of CUDA autocast's fp32 list (layer_norm, softmax, ...) when the oracle runs under torch.autocast on the CPU, whose
    own policy leaves those ops in the input dtype: inputs cast up, autocast off inside"""
    with torch.autocast("cpu", enabled=False):
        return fn(*[a.float() if torch.is_tensor(a) and a.is_floating_point() else a for a in args], **kw)


def pos_attention(p, pre, x, pos, num_heads, softmax_scale, window_shapes=(1, 1), policy=False):
    """``WindowMultiheadPosAttention.forward`` (losses/scalekd.py:293-324).
    q from the positional/query stream, k and v from the student tokens; logits scaled by
    head_dim^-0.5 * softmax_scale (:308); dropouts are p=0 and gamma1 is Identity (:288-291).
    policy: the caller runs this under torch.autocast(bfloat16); softmax is then forced to fp32 as CUDA's policy does."""
    B, N, D = x.shape
    n_out = pos.shape[1]
    hd = D // num_heads
    nwin = window_shapes[0] * window_shapes[1]

    def heads(t, n):
        return t.reshape(B, n, num_heads, hd).permute(0, 2, 1, 3)

    q = heads(F.linear(pos, p[pre + "q.weight"], p[pre + "q.bias"]), n_out)
    k = heads(F.linear(x, p[pre + "k.weight"], p[pre + "k.bias"]), N)
    v = heads(F.linear(x, p[pre + "v.weight"], p[pre + "v.bias"]), N)
    if nwin > 1:
        q, k, v = (_window_major(t, *window_shapes) for t in (q, k, v))
    logits = (q @ k.transpose(-1, -2)) * (hd ** -0.5) * softmax_scale
    w = _fp32(torch.softmax, logits, dim=-1) if policy else torch.softmax(logits, dim=-1)
    # :313-314 - the windowed result is only *viewed* back; tokens stay window-major (no inverse permute)
    o = (w @ v).reshape(B, num_heads, n_out, hd).transpose(1, 2).reshape(B, n_out, D)
    return F.linear(o, p[pre + "proj.weight"], p[pre + "proj.bias"])


def attention_projector(p, pre, x, query, hw, num_heads, softmax_scale, window_shapes=(1, 1),
                        training=True, bn_momentum=0.1, policy=False):
    """``AttentionProjector.forward`` (losses/scalekd.py:228-245).

    x: student feature [B,Cs,H,W]; query: None (use the learned ``query.weight``) or tokens [B,HW,D].
    Returns tokens [B,HW,D].  BatchNorm runs with batch statistics when ``training`` and updates
    ``running_mean/var`` in ``p`` in place like nn.BatchNorm2d (momentum 0.1, eps 1e-5) (:199-201).
    FFN is Linear-ReLU-Linear with the *normalised* input as residual (:206-213, :448, :464-471)."""
    H, W = hw
    B = x.shape[0]
    D = p[pre + "proj_student.0.weight"].shape[0]
    z = F.conv2d(x, p[pre + "proj_student.0.weight"], p[pre + "proj_student.0.bias"])
    z = F.batch_norm(z, p[pre + "proj_student.1.running_mean"], p[pre + "proj_student.1.running_var"],
                     p[pre + "proj_student.1.weight"], p[pre + "proj_student.1.bias"],
                     training, bn_momentum, 1e-5)
    if training and (pre + "proj_student.1.num_batches_tracked") in p:
        p[pre + "proj_student.1.num_batches_tracked"] += 1
    z = torch.relu(z) + p[pre + "pos_embed"]
    tok = z.flatten(2).transpose(1, 2)  # [B,HW,D]   (:241)
    if policy:  # torch's CPU bf16 Linear on a strided (transposed) input takes a path that loses 3e-3 against fp32 accumulation
        tok = tok.contiguous()  # (measured: tests/diag/diag_projector16.py); dense, it accumulates in fp32 like the GPU kernels
    if query is not None:
        pos = query  # :232 + :240 are a net identity on [B,HW,D]
    elif (pre + "query.weight") in p:
        pos = p[pre + "query.weight"].unsqueeze(0).expand(B, -1, -1)  # :234
    else:
        raise NotImplementedError("There is no query!")  # :236
    a = pos_attention(p, pre + "pos_attention.", tok, pos, num_heads, softmax_scale, window_shapes, policy)
    y = a + tok  # :243
    ln = (lambda *a_: _fp32(F.layer_norm, *a_)) if policy else F.layer_norm   # (policy: see pos_attention)
    zn = ln(y, (D,), p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-5)
    f = F.linear(torch.relu(F.linear(zn, p[pre + "ffn.layers.0.0.weight"], p[pre + "ffn.layers.0.0.bias"])),
                 p[pre + "ffn.layers.1.weight"], p[pre + "ffn.layers.1.bias"])
    return ln(zn + f, (D,), p[pre + "norm_2.weight"], p[pre + "norm_2.bias"], 1e-5)


# --------------------------------------------------------------------------------------------------
# loss tails                                                             losses/scalekd.py:67-127
# --------------------------------------------------------------------------------------------------
def _normalised_mse_and_cos(s, t, alpha, batch):
    """F.normalize(dim=1) (eps 1e-12), MSE(sum)/batch*alpha, mean cosine (eps 1e-8) of the already
    normalised maps (losses/scalekd.py:82-90 / :119-125)."""
    s = F.normalize(s, dim=1)
    t = F.normalize(t, dim=1)
    loss = ((s - t) ** 2).sum() / batch * alpha
    return loss, F.cosine_similarity(s, t, dim=1).mean()


def tokens_to_map(tok, shape):
    """:79 / :109  ``permute(0,2,1).contiguous().view(N,C,H,W)``."""
    return tok.permute(0, 2, 1).reshape(shape)


def spat_loss(tok_s, feat_t, alpha0):
    """``ScaleKD.get_spat_loss`` (losses/scalekd.py:67-92).  tok_s [B,HW,C], feat_t [B,C,H,W]."""
    return _normalised_mse_and_cos(tokens_to_map(tok_s, feat_t.shape), feat_t, alpha0, feat_t.shape[0])


def dct_matrices(n, dtype=torch.float32):
    """Unnormalised DCT-II matrix and its exact inverse as the reference builds them from fft(eye)
    (losses/scalekd.py:343-350 with ``_dct``:352-380, ``_idct``:382-418; norm=None):
        X_k = 2 * sum_j x_j cos(pi (2j+1) k / 2n),   x_j = X_0/(2n) + (1/n) sum_{k>=1} X_k cos(pi (2j+1) k / 2n).
    Returned as (fwd, inv) such that ``X = x @ fwd`` and ``x = X @ inv`` along the last axis."""
    j = torch.arange(n, dtype=torch.float64)
    ang = math.pi * (2 * j[:, None] + 1) * j[None, :] / (2 * n)  # [j, k]
    fwd = 2 * torch.cos(ang)
    inv = torch.cos(ang).t() / n  # [k, j]
    inv[0] = inv[0] / 2
    return fwd.to(dtype), inv.to(dtype)


def drop_dc(m):
    """2-D DCT over (H,W), zero coefficient [0,0], inverse 2-D DCT (losses/scalekd.py:111-118,420-428).
    Requires H == W (``DCT(resolution=H)`` is used for both axes, :107)."""
    n = m.shape[-1]
    fwd, inv = dct_matrices(n, m.dtype)
    c = (m @ fwd).transpose(-1, -2) @ fwd  # transform W then H; result is [.., kw, kh]
    c = c.transpose(-1, -2).clone()
    c[..., 0, 0] = 0
    r = (c @ inv).transpose(-1, -2) @ inv
    return r.transpose(-1, -2)


def freq_loss(tok_s, feat_t, alpha1, policy=False):
    """``ScaleKD.get_freq_loss`` (losses/scalekd.py:95-127).  policy (the caller runs under torch.autocast(bfloat16)): the
    reference's DCT is built from nn.Linear layers (:337-428), i.e. four 16-bit matmuls per map with 16-bit matrices - the four
    ``@`` of drop_dc are cast down by autocast exactly like those Linears; the normalise / MSE / cosine tail is fp32 on CUDA."""
    m = tokens_to_map(tok_s, feat_t.shape)
    if policy:
        m, feat_t = m.contiguous(), feat_t.contiguous()  # (dense matmul operands: see attention_projector)
    s = drop_dc(m)
    t = drop_dc(feat_t)
    if policy:
        s, t = s.float(), t.float()
    return _normalised_mse_and_cos(s, t, alpha1, feat_t.shape[0])


# --------------------------------------------------------------------------------------------------
# ScaleKD.forward                                                        losses/scalekd.py:31-54
# --------------------------------------------------------------------------------------------------
def scalekd_forward(p, pre, feat_s, feat_t, cfg, query_s=None, query_f=None, training=True, policy=False):
    """cfg: dict(alpha, hw, num_heads, softmax_scale, window_shapes)."""
    kw = dict(hw=cfg["hw"], num_heads=cfg["num_heads"], window_shapes=tuple(cfg.get("window_shapes", (1, 1))),
              training=training, policy=policy)
    ts = attention_projector(p, pre + "projector_0.", feat_s, query_s, softmax_scale=cfg["softmax_scale"][0], **kw)
    tf = attention_projector(p, pre + "projector_1.", feat_s, query_f, softmax_scale=cfg["softmax_scale"][1], **kw)
    ls, ss = spat_loss(ts, feat_t, cfg["alpha"][0])
    lf, sf = freq_loss(tf, feat_t, cfg["alpha"][1], policy)
    return {"spatial_loss": ls, "frequency_loss": lf, "spatial_similarity": ss,
            "frequency_similarity": sf, "loss": ls + lf}
