"""MiT (SegFormer Mix Transformer) student backbone on the MI355X kernels - tokens [B, H*W, C] ARE the NHWC map, so no
layout change happens anywhere - state-dict compatible with /root/reference/models/backbones/mit.py:
``patch_embed{i}.{proj.weight,proj.bias,norm.*}``, ``block{i}.{j}.{norm1,norm2}.*``,
``block{i}.{j}.attn.{q,kv,proj}.{weight,bias}``, ``block{i}.{j}.attn.{sr.weight,sr.bias,norm.*}`` (sr_ratio > 1),
``block{i}.{j}.mlp.{fc1,fc2}.{weight,bias}``, ``block{i}.{j}.mlp.dwconv.dwconv.{weight,bias}``, ``norm{i}.*``.
Every dropout / drop-path rate is 0 in the wrapper (mit_wrapper.py:76-86): the forward pass is deterministic."""
import math

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import functional16 as Fn16
from .. import ops


def _trunc_normal(w, std=0.02):
    nn.init.trunc_normal_(w, std=std)


class _Linear(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        _trunc_normal(self.weight)                                   # mit.py:57-60
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def forward(self, x, act=ops.ACT_NONE, residual=None):
        if getattr(self, "autocast16", False):   # x bf16; with a residual the fp32 stream's add rides in the epilogue
            return Fn16.Linear16Fn.apply(x, self.weight, self.bias, residual)
        return Fn.linear(x, self.weight, self.bias, act, residual)


class _Conv(nn.Module):
    """parameter holder named like nn.Conv2d: normal(0, sqrt(2 / fan_out)) weights, zero bias (mit.py:64-69)"""

    def __init__(self, cin, cout, k, groups=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
        self.weight.data.normal_(0, math.sqrt(2.0 / (k * k * cout // groups)))
        self.bias = nn.Parameter(torch.zeros(cout))


class _LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(dim)), nn.Parameter(torch.zeros(dim)), eps

    def forward(self, x, out16=False):
        if getattr(self, "autocast16", False):   # nn.LayerNorm is on autocast's fp32 list; out16: the consumer Linear's input cast
            return Fn16.LayerNorm16Fn.apply(x, self.weight, self.bias, self.eps, out16, True)
        return Fn.layer_norm(x, self.weight, self.bias, self.eps)


class DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = _Conv(dim, dim, 3, groups=dim)


class Mlp(nn.Module):
    """Mix-FFN (mit.py:36-78): fc1 -> depthwise 3x3 (+bias) -> GELU -> fc2"""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.dwconv, self.fc2 = _Linear(dim, hidden), DWConv(hidden), _Linear(hidden, dim)

    def forward(self, x, H, W, residual):
        B, N, _ = x.shape
        h = self.fc1(x)
        fn = Fn16.DWConvBiasGelu16Fn if getattr(self, "autocast16", False) else Fn.DWConvBiasGeluFn
        h = fn.apply(h.reshape(B, H, W, -1), self.dwconv.dwconv.weight, self.dwconv.dwconv.bias)
        return self.fc2(h.reshape(B, N, -1), residual=residual)


class Attention(nn.Module):
    """efficient self-attention with spatial reduction (mit.py:81-165)"""

    def __init__(self, dim, num_heads, sr_ratio):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        self.num_heads, self.scale, self.sr_ratio = num_heads, (dim // num_heads) ** -0.5, sr_ratio
        self.q, self.kv, self.proj = _Linear(dim, dim), _Linear(dim, 2 * dim), _Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = _Conv(dim, dim, sr_ratio)
            self.norm = _LayerNorm(dim)

    def forward(self, x, H, W, residual):
        B, N, C = x.shape
        if getattr(self, "autocast16", False):
            # x: norm1's fp32 output.  Its two consumers (q, and sr / kv) each cast it down: two cast nodes, so that their 16-bit
            # gradients are added in fp32 as autograd does at the fp32 tensor
            q = self.q(Fn16.ToBf16Fn.apply(x))
            x16 = Fn16.ToBf16Fn.apply(x)
            if self.sr_ratio > 1:
                r = Fn16.PatchConv16Fn.apply(x16.reshape(B, H, W, C), self.sr.weight, self.sr.bias, self.sr_ratio, False)
                x16 = self.norm(r.reshape(B, -1, C), out16=True)
            o = Fn16.CrossAttn16Fn.apply(q, self.kv(x16), self.num_heads, self.scale)
            return self.proj(o, residual=residual)
        q = self.q(x)
        if self.sr_ratio > 1:
            r = Fn.PatchConvFn.apply(x.reshape(B, H, W, C), self.sr.weight, self.sr.bias, self.sr_ratio, False)
            x = self.norm(r.reshape(B, -1, C))
        o = Fn.CrossAttnFn.apply(q, self.kv(x), self.num_heads, self.scale)
        return self.proj(o, residual=residual)


class TransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, sr_ratio):
        super().__init__()
        self.norm1, self.attn = _LayerNorm(dim), Attention(dim, num_heads, sr_ratio)
        self.norm2, self.mlp = _LayerNorm(dim), Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, H, W):
        a16 = getattr(self, "autocast16", False)
        x = self.attn(self.norm1(x), H, W, residual=x)     # x + attn(norm1(x)): the residual rides in proj's epilogue
        return self.mlp(self.norm2(x, out16=True) if a16 else self.norm2(x), H, W, residual=x)


class OverlapPatchEmbed(nn.Module):
    def __init__(self, patch_size, stride, in_chans, embed_dim):
        super().__init__()
        self.proj, self.norm = _Conv(in_chans, embed_dim, patch_size), _LayerNorm(embed_dim)
        self.patch_size, self.stride = patch_size, stride

    def forward(self, x, src_nchw):
        fn = Fn16.ConvBias16Fn if getattr(self, "autocast16", False) else Fn.ConvBiasFn
        y = fn.apply(x, self.proj.weight, self.proj.bias, self.patch_size, self.stride, src_nchw)
        B, H, W, C = y.shape
        return self.norm(y.reshape(B, H * W, C)), H, W


class MultiscaleImageTransformer(nn.Module):
    def __init__(self, embed_dims=(64, 128, 256, 512), num_heads=(1, 2, 4, 8), mlp_ratios=(4, 4, 4, 4), depths=(3, 4, 6, 3),
                 sr_ratios=(8, 4, 2, 1), out_features=None):
        super().__init__()
        self.depths = list(depths)
        cin = 3
        for i in range(4):  # registration order of the reference: the four patch embeddings first (mit.py:298-329)
            self.add_module(f"patch_embed{i + 1}", OverlapPatchEmbed(7 if i == 0 else 3, 4 if i == 0 else 2, cin, embed_dims[i]))
            cin = embed_dims[i]
        for i in range(4):
            self.add_module(f"block{i + 1}", nn.ModuleList(
                [TransformerBlock(embed_dims[i], num_heads[i], mlp_ratios[i], sr_ratios[i]) for _ in range(depths[i])]))
            self.add_module(f"norm{i + 1}", _LayerNorm(embed_dims[i]))
        self._out_features = out_features or ["res2", "res3", "res4", "res5"]
        self._out_feature_channels = {f"res{i + 2}": embed_dims[i] for i in range(4)}

    def set_precision(self, precision):
        """"fp32" (exact) or "bf16_autocast": the reference's default precision=16 policy (train.py:263) - every Linear, the
        overlapping patch embeds, the spatial-reduction conv, Mix-FFN's depthwise conv and both attention matmuls on bf16 operands
        with 16-bit results, LayerNorm / softmax / the residual stream fp32 (functional16.py)"""
        if precision not in ("fp32", "bf16_autocast"):
            raise ValueError(f"unknown student precision {precision!r}")
        self.precision = precision
        for m in self.modules():
            m.autocast16 = precision == "bf16_autocast"
        return self

    def forward_features_nhwc(self, x):
        outs = {}
        for i in range(4):
            x, H, W = getattr(self, f"patch_embed{i + 1}")(x, src_nchw=(i == 0))
            for blk in getattr(self, f"block{i + 1}"):
                x = blk(x, H, W)
            x = getattr(self, f"norm{i + 1}")(x).reshape(x.shape[0], H, W, -1)
            outs[f"res{i + 2}"] = x
        return outs

    def forward(self, x):
        return {k: v.permute(0, 3, 1, 2) for k, v in self.forward_features_nhwc(x).items()}
