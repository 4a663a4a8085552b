"""Swin-Transformer student backbone on MI355X kernels, state-dict compatible with
/root/reference/models/backbones/swin.py (``patch_embed.proj``, ``layers.{i}.blocks.{j}.attn.qkv``,
``layers.{i}.blocks.{j}.attn.relative_position_bias_table`` / ``relative_position_index``,
``layers.{i}.downsample.{reduction,norm}``, ``norm{i}``).

Tokens stay [B, H*W, C] (token-major) end to end.  Every re-ordering the reference performs with F.pad / torch.roll /
view-permute-contiguous chains (window partition and merge, cyclic shift, PatchMerging's strided concat) is a single
row gather with an index map computed once per (batch, grid, window, shift) and cached on the device.
"""
import math

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import functional16 as Fn16

PATCH = 4


class _P(nn.Module):
    """parameter holder with reference-compatible attribute names"""


def _linear(out_f, in_f, bias=True):
    """nn.Linear's default initialiser - the reference never applies its init_weights (swin.py:680-694)"""
    m = _P()
    m.weight = nn.Parameter(torch.empty(out_f, in_f))
    nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
    if bias:
        bound = 1.0 / math.sqrt(in_f)
        m.bias = nn.Parameter(torch.empty(out_f).uniform_(-bound, bound))
    return m


def _ln(c):
    m = _P()
    m.weight, m.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))
    return m


def relative_position_index(ws):
    """swin.py:124-138"""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    y, x = ys.reshape(-1), xs.reshape(-1)
    return (y[:, None] - y[None, :] + ws - 1) * (2 * ws - 1) + (x[:, None] - x[None, :] + ws - 1)


class BlockGeometry:
    """index maps of one (batch, grid, window, shift) combination; int32, on the device"""

    def __init__(self, B, H, W, ws, shift, device):
        Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
        self.N, self.nW = ws * ws, (Hp // ws) * (Wp // ws)
        ys = torch.arange(Hp)[:, None].expand(Hp, Wp)
        xs = torch.arange(Wp)[None, :].expand(Hp, Wp)
        sy, sx = (ys + shift) % Hp, (xs + shift) % Wp  # torch.roll(x, -shift): position p of the shifted grid holds p + shift
        tok = torch.where((sy < H) & (sx < W), sy * W + sx, torch.full_like(sy, -1))  # -1: zero-padded position (swin.py:287)

        def windows(t):  # [Hp, Wp] -> [nW * N], windows row-major, slots row-major (swin.py:55-58)
            return t.reshape(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1)

        slot_tok = windows(tok)
        offs = (torch.arange(B) * (H * W))[:, None]
        to_win = torch.where(slot_tok[None] >= 0, slot_tok[None] + offs, torch.full_like(slot_tok[None].expand(B, -1), -1)).reshape(-1)
        valid = to_win >= 0
        from_win = torch.full((B * H * W,), -1, dtype=torch.long)
        from_win[to_win[valid]] = torch.arange(to_win.numel())[valid]
        assert int(from_win.min()) >= 0  # every token sits in exactly one window slot
        self.to_win = to_win.to(torch.int32).to(device)      # window slot -> token row (-1: pad)
        self.from_win = from_win.to(torch.int32).to(device)  # token row -> window slot
        self.labels = None
        if shift > 0:  # region ids of the shifted grid (swin.py:457-472); mask = (label_i != label_j) * -100
            lab = torch.zeros(Hp, Wp, dtype=torch.long)
            c = 0
            for h0, h1 in ((0, Hp - ws), (Hp - ws, Hp - shift), (Hp - shift, Hp)):
                for w0, w1 in ((0, Wp - ws), (Wp - ws, Wp - shift), (Wp - shift, Wp)):
                    lab[h0:h1, w0:w1] = c
                    c += 1
            self.labels = windows(lab).to(torch.int32).to(device)
        self.rel_index = relative_position_index(ws).reshape(-1).to(torch.int32).to(device)


class MergeGeometry:
    """PatchMerging (swin.py:359-371): output row (b, i, j) = [x(2i,2j) | x(2i+1,2j) | x(2i,2j+1) | x(2i+1,2j+1)], zeros past an odd edge"""

    def __init__(self, B, H, W, device):
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        i = torch.arange(H2)[:, None, None]
        j = torch.arange(W2)[None, :, None]
        dy = torch.tensor([0, 1, 0, 1])[None, None, :]
        dx = torch.tensor([0, 0, 1, 1])[None, None, :]
        y, x = 2 * i + dy, 2 * j + dx
        tok = torch.where((y < H) & (x < W), y * W + x, torch.full_like(y * W + x, -1)).reshape(-1)  # [H2*W2*4]
        offs = (torch.arange(B) * (H * W))[:, None]
        merge = torch.where(tok[None] >= 0, tok[None] + offs, torch.full_like(tok[None].expand(B, -1), -1)).reshape(-1)
        valid = merge >= 0
        unmerge = torch.full((B * H * W,), -1, dtype=torch.long)
        unmerge[merge[valid]] = torch.arange(merge.numel())[valid]
        assert int(unmerge.min()) >= 0
        self.merge = merge.to(torch.int32).to(device)
        self.unmerge = unmerge.to(torch.int32).to(device)


_GEOM = {}


def _geom(kind, *key, device):
    k = (kind, *key, str(device))
    if k not in _GEOM:
        _GEOM[k] = (BlockGeometry if kind == "block" else MergeGeometry)(*key, device)
    return _GEOM[k]


class WindowAttention(nn.Module):
    def __init__(self, dim, ws, num_heads):
        super().__init__()
        t = torch.zeros((2 * ws - 1) ** 2, num_heads)
        nn.init.trunc_normal_(t, std=0.02)  # swin.py:148
        self.relative_position_bias_table = nn.Parameter(t)
        # an (integer) nn.Parameter in the reference too (swin.py:139-141): part of the state dict and of parameters()
        self.relative_position_index = nn.Parameter(relative_position_index(ws), requires_grad=False)
        self.qkv = _linear(3 * dim, dim)
        self.proj = _linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = _linear(hidden, dim), _linear(dim, hidden)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio, drop):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size, self.drop = dim, num_heads, window_size, shift_size, drop
        self.norm1 = _ln(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.norm2 = _ln(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, H, W):
        B = x.shape[0]
        g = _geom("block", B, H, W, self.window_size, self.shift_size, device=x.device)
        a, m = self.attn, self.mlp
        fn = Fn16.SwinBlock16Fn if getattr(self, "autocast16", False) else Fn.SwinBlockFn
        return fn.apply(x, g, self.num_heads, self.drop if self.training else 0.0, self.norm1.weight, self.norm1.bias,
                                    a.relative_position_bias_table, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                                    self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = _linear(2 * dim, 4 * dim, bias=False)
        self.norm = _ln(4 * dim)

    def forward(self, x, H, W):
        g = _geom("merge", x.shape[0], H, W, device=x.device)
        fn = Fn16.PatchMerge16Fn if getattr(self, "autocast16", False) else Fn.PatchMergeFn
        return fn.apply(x, g, self.norm.weight, self.norm.bias, self.reduction.weight)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, drop, downsample):
        super().__init__()
        self.blocks = nn.ModuleList(SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2,
                                                         mlp_ratio, drop) for i in range(depth))
        self.downsample = PatchMerging(dim) if downsample else None

    def forward(self, x, H, W):
        for blk in self.blocks:
            x = blk(x, H, W)
        if self.downsample is not None:
            return x, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
        return x, x, H, W


class PatchEmbed(nn.Module):
    def __init__(self, in_chans, embed_dim):
        super().__init__()
        self.proj = _P()
        self.proj.weight = nn.Parameter(torch.empty(embed_dim, in_chans, PATCH, PATCH))
        nn.init.kaiming_uniform_(self.proj.weight, a=math.sqrt(5))  # nn.Conv2d default
        bound = 1.0 / math.sqrt(in_chans * PATCH * PATCH)
        self.proj.bias = nn.Parameter(torch.empty(embed_dim).uniform_(-bound, bound))
        self.norm = _ln(embed_dim)


class SwinTransformer(nn.Module):
    """swin.py:541-727 with the D2 wrapper's fixed choices (ape=False, patch_norm=True, out_indices all, drop_path 0)."""

    def __init__(self, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4, drop_rate=0.0,
                 in_chans=3):
        super().__init__()
        self.embed_dim, self.num_layers, self.drop_rate = embed_dim, len(depths), drop_rate
        self.patch_embed = PatchEmbed(in_chans, embed_dim)
        self.layers = nn.ModuleList(BasicLayer(embed_dim * 2 ** i, depths[i], num_heads[i], window_size, mlp_ratio, drop_rate,
                                               downsample=i < len(depths) - 1) for i in range(len(depths)))
        self.num_features = [embed_dim * 2 ** i for i in range(len(depths))]
        for i, c in enumerate(self.num_features):
            self.add_module(f"norm{i}", _ln(c))

    def set_precision(self, precision):
        """"fp32" (exact) or "bf16_autocast": the reference's default trainer precision (train.py:263), i.e. what torch autocast makes
        of swin.py (functional16.SwinBlock16Fn): 16-bit Linears / conv / window attention, fp32 LayerNorm and softmax, the residual
        stream fp32 in stage 0 and 16-bit after the first PatchMerging"""
        if precision not in ("fp32", "bf16_autocast"):
            raise ValueError(f"unknown precision {precision!r}")
        self.autocast16 = precision == "bf16_autocast"
        for m in self.modules():
            if isinstance(m, (SwinTransformerBlock, PatchMerging)):
                m.autocast16 = self.autocast16
        return self

    def _forward_features_autocast(self, x):
        B, _, H, W = x.shape
        pe = self.patch_embed
        x = Fn16.PatchConv16Fn.apply(x, pe.proj.weight, pe.proj.bias, PATCH, True)           # bf16 NHWC
        H, W = x.shape[1], x.shape[2]
        x = Fn16.LayerNorm16Fn.apply(x.reshape(B, H * W, self.embed_dim), pe.norm.weight, pe.norm.bias, 1e-5, False, True)  # fp32
        if self.training and self.drop_rate > 0.0:
            x = Fn.DropoutFn.apply(x, self.drop_rate)  # pos_drop, on the fp32 norm output
        outs = {}
        for i, layer in enumerate(self.layers):
            x_out, x, Hn, Wn = layer(x, H, W)
            n = getattr(self, f"norm{i}")
            outs[f"res{i + 2}"] = Fn16.LayerNorm16Fn.apply(x_out, n.weight, n.bias, 1e-5, False, True).reshape(B, H, W, self.num_features[i])
            H, W = Hn, Wn
        return outs

    def forward_features_nhwc(self, x):
        if getattr(self, "autocast16", False):
            return self._forward_features_autocast(x)
        B, _, H, W = x.shape
        pe = self.patch_embed
        x = Fn.PatchConvFn.apply(x, pe.proj.weight, pe.proj.bias, PATCH, True)  # ragged edge == the zero pad of swin.py:526-529
        H, W = x.shape[1], x.shape[2]
        x = Fn.layer_norm(x.reshape(B, H * W, self.embed_dim), pe.norm.weight, pe.norm.bias, 1e-5)
        if self.training and self.drop_rate > 0.0:
            x = Fn.DropoutFn.apply(x, self.drop_rate)  # pos_drop
        outs = {}
        for i, layer in enumerate(self.layers):
            x_out, x, Hn, Wn = layer(x, H, W)
            n = getattr(self, f"norm{i}")
            outs[f"res{i + 2}"] = Fn.layer_norm(x_out, n.weight, n.bias, 1e-5).reshape(B, H, W, self.num_features[i])
            H, W = Hn, Wn
        return outs

    def forward(self, x):
        """reference contract (swin.py:711-727): dict res2..res5 of [B,C,H,W] (here: views of token-major memory)"""
        return {k: v.permute(0, 3, 1, 2) for k, v in self.forward_features_nhwc(x).items()}


class D2SwinTransformer(SwinTransformer):
    def __init__(self, embed_dims, depths, num_heads, window_size, drop_rate, out_features=None, drop_path_rate=0.0,
                 attn_drop_rate=0.0):
        if drop_path_rate != 0.0 or attn_drop_rate != 0.0:
            raise NotImplementedError("drop_path / attention dropout are 0 in the reference wrapper (swin_wrapper.py:90-91)")
        super().__init__(embed_dims, depths, num_heads, window_size, 4, drop_rate)
        self._out_features = out_features or ["res2", "res3", "res4", "res5"]
        self._out_feature_strides = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
        self._out_feature_channels = {f"res{i + 2}": self.num_features[i] for i in range(4)}

    def forward(self, x):
        if x.dim() != 4:
            raise AssertionError(f"SwinTransformer takes an input of shape (N, C, H, W). Got {x.shape} instead!")
        return {k: v for k, v in super().forward(x).items() if k in self._out_features}
