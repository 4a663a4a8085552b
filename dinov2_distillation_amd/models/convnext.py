"""ConvNeXt-V2 student backbone on MI355X kernels (NHWC inside), state-dict compatible with
/root/reference/models/backbones/convnext.py (``downsample_layers.{i}.{j}``, ``stages.{i}.{j}.dwconv`` ...)."""
import torch
import torch.nn as nn

from .. import functional as Fn
from .. import functional16 as Fn16


class _P(nn.Module):
    """parameter holder with reference-compatible attribute names"""


def _tn(*shape):
    t = torch.empty(*shape)
    nn.init.trunc_normal_(t, std=0.02)  # convnext.py:150-153
    return nn.Parameter(t)


def _conv(o, i, k):
    m = _P()
    m.weight, m.bias = _tn(o, i, k, k), nn.Parameter(torch.zeros(o))
    return m


def _ln(c):
    m = _P()
    m.weight, m.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))
    return m


class Block(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = _conv(dim, 1, 7)
        self.norm = _ln(dim)
        self.pwconv1, self.pwconv2, self.grn = _P(), _P(), _P()
        self.pwconv1.weight, self.pwconv1.bias = _tn(4 * dim, dim), nn.Parameter(torch.zeros(4 * dim))
        self.grn.gamma = nn.Parameter(torch.zeros(1, 1, 1, 4 * dim))
        self.grn.beta = nn.Parameter(torch.zeros(1, 1, 1, 4 * dim))
        self.pwconv2.weight, self.pwconv2.bias = _tn(dim, 4 * dim), nn.Parameter(torch.zeros(dim))

    def forward(self, x, autocast=False):  # x NHWC
        fn = Fn16.ConvNeXtBlock16Fn if autocast else Fn.ConvNeXtBlockFn
        return fn.apply(x, self.dwconv.weight, self.dwconv.bias, self.norm.weight, self.norm.bias,
                        self.pwconv1.weight, self.pwconv1.bias, self.grn.gamma, self.grn.beta,
                        self.pwconv2.weight, self.pwconv2.bias)


class ConvNeXtV2(nn.Module):
    def __init__(self, depths, dims, in_chans=3):
        super().__init__()
        self.depths, self.dims = list(depths), list(dims)
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.ModuleList([_conv(dims[0], in_chans, 4), _ln(dims[0])]))  # stem :116-120
        for i in range(3):
            self.downsample_layers.append(nn.ModuleList([_ln(dims[i]), _conv(dims[i + 1], dims[i], 2)]))  # :121-126
        self.stages = nn.ModuleList(nn.ModuleList(Block(dims[i]) for _ in range(depths[i])) for i in range(4))
        self.precision = "fp32"

    def set_precision(self, precision):
        """"fp32": exact fp32 (every parity claim and bench.py's `value`); "bf16_autocast": the reference's default precision=16
        policy (train.py:263) in bf16 with bf16 activations in HBM (functional16.py); the features are handed on as fp32"""
        if precision not in ("fp32", "bf16_autocast"):
            raise ValueError(f"unknown student precision {precision!r}")
        self.precision = precision
        return self

    def _forward_features_nhwc_autocast(self, x):
        outs = {}
        for i in range(4):
            ds = self.downsample_layers[i]
            if i == 0:  # stem conv -> 16 bit; its channels-first LayerNorm returns fp32: stage 0 keeps an fp32 residual stream
                x = Fn16.PatchConv16Fn.apply(x, ds[0].weight, ds[0].bias, 4, True)
                x = Fn16.LayerNorm16Fn.apply(x, ds[1].weight, ds[1].bias, 1e-6, False)
            else:       # LayerNorm (fp32 arithmetic) -> the conv's 16-bit input -> 16-bit output
                x = Fn16.LayerNorm16Fn.apply(x, ds[0].weight, ds[0].bias, 1e-6, True)
                x = Fn16.PatchConv16Fn.apply(x, ds[1].weight, ds[1].bias, 2, False)
            for blk in self.stages[i]:
                x = blk(x, autocast=True)
            outs[f"res{i + 2}"] = x if x.dtype == torch.float32 else Fn16.ToFloatFn.apply(x)
        return outs

    def forward_features_nhwc(self, x):
        if self.precision == "bf16_autocast":
            return self._forward_features_nhwc_autocast(x)
        outs = {}
        for i in range(4):
            ds = self.downsample_layers[i]
            if i == 0:
                x = Fn.PatchConvFn.apply(x, ds[0].weight, ds[0].bias, 4, True)
                x = Fn.layer_norm(x, ds[1].weight, ds[1].bias, 1e-6)  # channels_first LN == row LN in NHWC
            else:
                x = Fn.layer_norm(x, ds[0].weight, ds[0].bias, 1e-6)
                x = Fn.PatchConvFn.apply(x, ds[1].weight, ds[1].bias, 2, False)
            for blk in self.stages[i]:
                x = blk(x)
            outs[f"res{i + 2}"] = x
        return outs

    def forward_features(self, x):
        """reference contract (convnext.py:155-162): dict res2..res5 of [B,C,H,W] (here: views of NHWC memory)"""
        return {k: v.permute(0, 3, 1, 2) for k, v in self.forward_features_nhwc(x).items()}

    forward = forward_features


class D2ConvNextV2(ConvNeXtV2):
    def __init__(self, depths, embed_dims, drop_path_rate=0.0, out_features=None):
        if drop_path_rate != 0.0:
            raise NotImplementedError("drop_path > 0 is not used by the reference wrappers (convnext_wrapper.py:97)")
        super().__init__(depths, embed_dims)
        self._out_features = out_features
        self._out_feature_strides = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
        self._out_feature_channels = {f"res{i + 2}": embed_dims[i] for i in range(4)}
