"""DarkNet (YOLOv8 CSP backbone) student on the MI355X kernels (NHWC inside), state-dict compatible with
/root/reference/models/backbones/darknet.py: ``p1.0.{conv.weight,bn.*}``, ``p{2..5}.0.{conv.weight,bn.*}``,
``p{2..5}.1.cv{1,2}.{conv.weight,bn.*}``, ``p{2..5}.1.m.{j}.cv{1,2}.{conv.weight,bn.*}``.
Every ``Conv`` is conv (no bias, pad k//2) -> BatchNorm2d(eps 1e-3, momentum 0.03) -> SiLU (darknet.py:44-64)."""
import math

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import functional16 as Fn16

ACT_SILU = 3  # dkd_bn_act_pos_fwd_f32 / dkd_bn_bwd_prep_f32 activation code

VERSIONS = {  # models/wrappers/darknet_wrapper.py:8-14: [C2f depths, widths]
    "n": [[1, 2, 2, 1], [3, 16, 32, 64, 128, 256]],
    "s": [[1, 2, 2, 1], [3, 32, 64, 128, 256, 512]],
    "m": [[2, 4, 4, 2], [3, 48, 96, 192, 384, 576]],
    "l": [[3, 6, 6, 3], [3, 64, 128, 256, 512, 512]],
    "x": [[3, 6, 6, 3], [3, 80, 160, 320, 640, 640]],
}


class _ConvWeight(nn.Module):
    """parameter holder named like ``nn.Conv2d(..., bias=False)``: default PyTorch init (kaiming_uniform, a=sqrt(5))"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class Conv(nn.Module):
    """darknet.py:44-64: one im2col GEMM + batch statistics + fused BN/SiLU launch"""

    def __init__(self, c1, c2, k=1, s=1):
        super().__init__()
        self.conv = _ConvWeight(c1, c2, k)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.kernel, self.stride = k, s

    def forward(self, x, src_nchw=False):
        if self.training:
            self.bn.num_batches_tracked += 1
        fn = Fn16.ConvBNAct16Fn if getattr(self, "autocast16", False) else Fn.ConvBNActFn
        return fn.apply(x, self.conv.weight, self.bn.weight, self.bn.bias,
                        (self.bn.running_mean, self.bn.running_var, self.bn.momentum, self.bn.eps),
                        self.kernel, self.stride, ACT_SILU, self.training, src_nchw)


class Bottleneck(nn.Module):
    """darknet.py:71-84 with e = 1.0, k = (3, 3), g = 1 (the only form C2f builds)"""

    def __init__(self, c1, c2, shortcut=True):
        super().__init__()
        self.cv1 = Conv(c1, c2, 3, 1)
        self.cv2 = Conv(c2, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return (Fn16.Add16Fn if getattr(self, "autocast16", False) else Fn.AddFn).apply(x, y) if self.add else y


class C2f(nn.Module):
    """darknet.py:87-105.  ``cv1(x).chunk(2, 1)`` keeps both halves in the concat, so the concat's first operand is cv1's whole
    output and only the second half is sliced out (for the bottleneck chain)."""

    def __init__(self, c1, c2, n=1, shortcut=False):
        super().__init__()
        self.c = int(c2 * 0.5)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut) for _ in range(n))

    def forward(self, x):
        t = self.cv1(x)
        ys = [t, Fn.SliceChannelsFn.apply(t, self.c, self.c)]
        for m in self.m:
            ys.append(m(ys[-1]))
        return self.cv2(Fn.ConcatChannelsFn.apply(ys[0], *ys[2:]))


class DarkNet(nn.Module):
    """darknet.py:114-166"""

    def __init__(self, depth, width):
        super().__init__()
        self.p1 = nn.Sequential(Conv(width[0], width[1], 3, 2))
        for i in range(4):
            setattr(self, f"p{i + 2}", nn.Sequential(Conv(width[i + 1], width[i + 2], 3, 2),
                                                     C2f(width[i + 2], width[i + 2], n=depth[i], shortcut=True)))
        self._out_features = ["res2", "res3", "res4", "res5"]
        self._out_feature_strides = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
        self._out_feature_channels = {f"res{i + 2}": width[i + 2] for i in range(4)}

    def set_precision(self, precision):
        """"fp32" (exact) or "bf16_autocast": the reference's default precision=16 policy (train.py:263) - convolutions on bf16 operands
        with 16-bit results, BatchNorm and SiLU on 16-bit tensors (each rounded), 16-bit residual adds; values held in fp32 tensors"""
        if precision not in ("fp32", "bf16_autocast"):
            raise ValueError(f"unknown student precision {precision!r}")
        self.precision = precision
        for m in self.modules():
            m.autocast16 = precision == "bf16_autocast"
        return self

    def forward_features_nhwc(self, x):
        x = self.p1[0](x, src_nchw=True)
        a16 = getattr(self, "autocast16", False)
        outs = {}
        for i in range(4):
            x = getattr(self, f"p{i + 2}")(x)
            outs[f"res{i + 2}"] = Fn16.RoundGradFn.apply(x) if a16 else x
        return outs

    def forward(self, x):
        return {k: v.permute(0, 3, 1, 2) for k, v in self.forward_features_nhwc(x).items()}
