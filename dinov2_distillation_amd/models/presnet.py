"""PResNet student backbone (the RT-DETR ResNet-vd: three 3x3 stem convs, average-pool "d" shortcuts, stride in the 3x3) on
the MI355X kernels (NHWC inside), state-dict compatible with /root/reference/models/backbones/presnet.py:
``conv1.conv1_{1,2,3}.{conv.weight,norm.*}``, ``res_layers.{i}.blocks.{j}.branch2{a,b,c}.{conv.weight,norm.*}``,
``res_layers.{i}.blocks.{j}.short.{conv.weight,norm.*}`` (stage 2) / ``...short.conv.{conv.weight,norm.*}`` (after the pool).
BatchNorm is live (``freeze_norm=False``, presnet_wrapper.py:76-84), activation ReLU."""
import math

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import functional16 as Fn16

BLOCKS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}  # presnet.py:186-191


class _ConvWeight(nn.Module):
    """parameter holder named like ``nn.Conv2d(..., bias=False)``: default PyTorch init (kaiming_uniform, a=sqrt(5))"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class ConvNormLayer(nn.Module):
    """presnet.py:200-217: conv (no bias, pad (k-1)//2) -> BatchNorm2d -> optional ReLU, one fused launch chain"""

    def __init__(self, cin, cout, kernel, stride, act=False):
        super().__init__()
        self.conv = _ConvWeight(cin, cout, kernel)
        self.norm = nn.BatchNorm2d(cout)
        self.kernel, self.stride, self.relu = kernel, stride, act

    def forward(self, x, src_nchw=False):
        if self.training:
            self.norm.num_batches_tracked += 1
        fn = Fn16.ConvBNAct16Fn if getattr(self, "autocast16", False) else Fn.ConvBNActFn
        return fn.apply(x, self.conv.weight, self.norm.weight, self.norm.bias,
                        (self.norm.running_mean, self.norm.running_var), self.kernel, self.stride, self.relu,
                        self.training, src_nchw)


class _PoolShort(nn.Module):
    """variant "d", stride 2: AvgPool2d(2, 2, 0, ceil_mode=True) then a 1x1 ConvNormLayer (presnet.py:223-232, 264-276)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = ConvNormLayer(cin, cout, 1, 1)

    def forward(self, x):
        return self.conv((Fn16.AvgPool2_16Fn if getattr(self, "autocast16", False) else Fn.AvgPool2Fn).apply(x))


def _shortcut(cin, cout, stride):
    return _PoolShort(cin, cout) if stride == 2 else ConvNormLayer(cin, cout, 1, stride)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, cout, stride, shortcut):
        super().__init__()
        self.shortcut = shortcut
        if not shortcut:
            self.short = _shortcut(cin, cout, stride)
        self.branch2a = ConvNormLayer(cin, cout, 3, stride, act=True)
        self.branch2b = ConvNormLayer(cout, cout, 3, 1)

    def forward(self, x):
        out = self.branch2b(self.branch2a(x))
        return (Fn16.AddRelu16Fn if getattr(self, "autocast16", False) else Fn.AddReluFn).apply(out, x if self.shortcut else self.short(x))


class BottleNeck(nn.Module):
    expansion = 4

    def __init__(self, cin, cout, stride, shortcut):
        super().__init__()
        self.branch2a = ConvNormLayer(cin, cout, 1, 1, act=True)       # variants b/d: the stride sits in the 3x3
        self.branch2b = ConvNormLayer(cout, cout, 3, stride, act=True)
        self.branch2c = ConvNormLayer(cout, cout * 4, 1, 1)
        self.shortcut = shortcut
        if not shortcut:
            self.short = _shortcut(cin, cout * 4, stride)

    def forward(self, x):
        out = self.branch2c(self.branch2b(self.branch2a(x)))
        return (Fn16.AddRelu16Fn if getattr(self, "autocast16", False) else Fn.AddReluFn).apply(out, x if self.shortcut else self.short(x))


class Blocks(nn.Module):
    def __init__(self, block, cin, cout, count, stage_num):
        super().__init__()
        self.blocks = nn.ModuleList()
        for i in range(count):
            self.blocks.append(block(cin, cout, 2 if i == 0 and stage_num != 2 else 1, i != 0))
            if i == 0:
                cin = cout * block.expansion

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x


class D2Presnet(nn.Module):
    """presnet.py:302-476 with variant "d", act "relu", live BatchNorm, nothing frozen"""

    def __init__(self, depth=50, out_features=None, stem_width=64, widths=(64, 128, 256, 512), blocks=None):
        super().__init__()
        nums = list(blocks) if blocks is not None else BLOCKS[depth]
        block = BottleNeck if depth >= 50 else BasicBlock
        c = stem_width
        self.conv1 = nn.Sequential()
        for name, (ci, co, s) in zip(("conv1_1", "conv1_2", "conv1_3"), ((3, c // 2, 2), (c // 2, c // 2, 1), (c // 2, c, 1))):
            self.conv1.add_module(name, ConvNormLayer(ci, co, 3, s, act=True))
        self.res_layers = nn.ModuleList()
        cin = c
        for i, w in enumerate(widths):
            self.res_layers.append(Blocks(block, cin, w, nums[i], i + 2))
            cin = w * block.expansion
        names = ["res2", "res3", "res4", "res5"]
        self._out_features = out_features or names
        self._out_feature_channels = {n: w * block.expansion for n, w in zip(names, widths)}
        self._out_feature_strides = dict(zip(names, (4, 8, 16, 32)))

    def set_precision(self, precision):
        """"fp32" (exact) or "bf16_autocast": the reference's default precision=16 policy (train.py:263) - convolutions on bf16 operands
        with 16-bit results, BatchNorm / ReLU / pooling / the residual add on 16-bit values held in fp32 tensors (the form
        functional16.ConvBNAct16Fn pins against torch.autocast for ResNet and STDC)"""
        if precision not in ("fp32", "bf16_autocast"):
            raise ValueError(f"unknown student precision {precision!r}")
        self.precision = precision
        for m in self.modules():
            m.autocast16 = precision == "bf16_autocast"
        return self

    def forward_features_nhwc(self, x):
        x = self.conv1.conv1_1(x, src_nchw=True)
        x = self.conv1.conv1_3(self.conv1.conv1_2(x))
        x = Fn.MaxPool3Fn.apply(x)
        a16 = getattr(self, "autocast16", False)
        outs = {}
        for i, stage in enumerate(self.res_layers):
            x = stage(x)
            outs[f"res{i + 2}"] = Fn16.RoundGradFn.apply(x) if a16 else x
        return outs

    def forward(self, x):
        return {k: v.permute(0, 3, 1, 2) for k, v in self.forward_features_nhwc(x).items()}
