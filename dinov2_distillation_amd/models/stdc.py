"""STDC student backbone on MI355X kernels (NHWC inside), state-dict compatible with
/root/reference/models/backbones/stdc.py (``features.{i}.conv/bn``, ``features.{i}.conv_list.{j}.conv/bn``,
``features.{i}.avd_layer.{0,1}``; the stage aliases ``x2/x4/x8/x16/x32`` share the same modules, so the duplicated keys of
the reference's state dict are reproduced).  CatBottleneck path only (``block_type='cat'``, ``use_conv_last=False``)."""
import math

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import functional16 as Fn16


class ConvX(nn.Module):
    """conv (no bias) + BatchNorm2d + ReLU (stdc.py:8-24); nn layers are parameter containers only"""

    def __init__(self, cin, cout, kernel=3, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=kernel, stride=stride, padding=kernel // 2, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.kernel, self.stride = kernel, stride

    def forward(self, x, src_nchw=False):
        if self.training:
            self.bn.num_batches_tracked += 1
        fn = Fn16.ConvBNAct16Fn if getattr(self, "autocast16", False) else Fn.ConvBNActFn
        return fn.apply(x, self.conv.weight, self.bn.weight, self.bn.bias,
                                    (self.bn.running_mean, self.bn.running_var), self.kernel, self.stride, True,
                                    self.training, src_nchw)


class CatBottleneck(nn.Module):
    def __init__(self, cin, cout, block_num=4, stride=1):
        super().__init__()
        assert block_num > 1
        self.stride = stride
        self.conv_list = nn.ModuleList()
        if stride == 2:
            self.avd_layer = nn.Sequential(
                nn.Conv2d(cout // 2, cout // 2, kernel_size=3, stride=2, padding=1, groups=cout // 2, bias=False),
                nn.BatchNorm2d(cout // 2))
        for idx in range(block_num):  # stdc.py:129-151
            if idx == 0:
                self.conv_list.append(ConvX(cin, cout // 2, kernel=1))
            elif idx == 1 and block_num == 2:
                self.conv_list.append(ConvX(cout // 2, cout // 2, stride=1))
            elif idx == 1 and block_num > 2:
                self.conv_list.append(ConvX(cout // 2, cout // 4, stride=1))
            elif idx < block_num - 1:
                self.conv_list.append(ConvX(cout // int(math.pow(2, idx)), cout // int(math.pow(2, idx + 1))))
            else:
                self.conv_list.append(ConvX(cout // int(math.pow(2, idx)), cout // int(math.pow(2, idx))))

    def forward(self, x):  # NHWC; stdc.py:153-172
        out1 = self.conv_list[0](x)
        outs, out = [], out1
        for idx, conv in enumerate(self.conv_list[1:]):
            if idx == 0 and self.stride == 2:
                bn = self.avd_layer[1]
                if self.training:
                    bn.num_batches_tracked += 1
                dw = Fn16.DepthwiseBN16Fn if getattr(self, "autocast16", False) else Fn.DepthwiseBNFn
                out = conv(dw.apply(out1, self.avd_layer[0].weight, bn.weight, bn.bias, (bn.running_mean, bn.running_var), 2,
                                    self.training, 0, 1))
            else:
                out = conv(out)
            outs.append(out)
        if self.stride == 2:
            out1 = (Fn16.AvgPool3_16Fn if getattr(self, "autocast16", False) else Fn.AvgPool3Fn).apply(out1, 2)
        return Fn.ConcatChannelsFn.apply(out1, *outs)


class STDCNet(nn.Module):
    def __init__(self, base=64, layers=(2, 2, 2), block_num=4, block_type="cat", use_conv_last=False):
        super().__init__()
        if block_type != "cat" or use_conv_last:
            raise NotImplementedError("only the CatBottleneck / no conv_last configuration the reference wrapper builds")
        layers = list(layers)
        feats = [ConvX(3, base // 2, 3, 2), ConvX(base // 2, base, 3, 2)]
        for i, n in enumerate(layers):
            for j in range(n):
                if i == 0 and j == 0:
                    feats.append(CatBottleneck(base, base * 4, block_num, 2))
                elif j == 0:
                    feats.append(CatBottleneck(base * 2 ** (i + 1), base * 2 ** (i + 2), block_num, 2))
                else:
                    feats.append(CatBottleneck(base * 2 ** (i + 2), base * 2 ** (i + 2), block_num, 1))
        self.features = nn.Sequential(*feats)
        self.layers = layers
        b = [2, 2 + layers[0], 2 + layers[0] + layers[1]]
        self.x2 = nn.Sequential(self.features[:1])   # aliases: same modules, duplicated state-dict keys (stdc.py:194-205)
        self.x4 = nn.Sequential(self.features[1:2])
        self.x8 = nn.Sequential(self.features[2:b[1]])
        self.x16 = nn.Sequential(self.features[b[1]:b[2]])
        self.x32 = nn.Sequential(self.features[b[2]:])
        self._bounds = b

    def init_params(self):
        """stdc.py:210-222"""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def set_precision(self, precision):
        """"fp32" (exact) or "bf16_autocast": the backbone under the reference's default precision=16 policy (functional16:
        ConvBNAct16Fn, DepthwiseBN16Fn, AvgPool3_16Fn; concatenation is exact)"""
        if precision not in ("fp32", "bf16_autocast"):
            raise ValueError(f"unknown student precision {precision!r}")
        self.precision = precision
        for m in self.modules():
            m.autocast16 = precision == "bf16_autocast"
        return self

    def forward_features_nhwc(self, x):
        a16 = getattr(self, "autocast16", False)
        out_ = (lambda t: Fn16.RoundGradFn.apply(t)) if a16 else (lambda t: t)
        x = self.features[0](x, src_nchw=True)
        x = self.features[1](x)
        outs = {"res2": out_(x)}
        b = self._bounds + [len(self.features)]
        for s in range(3):
            for i in range(b[s], b[s + 1]):
                x = self.features[i](x)
            outs[f"res{s + 3}"] = out_(x)
        return outs

    def forward(self, x):
        return {k: v.permute(0, 3, 1, 2) for k, v in self.forward_features_nhwc(x).items()}
