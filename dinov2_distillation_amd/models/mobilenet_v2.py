"""MobileNetV2 student backbone on the MI355X kernels (NHWC inside), state-dict compatible with
/root/reference/models/backbones/mobilenet_v2.py: ``conv1.{weight,norm.*}``, ``layer{i}.{j}.conv.{k}.{weight,norm.*}``.
As in the reference every conv of an InvertedResidual - including the projecting 1x1 - is followed by BatchNorm and ReLU6
(mobilenet_v2.py:326-366 passes ``activation`` to all three), BatchNorm is live, no bias anywhere.  The dilated os8 / os16
configurations use the same depthwise kernel (padding = dilation, first block of a layer only, mobilenet_v2.py:489-497)."""
import math

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import functional16 as Fn16

ARCH = [[1, 16, 1], [6, 24, 2], [6, 32, 3], [6, 64, 4], [6, 96, 3], [6, 160, 3], [6, 320, 1]]  # mobilenet_v2.py:391-399
LAYER_TO_RES = {"layer2": "res2", "layer3": "res3", "layer5": "res4", "layer7": "res5"}        # :439-444
RELU6 = 2


class ConvNormAct(nn.Module):
    """the reference's ``Conv2d(..., bias="", norm=BatchNorm2d, activation=relu6)``: parameter ``weight`` + ``norm``"""

    def __init__(self, cin, cout, kernel, stride=1, depthwise=False, dilation=1):
        super().__init__()
        self.dilation = dilation
        self.weight = nn.Parameter(torch.empty(cout, 1 if depthwise else cin, kernel, kernel))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))  # nn.Conv2d default
        self.norm = nn.BatchNorm2d(cout)
        self.kernel, self.stride, self.depthwise = kernel, stride, depthwise

    def forward(self, x, src_nchw=False):
        if self.training:
            self.norm.num_batches_tracked += 1
        state = (self.norm.running_mean, self.norm.running_var)
        a16 = getattr(self, "autocast16", False)
        if self.depthwise:
            return (Fn16.DepthwiseBN16Fn if a16 else Fn.DepthwiseBNFn).apply(x, self.weight, self.norm.weight, self.norm.bias, state,
                                                                           self.stride, self.training, RELU6, self.dilation)
        return (Fn16.ConvBNAct16Fn if a16 else Fn.ConvBNActFn).apply(x, self.weight, self.norm.weight, self.norm.bias, state,
                                                                    self.kernel, self.stride, RELU6, self.training, src_nchw)


class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, expand_ratio, dilation=1):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        hidden = int(round(cin * expand_ratio))
        layers = []
        if expand_ratio != 1:
            layers.append(ConvNormAct(cin, hidden, 1))
        layers += [ConvNormAct(hidden, hidden, 3, stride, depthwise=True, dilation=dilation), ConvNormAct(hidden, cout, 1)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        y = self.conv(x)
        return (Fn16.Add16Fn if getattr(self, "autocast16", False) else Fn.AddFn).apply(x, y) if self.use_res_connect else y


class D2MobileNetV2(nn.Module):
    def __init__(self, widen_factor=1.0, strides=(1, 2, 2, 2, 1, 2, 1), dilations=(1, 1, 1, 1, 1, 1, 1),
                 out_features=("res2", "res3", "res4", "res5")):
        super().__init__()
        cin = int(32 * widen_factor)
        self.conv1 = ConvNormAct(3, cin, 3, 2)
        self._out_features = list(out_features)
        self._out_feature_channels, self._out_feature_strides = {}, {}
        self.layers = []
        tot = 1  # as the reference: the stem's stride is not counted in _out_feature_strides (mobilenet_v2.py:446)
        for i, ((t, c, n), s, d) in enumerate(zip(ARCH, strides, dilations)):
            cout = int(c * widen_factor)
            blocks = []
            for j in range(n):
                blocks.append(InvertedResidual(cin, cout, s if j == 0 else 1, t, d if j == 0 else 1))
                cin = cout
            tot *= s
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*blocks))
            self.layers.append(name)
            if name in LAYER_TO_RES:
                self._out_feature_channels[LAYER_TO_RES[name]] = cout
                self._out_feature_strides[LAYER_TO_RES[name]] = tot

    def set_precision(self, precision):
        """"fp32" (exact) or "bf16_autocast": the reference's default precision=16 policy (train.py:263) - dense and depthwise
        convolutions on bf16 operands with 16-bit results, BatchNorm / ReLU6 / the residual add on 16-bit values held in fp32 tensors"""
        if precision not in ("fp32", "bf16_autocast"):
            raise ValueError(f"unknown student precision {precision!r}")
        self.precision = precision
        for m in self.modules():
            m.autocast16 = precision == "bf16_autocast"
        return self

    def forward_features_nhwc(self, x):
        x = self.conv1(x, src_nchw=True)
        a16 = getattr(self, "autocast16", False)
        outs = {}
        for name in self.layers:
            x = getattr(self, name)(x)
            if name in LAYER_TO_RES and LAYER_TO_RES[name] in self._out_features:
                outs[LAYER_TO_RES[name]] = Fn16.RoundGradFn.apply(x) if a16 else x
        return outs

    def forward(self, x):
        return {k: v.permute(0, 3, 1, 2) for k, v in self.forward_features_nhwc(x).items()}
