"""MobileNetV3 student backbone on the MI355X kernels (NHWC inside).

The reference's D2MobileNetV3 (/root/reference/models/backbones/mobilenet_v3.py:10-63) IS torchvision's ``MobileNetV3``
(third-party, not vendored, absent from this image) built from ``_mobilenet_v3_conf(arch, width_mult, dilated)`` with
``num_classes=1``; its forward walks ``self.features`` and taps four indices (small: 2 / 4 / 9 / 12, large: 4 / 7 / 13 / 16).
This file restates torchvision's published architecture (torchvision/models/mobilenetv3.py, unchanged 0.13-0.20): the
configuration tables, ``_make_divisible`` channel rounding, ``Conv2dNormActivation`` = conv (no bias, padding
(k - 1) // 2 * dilation) + BatchNorm2d(eps 1e-3, momentum 0.01) + {ReLU, Hardswish, none}, ``InvertedResidual`` = expand 1x1 ->
depthwise k x k (k 3 or 5; stride 1 when dilated) -> SqueezeExcitation(squeeze = make_divisible(expanded // 4, 8), ReLU,
Hardsigmoid) -> linear 1x1 projection, + input when stride 1 and equal widths.  Parameter names and order are torchvision's
(``features.i.0.weight``, ``features.i.1.*``, ``features.i.block.j.{0,1}.*``, ``features.i.block.j.fc{1,2}.*``,
``classifier.{0,3}.*``), so the reference's checkpoint remap (train/distillation_module.py:419-421) loads unchanged.  The
classifier head is part of the state dict and of ``parameters()`` exactly as in the reference but never runs."""
import torch
import torch.nn as nn

from .. import functional as Fn
from .. import functional16 as Fn16

NONE, RELU, HSWISH = 0, 1, 4  # activation codes of the BatchNorm epilogue kernels (include/dkd_hip.h)
BN_EPS, BN_MOMENTUM = 1e-3, 0.01


def make_divisible(v, divisor=8, min_value=None):
    """torchvision.models._utils._make_divisible"""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def bneck_table(size, width_mult=1.0, dilated=False):
    """_mobilenet_v3_conf (reduced_tail False): rows (cin, kernel, expanded, cout, use_se, use_hs, stride, dilation)"""
    d = 2 if dilated else 1
    if size == "large":
        rows = [(16, 3, 16, 16, False, "RE", 1, 1), (16, 3, 64, 24, False, "RE", 2, 1), (24, 3, 72, 24, False, "RE", 1, 1),
                (24, 5, 72, 40, True, "RE", 2, 1), (40, 5, 120, 40, True, "RE", 1, 1), (40, 5, 120, 40, True, "RE", 1, 1),
                (40, 3, 240, 80, False, "HS", 2, 1), (80, 3, 200, 80, False, "HS", 1, 1), (80, 3, 184, 80, False, "HS", 1, 1),
                (80, 3, 184, 80, False, "HS", 1, 1), (80, 3, 480, 112, True, "HS", 1, 1), (112, 3, 672, 112, True, "HS", 1, 1),
                (112, 5, 672, 160, True, "HS", 2, d), (160, 5, 960, 160, True, "HS", 1, d), (160, 5, 960, 160, True, "HS", 1, d)]
        last_channel = 1280
    elif size == "small":
        rows = [(16, 3, 16, 16, True, "RE", 2, 1), (16, 3, 72, 24, False, "RE", 2, 1), (24, 3, 88, 24, False, "RE", 1, 1),
                (24, 5, 96, 40, True, "HS", 2, 1), (40, 5, 240, 40, True, "HS", 1, 1), (40, 5, 240, 40, True, "HS", 1, 1),
                (40, 5, 120, 48, True, "HS", 1, 1), (48, 5, 144, 48, True, "HS", 1, 1), (48, 5, 288, 96, True, "HS", 2, d),
                (96, 5, 576, 96, True, "HS", 1, d), (96, 5, 576, 96, True, "HS", 1, d)]
        last_channel = 1024
    else:
        raise ValueError(f"MobileNetv3 can only be small or large. Size is {size}")
    adj = lambda c: make_divisible(c * width_mult, 8)
    return [(adj(ci), k, adj(ce), adj(co), se, act == "HS", s, dl) for ci, k, ce, co, se, act, s, dl in rows], adj(last_channel)


class _BatchNorm(nn.BatchNorm2d):
    def __init__(self, c):
        super().__init__(c, eps=BN_EPS, momentum=BN_MOMENTUM)


class ConvNormActivation(nn.Sequential):
    """torchvision.ops.Conv2dNormActivation: children ``0`` (conv weight) and ``1`` (BatchNorm); the activation has no state"""

    def __init__(self, cin, cout, kernel=1, stride=1, groups=1, act=NONE, dilation=1):
        conv = nn.Conv2d(cin, cout, kernel, stride, (kernel - 1) // 2 * dilation, dilation, groups, bias=False)
        super().__init__(conv, _BatchNorm(cout))
        self.act, self.depthwise = act, groups > 1
        if self.depthwise and not (groups == cin == cout):
            raise NotImplementedError("grouped convolutions other than depthwise")
        self.out_channels = cout

    def forward(self, x, src_nchw=False):
        conv, bn = self[0], self[1]
        if self.training:
            bn.num_batches_tracked += 1
        state = (bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        a16 = getattr(self, "autocast16", False)
        if self.depthwise:
            return (Fn16.DepthwiseBN16Fn if a16 else Fn.DepthwiseBNFn).apply(x, conv.weight, bn.weight, bn.bias, state, conv.stride[0],
                                                                           self.training, self.act, conv.dilation[0])
        return (Fn16.ConvBNAct16Fn if a16 else Fn.ConvBNActFn).apply(x, conv.weight, bn.weight, bn.bias, state, conv.kernel_size[0],
                                                                    conv.stride[0], self.act, self.training, src_nchw)


class SqueezeExcitation(nn.Module):
    """torchvision.ops.SqueezeExcitation(input, squeeze, activation=ReLU, scale_activation=Hardsigmoid)"""

    def __init__(self, channels, squeeze):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)

    def forward(self, x):
        fn = Fn16.SqueezeExcite16Fn if getattr(self, "autocast16", False) else Fn.SqueezeExciteFn
        return fn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)


class InvertedResidual(nn.Module):
    def __init__(self, cin, kernel, expanded, cout, use_se, use_hs, stride, dilation):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        act = HSWISH if use_hs else RELU
        layers = []
        if expanded != cin:
            layers.append(ConvNormActivation(cin, expanded, 1, act=act))
        layers.append(ConvNormActivation(expanded, expanded, kernel, 1 if dilation > 1 else stride, groups=expanded, act=act,
                                         dilation=dilation))
        if use_se:
            layers.append(SqueezeExcitation(expanded, make_divisible(expanded // 4, 8)))
        layers.append(ConvNormActivation(expanded, cout, 1, act=NONE))
        self.block = nn.Sequential(*layers)
        self.out_channels = cout

    def forward(self, x):
        y = self.block(x)
        return (Fn16.Add16Fn if getattr(self, "autocast16", False) else Fn.AddFn).apply(y, x) if self.use_res_connect else y


class D2MobileNetV3(nn.Module):
    """models/backbones/mobilenet_v3.py:10-63"""

    TAPS = {"small": {2: "res2", 4: "res3", 9: "res4", 12: "res5"}, "large": {4: "res2", 7: "res3", 13: "res4", 16: "res5"}}

    def __init__(self, size="small", width_mult=1.0, dilated=False, out_features=("res2", "res3", "res4", "res5")):
        super().__init__()
        table, last_channel = bneck_table(size, width_mult, dilated)
        layers = [ConvNormActivation(3, table[0][0], 3, 2, act=HSWISH)]
        layers += [InvertedResidual(*row) for row in table]
        layers.append(ConvNormActivation(table[-1][3], 6 * table[-1][3], 1, act=HSWISH))
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(1)  # (stateless; kept for the module tree of the reference)
        self.classifier = nn.Sequential(nn.Linear(6 * table[-1][3], last_channel), nn.Hardswish(), nn.Dropout(0.2),
                                        nn.Linear(last_channel, 1))  # num_classes = 1 (mobilenet_v3.py:27); never runs
        for m in self.modules():  # torchvision MobileNetV3.__init__
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)
        self.layer2res = dict(self.TAPS[size])
        self._out_feature_channels = {name: self.features[i].out_channels for i, name in self.layer2res.items()}
        self._out_feature_strides = ({"res2": 4, "res3": 8, "res4": 8, "res5": 8} if dilated else
                                     {"res2": 4, "res3": 8, "res4": 16, "res5": 32})  # mobilenet_v3.py:50-53 (metadata as is)
        self._out_features = out_features

    def set_precision(self, precision):
        """"fp32" (exact) or "bf16_autocast": the reference's default precision=16 policy (train.py:263) - dense and depthwise
        convolutions on bf16 operands with 16-bit results, BatchNorm / ReLU / Hardswish / squeeze-excitation / the residual add on
        16-bit values held in fp32 tensors (functional16.py)"""
        if precision not in ("fp32", "bf16_autocast"):
            raise ValueError(f"unknown student precision {precision!r}")
        self.precision = precision
        for m in self.modules():
            m.autocast16 = precision == "bf16_autocast"
        return self

    def forward_features_nhwc(self, x):
        outs = {}
        a16 = getattr(self, "autocast16", False)
        for idx, layer in enumerate(self.features):
            x = layer(x, src_nchw=True) if idx == 0 else layer(x)
            if idx in self.layer2res:
                outs[self.layer2res[idx]] = Fn16.RoundGradFn.apply(x) if a16 else x
        return outs

    def forward(self, x):
        return {k: v.permute(0, 3, 1, 2) for k, v in self.forward_features_nhwc(x).items()}
