"""ResNet student backbone (detectron2-style, stride in the 3x3) on MI355X kernels (NHWC inside), state-dict compatible
with /root/reference/models/backbones/resnet.py: ``stem.conv1.{weight,norm.*}``, ``res{2..5}.{i}.conv{1,2,3}.{weight,norm.*}``,
``res{k}.{i}.shortcut.{weight,norm.*}``.  BatchNorm is live (not frozen), train-mode statistics (resnet_wrapper.py:62-66)."""
import torch
import torch.nn as nn

from .. import functional as Fn
from .. import functional16 as Fn16


class ConvNorm(nn.Module):
    """the reference's ``Conv2d(..., bias=False, norm=BatchNorm2d)`` (resnet.py:236-289): parameters only"""

    def __init__(self, cin, cout, kernel, stride=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, kernel, kernel))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")  # fvcore c2_msra_fill (resnet.py:385,482,538)
        self.norm = nn.BatchNorm2d(cout)
        self.kernel, self.stride = kernel, stride

    def forward(self, x, relu, src_nchw=False):
        if self.training:
            self.norm.num_batches_tracked += 1
        fn = Fn16.ConvBNAct16Fn if getattr(self, "autocast16", False) else Fn.ConvBNActFn
        return fn.apply(x, self.weight, self.norm.weight, self.norm.bias,
                                    (self.norm.running_mean, self.norm.running_var), self.kernel, self.stride, relu,
                                    self.training, src_nchw)


class BasicStem(nn.Module):
    def __init__(self, in_channels=3, out_channels=64):
        super().__init__()
        self.conv1 = ConvNorm(in_channels, out_channels, 7, 2)
        self.out_channels, self.stride = out_channels, 4

    def forward(self, x):  # NCHW image -> NHWC
        y = Fn.MaxPool3Fn.apply(self.conv1(x, True, src_nchw=True))
        # under the policy the stem and its max pool keep their 16-bit values in fp32 (3-channel im2col, arg-max pooling); from here
        # on the activations are bf16 in HBM
        return Fn16.ToBf16Fn.apply(y) if getattr(self, "autocast16", False) else y


class BottleneckBlock(nn.Module):
    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1):
        super().__init__()
        self.shortcut = ConvNorm(in_channels, out_channels, 1, stride) if in_channels != out_channels else None
        self.conv1 = ConvNorm(in_channels, bottleneck_channels, 1, 1)        # stride_in_1x1=False (resnet.py:449)
        self.conv2 = ConvNorm(bottleneck_channels, bottleneck_channels, 3, stride)
        self.conv3 = ConvNorm(bottleneck_channels, out_channels, 1, 1)
        self.stride, self.out_channels = stride, out_channels

    def forward(self, x):
        out = self.conv3(self.conv2(self.conv1(x, True), True), False)
        sc = self.shortcut(x, False) if self.shortcut is not None else x
        return (Fn16.AddRelu16Fn if getattr(self, "autocast16", False) else Fn.AddReluFn).apply(out, sc)


class BasicBlock(nn.Module):
    def __init__(self, in_channels, out_channels, *, stride=1):
        super().__init__()
        self.shortcut = ConvNorm(in_channels, out_channels, 1, stride) if in_channels != out_channels else None
        self.conv1 = ConvNorm(in_channels, out_channels, 3, stride)
        self.conv2 = ConvNorm(out_channels, out_channels, 3, 1)
        self.stride, self.out_channels = stride, out_channels

    def forward(self, x):
        out = self.conv2(self.conv1(x, True), False)
        sc = self.shortcut(x, False) if self.shortcut is not None else x
        return (Fn16.AddRelu16Fn if getattr(self, "autocast16", False) else Fn.AddReluFn).apply(out, sc)


def make_stage(block_class, num_blocks, *, in_channels, out_channels, stride_per_block, **kwargs):
    blocks = []
    for i in range(num_blocks):
        blocks.append(block_class(in_channels, out_channels, stride=stride_per_block[i], **kwargs))
        in_channels = out_channels
    return blocks


def make_resnet_stages(depth):
    """resnet.py:835-900"""
    nb = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[depth]
    if depth < 50:
        cin, cout, cls = [64, 64, 128, 256], [64, 128, 256, 512], BasicBlock
    else:
        cin, cout, cls = [64, 256, 512, 1024], [256, 512, 1024, 2048], BottleneckBlock
    ret = []
    for n, s, i, o in zip(nb, [1, 2, 2, 2], cin, cout):
        kw = {"bottleneck_channels": o // 4} if depth >= 50 else {}
        ret.append(make_stage(cls, n, in_channels=i, out_channels=o, stride_per_block=[s] + [1] * (n - 1), **kw))
    return ret


class ResNet(nn.Module):
    def __init__(self, stem, stages, out_features=None):
        super().__init__()
        self.stem = stem
        if out_features is not None:  # stages beyond the deepest requested one are never built (resnet.py:576-585)
            n = max({"res2": 1, "res3": 2, "res4": 3, "res5": 4}.get(f, 0) for f in out_features)
            stages = stages[:n]
        self.stage_names = []
        self._out_feature_channels = {"stem": stem.out_channels}
        for i, blocks in enumerate(stages):
            name = f"res{i + 2}"
            self.add_module(name, nn.Sequential(*blocks))
            self.stage_names.append(name)
            self._out_feature_channels[name] = blocks[-1].out_channels
        self._out_features = out_features or [self.stage_names[-1]]

    def set_precision(self, precision):
        """"fp32" (exact) or "bf16_autocast": the backbone under the reference's default precision=16 policy (train.py:263):
        convolutions on bf16 operands with 16-bit results, BatchNorm / ReLU / pooling / the residual add on 16-bit values
        (functional16.ConvBNAct16Fn), bf16 in HBM behind the stem; features leave as fp32 holding those values"""
        if precision not in ("fp32", "bf16_autocast"):
            raise ValueError(f"unknown student precision {precision!r}")
        self.precision = precision
        for m in self.modules():
            m.autocast16 = precision == "bf16_autocast"
        return self

    def forward_features_nhwc(self, x):
        outs = {}
        x = self.stem(x)
        for name in self.stage_names:
            for blk in getattr(self, name):
                x = blk(x)
            if name in self._out_features:
                outs[name] = Fn16.ToFloatFn.apply(x) if getattr(self, "autocast16", False) else x
        return outs

    def forward(self, x):
        return {k: v.permute(0, 3, 1, 2) for k, v in self.forward_features_nhwc(x).items()}
