"""Oracle (TEST INFRASTRUCTURE): MobileNetV3 student, CPU restatement on torch's functional ops.

PARITY UNPINNED: the reference's D2MobileNetV3 (/root/reference/models/backbones/mobilenet_v3.py:10-63) subclasses
torchvision's ``MobileNetV3`` (third-party, no version pinned by the reference, not vendored under /root/reference and not
installed in this image), so neither a fixture generated from the reference nor the class itself is available.  What pins this
file instead: torchvision's published parameter counts (mobilenet_v3_small 2 542 856 / mobilenet_v3_large 5 483 032 at 1000
classes -> 1 518 881 / 4 203 313 with the reference's num_classes=1), its documented state-dict key layout, and the tap indices
and channel widths the reference itself reads (mobilenet_v3.py:31-48: features[2|4|9|12] / features[4|7|13|16]).

Restates torchvision/models/mobilenetv3.py (0.13-0.20, unchanged): ``_mobilenet_v3_conf`` tables, ``InvertedResidualConfig.
adjust_channels`` (``_make_divisible(c * width_mult, 8)``), ``InvertedResidual`` (expand 1x1 only when expanded != input;
depthwise stride forced to 1 when dilation > 1; SqueezeExcitation(squeeze = _make_divisible(expanded // 4, 8)) with ReLU and
Hardsigmoid; linear projection; residual when stride == 1 and cin == cout), ``MobileNetV3.__init__`` (stem 3x3 / 2 Hardswish,
last 1x1 conv to 6 x cout Hardswish, BatchNorm2d(eps=0.001, momentum=0.01)) and the reference's forward (mobilenet_v3.py:56-63:
walk ``features``, tap four indices).  Functional on a state dict with torchvision's key names."""
import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-3, 0.01
TAPS = {"small": {2: "res2", 4: "res3", 9: "res4", 12: "res5"}, "large": {4: "res2", 7: "res3", 13: "res4", 16: "res5"}}


def _make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def conf(size, dilated=False):
    """rows: (input, kernel, expanded, out, use_se, activation, stride, dilation) at width_mult 1"""
    d = 2 if dilated else 1
    if size == "large":
        return [(16, 3, 16, 16, False, "RE", 1, 1), (16, 3, 64, 24, False, "RE", 2, 1), (24, 3, 72, 24, False, "RE", 1, 1),
                (24, 5, 72, 40, True, "RE", 2, 1), (40, 5, 120, 40, True, "RE", 1, 1), (40, 5, 120, 40, True, "RE", 1, 1),
                (40, 3, 240, 80, False, "HS", 2, 1), (80, 3, 200, 80, False, "HS", 1, 1), (80, 3, 184, 80, False, "HS", 1, 1),
                (80, 3, 184, 80, False, "HS", 1, 1), (80, 3, 480, 112, True, "HS", 1, 1), (112, 3, 672, 112, True, "HS", 1, 1),
                (112, 5, 672, 160, True, "HS", 2, d), (160, 5, 960, 160, True, "HS", 1, d), (160, 5, 960, 160, True, "HS", 1, d)]
    return [(16, 3, 16, 16, True, "RE", 2, 1), (16, 3, 72, 24, False, "RE", 2, 1), (24, 3, 88, 24, False, "RE", 1, 1),
            (24, 5, 96, 40, True, "HS", 2, 1), (40, 5, 240, 40, True, "HS", 1, 1), (40, 5, 240, 40, True, "HS", 1, 1),
            (40, 5, 120, 48, True, "HS", 1, 1), (48, 5, 144, 48, True, "HS", 1, 1), (48, 5, 288, 96, True, "HS", 2, d),
            (96, 5, 576, 96, True, "HS", 1, d), (96, 5, 576, 96, True, "HS", 1, d)]


def conv_norm_act(p, pre, x, act, stride=1, dilation=1, groups=1, training=True, record=None):
    """Conv2dNormActivation: ``{pre}0.weight`` + ``{pre}1.*``; padding (k - 1) // 2 * dilation.  record: list receiving the
    pre-activation (tests use it to keep clear of the ReLU / Hardswish kinks)"""
    w = p[pre + "0.weight"]
    y = F.conv2d(x, w, None, stride, (w.shape[-1] - 1) // 2 * dilation, dilation, groups)
    y = F.batch_norm(y, p[pre + "1.running_mean"], p[pre + "1.running_var"], p[pre + "1.weight"], p[pre + "1.bias"], training,
                     BN_MOMENTUM, BN_EPS)
    if record is not None and act is not None:
        record.append((act, y.detach()))
    return F.hardswish(y) if act == "HS" else F.relu(y) if act == "RE" else y


def squeeze_excitation(p, pre, x, record=None):
    s = F.adaptive_avg_pool2d(x, 1)
    s = F.conv2d(s, p[pre + "fc1.weight"], p[pre + "fc1.bias"])
    if record is not None:
        record.append(("RE", s.detach()))
    s = F.conv2d(F.relu(s), p[pre + "fc2.weight"], p[pre + "fc2.bias"])
    if record is not None:
        record.append(("HSIG", s.detach()))
    return F.hardsigmoid(s) * x


def inverted_residual(p, pre, x, row, training=True, record=None):
    cin, k, exp, cout, use_se, act, stride, dil = row
    y, j = x, 0
    if exp != cin:
        y = conv_norm_act(p, f"{pre}block.{j}.", y, act, training=training, record=record)
        j += 1
    y = conv_norm_act(p, f"{pre}block.{j}.", y, act, 1 if dil > 1 else stride, dil, exp, training, record)
    j += 1
    if use_se:
        y = squeeze_excitation(p, f"{pre}block.{j}.", y, record)
        j += 1
    y = conv_norm_act(p, f"{pre}block.{j}.", y, None, training=training)
    return y + x if (stride == 1 and cin == cout) else y


def forward(p, x, size="small", dilated=False, training=True):
    rows = conf(size, dilated)
    outs = {}
    x = conv_norm_act(p, "features.0.", x, "HS", 2, training=training)
    for i, row in enumerate(rows, start=1):
        x = inverted_residual(p, f"features.{i}.", x, row, training)
        if i in TAPS[size]:
            outs[TAPS[size][i]] = x
    last = len(rows) + 1
    x = conv_norm_act(p, f"features.{last}.", x, "HS", training=training)
    if last in TAPS[size]:
        outs[TAPS[size][last]] = x
    return outs


def state_shapes(size):
    """torchvision's state-dict layout for MobileNetV3(conf(size), last_channel, num_classes=1), in registration order"""
    rows = conf(size)
    shapes = {}

    def cna(pre, cin, cout, k, groups=1):
        shapes[pre + "0.weight"] = (cout, cin // groups, k, k)
        for n in ("weight", "bias", "running_mean", "running_var"):
            shapes[pre + "1." + n] = (cout,)
        shapes[pre + "1.num_batches_tracked"] = ()

    cna("features.0.", 3, rows[0][0], 3)
    for i, (cin, k, exp, cout, use_se, act, stride, dil) in enumerate(rows, start=1):
        j = 0
        if exp != cin:
            cna(f"features.{i}.block.{j}.", cin, exp, 1)
            j += 1
        cna(f"features.{i}.block.{j}.", exp, exp, k, exp)
        j += 1
        if use_se:
            sq = _make_divisible(exp // 4, 8)
            shapes[f"features.{i}.block.{j}.fc1.weight"] = (sq, exp, 1, 1)
            shapes[f"features.{i}.block.{j}.fc1.bias"] = (sq,)
            shapes[f"features.{i}.block.{j}.fc2.weight"] = (exp, sq, 1, 1)
            shapes[f"features.{i}.block.{j}.fc2.bias"] = (exp,)
            j += 1
        cna(f"features.{i}.block.{j}.", exp, cout, 1)
    last_in = rows[-1][3]
    cna(f"features.{len(rows) + 1}.", last_in, 6 * last_in, 1)
    last_channel = _make_divisible(1280 if size == "large" else 1024, 8)
    shapes["classifier.0.weight"], shapes["classifier.0.bias"] = (last_channel, 6 * last_in), (last_channel,)
    shapes["classifier.3.weight"], shapes["classifier.3.bias"] = (1, last_channel), (1,)
    return shapes
