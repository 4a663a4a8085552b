"""Oracle (TEST INFRASTRUCTURE): MobileNetV2 student, CPU fp32 restatement.  PINNED by tests/golden/mobilenet_v2_small.npz
(generated from the reference's own D2MobileNetV2).  Restates /root/reference/models/backbones/mobilenet_v2.py:258-289
(Conv2d wrapper: conv -> norm -> activation), :292-372 (InvertedResidual; ReLU6 after ALL of its convs, the projecting 1x1
included) and :504-517 (forward) functionally; parameter names are the reference's state-dict keys."""
import torch
import torch.nn.functional as F

ARCH = [[1, 16, 1], [6, 24, 2], [6, 32, 3], [6, 64, 4], [6, 96, 3], [6, 160, 3], [6, 320, 1]]
LAYER_TO_RES = {2: "res2", 3: "res3", 5: "res4", 7: "res5"}


def conv_norm_act(p, pre, x, stride=1, groups=1, training=True, record=None, dilation=1):
    """record: optional list that receives the pre-activation (tests use it to measure the distance to the ReLU6 kinks);
    the depthwise conv is padded by its dilation (mobilenet_v2.py:344-345), the 1x1 convs are not padded"""
    w = p[pre + "weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=(w.shape[-1] // 2) * dilation, dilation=dilation, groups=groups)
    y = F.batch_norm(y, p[pre + "norm.running_mean"], p[pre + "norm.running_var"], p[pre + "norm.weight"],
                     p[pre + "norm.bias"], training, 0.1, 1e-5)
    if record is not None:
        record.append(y.detach())
    return F.relu6(y)


def inverted_residual(p, pre, x, stride, expand, training=True, record=None, dilation=1):
    y, k = x, 0
    if expand != 1:
        y = conv_norm_act(p, f"{pre}conv.0.", y, 1, 1, training, record)
        k = 1
    y = conv_norm_act(p, f"{pre}conv.{k}.", y, stride, y.shape[1], training, record, dilation)
    y = conv_norm_act(p, f"{pre}conv.{k + 1}.", y, 1, 1, training, record)
    return x + y if (stride == 1 and x.shape[1] == y.shape[1]) else y


def forward(p, x, strides=(1, 2, 2, 2, 1, 2, 1), training=True, dilations=(1, 1, 1, 1, 1, 1, 1)):
    x = conv_norm_act(p, "conv1.", x, 2, 1, training)
    outs = {}
    for i, ((t, c, n), s, d) in enumerate(zip(ARCH, strides, dilations)):
        for j in range(n):
            x = inverted_residual(p, f"layer{i + 1}.{j}.", x, s if j == 0 else 1, t, training, None, d if j == 0 else 1)
        if i + 1 in LAYER_TO_RES:
            outs[LAYER_TO_RES[i + 1]] = x
    return outs
