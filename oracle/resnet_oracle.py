"""Oracle (TEST INFRASTRUCTURE): ResNet student (detectron2-style bottleneck, stride in the 3x3, live BatchNorm), CPU fp32
restatement.  PINNED by tests/golden/resnet_small.npz (generated from the reference's own BasicStem / BottleneckBlock /
ResNet).  Restates /root/reference/models/backbones/resnet.py:402-544,625-650 functionally; parameter names are the
reference's state-dict keys."""
import torch
import torch.nn.functional as F


def conv_norm(p, pre, x, stride=1, relu=False, training=True):
    """``Conv2d`` with norm (resnet.py:258-289): conv (no bias, pad k//2) -> BatchNorm2d (train) [-> relu by the caller]"""
    w = p[pre + "weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    y = F.batch_norm(y, p[pre + "norm.running_mean"], p[pre + "norm.running_var"], p[pre + "norm.weight"],
                     p[pre + "norm.bias"], training, 0.1, 1e-5)
    return torch.relu(y) if relu else y


def bottleneck(p, pre, x, stride):
    """``BottleneckBlock.forward`` (resnet.py:496-512), stride_in_1x1=False"""
    out = conv_norm(p, pre + "conv1.", x, 1, True)
    out = conv_norm(p, pre + "conv2.", out, stride, True)
    out = conv_norm(p, pre + "conv3.", out, 1, False)
    sc = conv_norm(p, pre + "shortcut.", x, stride, False) if (pre + "shortcut.weight") in p else x
    return torch.relu(out + sc)


def forward(p, x, blocks_per_stage, strides=(1, 2, 2, 2)):
    """``BasicStem.forward`` (resnet.py:540-544) + ``ResNet.forward`` (:625-650)"""
    x = conv_norm(p, "stem.conv1.", x, 2, True)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    for i, n in enumerate(blocks_per_stage):
        for j in range(n):
            x = bottleneck(p, f"res{i + 2}.{j}.", x, strides[i] if j == 0 else 1)
        outs[f"res{i + 2}"] = x
    return outs
