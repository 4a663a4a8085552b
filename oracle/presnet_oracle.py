"""Oracle (TEST INFRASTRUCTURE): PResNet student (ResNet-vd: 3-conv stem, average-pool "d" shortcuts, live BatchNorm), CPU fp32
restatement.  PINNED by tests/golden/presnet_small.npz (generated from the reference's own D2Presnet).  Restates
/root/reference/models/backbones/presnet.py:200-300 (ConvNormLayer, BasicBlock, BottleNeck, Blocks) and :396-476
(PResNet.forward / D2Presnet.forward) functionally; parameter names are the reference's state-dict keys."""
import torch
import torch.nn.functional as F


def conv_norm(p, pre, x, stride=1, relu=False, training=True):
    """``ConvNormLayer.forward`` (presnet.py:216-217): conv (no bias, pad (k-1)//2) -> BatchNorm2d -> act"""
    w = p[pre + "conv.weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=(w.shape[-1] - 1) // 2)
    y = F.batch_norm(y, p[pre + "norm.running_mean"], p[pre + "norm.running_var"], p[pre + "norm.weight"],
                     p[pre + "norm.bias"], training, 0.1, 1e-5)
    return torch.relu(y) if relu else y


def shortcut(p, pre, x, stride, training=True):
    """variant "d": stride-2 shortcuts average-pool first (presnet.py:223-234, 264-278)"""
    if (pre + "short.conv.conv.weight") in p:
        return conv_norm(p, pre + "short.conv.", F.avg_pool2d(x, 2, 2, 0, ceil_mode=True), 1, False, training)
    return conv_norm(p, pre + "short.", x, stride, False, training)


def block(p, pre, x, stride, has_identity, bottleneck, training=True):
    """``BasicBlock.forward`` (presnet.py:238-249) / ``BottleNeck.forward`` (:281-293), stride in the 3x3"""
    if bottleneck:
        out = conv_norm(p, pre + "branch2a.", x, 1, True, training)
        out = conv_norm(p, pre + "branch2b.", out, stride, True, training)
        out = conv_norm(p, pre + "branch2c.", out, 1, False, training)
    else:
        out = conv_norm(p, pre + "branch2a.", x, stride, True, training)
        out = conv_norm(p, pre + "branch2b.", out, 1, False, training)
    return torch.relu(out + (x if has_identity else shortcut(p, pre, x, stride, training)))


def forward(p, x, blocks_per_stage, bottleneck, training=True):
    """``PResNet.forward`` (presnet.py:396-405) + the dict of ``D2Presnet.forward`` (:466-476)"""
    for i, s in ((1, 2), (2, 1), (3, 1)):
        x = conv_norm(p, f"conv1.conv1_{i}.", x, s, True, training)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    for i, n in enumerate(blocks_per_stage):
        for j in range(n):
            x = block(p, f"res_layers.{i}.blocks.{j}.", x, 2 if (j == 0 and i != 0) else 1, j != 0, bottleneck, training)
        outs[f"res{i + 2}"] = x
    return outs
