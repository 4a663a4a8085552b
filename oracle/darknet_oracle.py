"""Oracle (TEST INFRASTRUCTURE): DarkNet (YOLOv8 CSP backbone) student, CPU fp32 restatement.  PINNED by
tests/golden/darknet_n_small.npz (generated from the reference's own DarkNet).  Restates
/root/reference/models/backbones/darknet.py:44-64 (Conv), :71-84 (Bottleneck), :87-105 (C2f) and :114-166 (DarkNet)
functionally; parameter names are the reference's state-dict keys."""
import torch
import torch.nn.functional as F


def conv(p, pre, x, stride=1, training=True):
    """``Conv.forward`` (darknet.py:62-64): conv (no bias, pad k//2) -> BatchNorm2d(eps 1e-3, momentum 0.03) -> SiLU"""
    w = p[pre + "conv.weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    y = F.batch_norm(y, p[pre + "bn.running_mean"], p[pre + "bn.running_var"], p[pre + "bn.weight"], p[pre + "bn.bias"],
                     training, 0.03, 1e-3)
    return F.silu(y)   # nn.SiLU (darknet.py:47): ONE op - under autocast one rounding, not the two of y * sigmoid(y)


def bottleneck(p, pre, x, training=True):
    """``Bottleneck.forward`` (darknet.py:82-84) as C2f builds it: two 3x3 Convs, shortcut (c1 == c2)"""
    return x + conv(p, pre + "cv2.", conv(p, pre + "cv1.", x, 1, training), 1, training)


def c2f(p, pre, x, n, training=True):
    """``C2f.forward`` (darknet.py:101-105): chunk cv1's output in two, chain n bottlenecks off the second half,
    concatenate everything, cv2"""
    y = list(conv(p, pre + "cv1.", x, 1, training).chunk(2, 1))
    for j in range(n):
        y.append(bottleneck(p, f"{pre}m.{j}.", y[-1], training))
    return conv(p, pre + "cv2.", torch.cat(y, 1), 1, training)


def forward(p, x, depth, training=True):
    """``DarkNet.forward`` (darknet.py:151-166)"""
    x = conv(p, "p1.0.", x, 2, training)
    outs = {}
    for i in range(4):
        x = conv(p, f"p{i + 2}.0.", x, 2, training)
        x = c2f(p, f"p{i + 2}.1.", x, depth[i], training)
        outs[f"res{i + 2}"] = x
    return outs
