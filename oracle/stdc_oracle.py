"""Oracle (TEST INFRASTRUCTURE): STDC student, CPU fp32 restatement.  PINNED by tests/golden/stdc_small.npz (generated
from the reference's own STDCNet).  Restates /root/reference/models/backbones/stdc.py (CatBottleneck path only:
``block_type='cat'``, ``use_conv_last=False``, stdc_wrapper.py:61-66) in functional form; parameter names are the
reference's ``features.{i}...`` state-dict keys; BatchNorm runs in train mode and updates running stats in place."""
import torch
import torch.nn.functional as F

STDC_LAYERS = {"stdc_1": [2, 2, 2], "stdc_2": [4, 5, 3]}  # models/wrappers/stdc_wrapper.py:8-11


def _bn(p, pre, x, training=True):
    return F.batch_norm(x, p[pre + "running_mean"], p[pre + "running_var"], p[pre + "weight"], p[pre + "bias"],
                        training, 0.1, 1e-5)


def convx(p, pre, x, stride=1):
    """``ConvX`` (stdc.py:8-24): conv (k from the weight shape, pad k//2, no bias) -> BN -> ReLU"""
    w = p[pre + "conv.weight"]
    return torch.relu(_bn(p, pre + "bn.", F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)))


def cat_bottleneck(p, pre, x, stride, block_num=4):
    """``CatBottleneck.forward`` (stdc.py:153-172)"""
    out1 = convx(p, pre + "conv_list.0.", x)
    outs = []
    out = out1
    for idx in range(1, block_num):
        if idx == 1 and stride == 2:
            a = F.conv2d(out1, p[pre + "avd_layer.0.weight"], None, stride=2, padding=1, groups=out1.shape[1])
            out = convx(p, pre + f"conv_list.{idx}.", _bn(p, pre + "avd_layer.1.", a))
        else:
            out = convx(p, pre + f"conv_list.{idx}.", out)
        outs.append(out)
    if stride == 2:
        out1 = F.avg_pool2d(out1, kernel_size=3, stride=2, padding=1)
    return torch.cat([out1] + outs, dim=1)


def block_plan(base, layers):
    """(in, out, stride) of every CatBottleneck in ``_make_layers`` order (stdc.py:224-252)"""
    plan = []
    for i, n in enumerate(layers):
        for j in range(n):
            if i == 0 and j == 0:
                plan.append((base, base * 4, 2))
            elif j == 0:
                plan.append((base * 2 ** (i + 1), base * 2 ** (i + 2), 2))
            else:
                plan.append((base * 2 ** (i + 2), base * 2 ** (i + 2), 1))
    return plan


def forward(p, x, layers, base=64):
    """``STDCNet.forward`` (stdc.py:254-272): res2 = x4, res3 = x8, res4 = x16, res5 = x32"""
    x = convx(p, "features.0.", x, stride=2)
    x = convx(p, "features.1.", x, stride=2)
    outs = {"res2": x}
    idx = 2
    for i, n in enumerate(layers):
        for j in range(n):
            x = cat_bottleneck(p, f"features.{idx}.", x, 2 if j == 0 else 1)
            idx += 1
        outs[f"res{i + 3}"] = x
    return outs
