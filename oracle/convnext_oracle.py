"""Oracle (TEST INFRASTRUCTURE): ConvNeXt-V2 student + ModelWrapper resize, CPU fp32 restatement.
PINNED by tests/golden/convnext_small.npz (generated from the reference's own backbone).

Restates /root/reference/models/backbones/convnext.py and the resize of models/model_zoo.py:116-129 in
functional form; parameter names are the reference's state-dict names of ``D2ConvNextV2``
(``downsample_layers.{i}.{j}.weight``, ``stages.{i}.{j}.dwconv.weight`` ...).
"""
import torch
import torch.nn.functional as F

# models/wrappers/convnext_wrapper.py:19-49
CONVNEXT_CONFIGS = {
    "convnext_atto": dict(depths=[2, 2, 6, 2], dims=[40, 80, 160, 320]),
    "convnext_pico": dict(depths=[2, 2, 6, 2], dims=[64, 128, 256, 512]),
    "convnext_nano": dict(depths=[2, 2, 8, 2], dims=[80, 160, 320, 640]),
    "convnext_tiny": dict(depths=[3, 3, 9, 3], dims=[96, 192, 384, 768]),
    "convnext_base": dict(depths=[3, 3, 27, 3], dims=[128, 256, 512, 1024]),
}


def init_convnext_params(depths, dims, seed=1, in_chans=3, dtype=torch.float32):
    """Reference init rule (convnext.py:150-153): trunc-normal(0.02) for conv/linear weights, zero bias;
    LayerNorm (1,0) (:20-21); GRN gamma/beta zero (:43-44)."""
    g = torch.Generator().manual_seed(seed)

    def tn(*shape):
        t = torch.empty(*shape, dtype=dtype)
        torch.nn.init.trunc_normal_(t, std=0.02, generator=g)
        return t

    p = {}
    p["downsample_layers.0.0.weight"] = tn(dims[0], in_chans, 4, 4)
    p["downsample_layers.0.0.bias"] = torch.zeros(dims[0], dtype=dtype)
    p["downsample_layers.0.1.weight"] = torch.ones(dims[0], dtype=dtype)
    p["downsample_layers.0.1.bias"] = torch.zeros(dims[0], dtype=dtype)
    for i in range(3):
        p[f"downsample_layers.{i+1}.0.weight"] = torch.ones(dims[i], dtype=dtype)
        p[f"downsample_layers.{i+1}.0.bias"] = torch.zeros(dims[i], dtype=dtype)
        p[f"downsample_layers.{i+1}.1.weight"] = tn(dims[i + 1], dims[i], 2, 2)
        p[f"downsample_layers.{i+1}.1.bias"] = torch.zeros(dims[i + 1], dtype=dtype)
    for i in range(4):
        C = dims[i]
        for j in range(depths[i]):
            b = f"stages.{i}.{j}."
            p[b + "dwconv.weight"] = tn(C, 1, 7, 7)
            p[b + "dwconv.bias"] = torch.zeros(C, dtype=dtype)
            p[b + "norm.weight"] = torch.ones(C, dtype=dtype)
            p[b + "norm.bias"] = torch.zeros(C, dtype=dtype)
            p[b + "pwconv1.weight"] = tn(4 * C, C)
            p[b + "pwconv1.bias"] = torch.zeros(4 * C, dtype=dtype)
            p[b + "grn.gamma"] = torch.zeros(1, 1, 1, 4 * C, dtype=dtype)
            p[b + "grn.beta"] = torch.zeros(1, 1, 1, 4 * C, dtype=dtype)
            p[b + "pwconv2.weight"] = tn(C, 4 * C)
            p[b + "pwconv2.bias"] = torch.zeros(C, dtype=dtype)
    return p


def _ln_channels_first(x, w, b, eps=1e-6):
    """convnext.py:31-36: biased variance over the channel axis of an NCHW tensor."""
    u = x.mean(1, keepdim=True)
    s = ((x - u) ** 2).mean(1, keepdim=True)
    return w[None, :, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[None, :, None, None]


def grn(x, gamma, beta):
    """convnext.py:47-50 on NHWC: L2 over (H,W), divided by its channel mean (+1e-6)."""
    gx = torch.linalg.vector_norm(x, ord=2, dim=(1, 2), keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    return gamma * (x * nx) + beta + x


def block(p, pre, x):
    """``Block.forward`` (convnext.py:75-87); drop_path is Identity at rate 0 (convnext_wrapper.py:97)."""
    C = x.shape[1]
    y = F.conv2d(x, p[pre + "dwconv.weight"], p[pre + "dwconv.bias"], padding=3, groups=C)
    y = y.permute(0, 2, 3, 1)
    y = F.layer_norm(y, (C,), p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-6)
    y = F.gelu(F.linear(y, p[pre + "pwconv1.weight"], p[pre + "pwconv1.bias"]))
    y = grn(y, p[pre + "grn.gamma"], p[pre + "grn.beta"])
    y = F.linear(y, p[pre + "pwconv2.weight"], p[pre + "pwconv2.bias"])
    return x + y.permute(0, 3, 1, 2)


def forward_features(p, x, depths):
    """``ConvNeXtV2.forward_features`` (convnext.py:155-162): all four stages, dict res2..res5 (NCHW)."""
    outs = {}
    for i in range(4):
        if i == 0:  # stem: 4x4 s4 conv then channels-first LN (:116-120)
            x = F.conv2d(x, p["downsample_layers.0.0.weight"], p["downsample_layers.0.0.bias"], stride=4)
            x = _ln_channels_first(x, p["downsample_layers.0.1.weight"], p["downsample_layers.0.1.bias"])
        else:  # LN then 2x2 s2 conv (:121-126)
            x = _ln_channels_first(x, p[f"downsample_layers.{i}.0.weight"], p[f"downsample_layers.{i}.0.bias"])
            x = F.conv2d(x, p[f"downsample_layers.{i}.1.weight"], p[f"downsample_layers.{i}.1.bias"], stride=2)
        for j in range(depths[i]):
            x = block(p, f"stages.{i}.{j}.", x)
        outs[f"res{i+2}"] = x
    return outs


def student_features(p, x, depths, n_patches, targets=("res5", "res4")):
    """``ModelWrapper.forward`` (models/model_zoo.py:116-129): every target feature is bilinearly resized
    (align_corners=False) to the teacher grid."""
    feats = forward_features(p, x, depths)
    return {k: F.interpolate(feats[k], size=tuple(n_patches), mode="bilinear", align_corners=False)
            for k in targets if k in feats}


# ---- the same backbone under the reference's DEFAULT trainer precision (train.py:263 precision=16 == torch autocast) --------------
# torch.autocast itself decides what conv2d / linear do (every tensor argument cast to 16 bit, 16-bit result) and autograd replays
# the backward in the dtypes of the forward; run on the CPU in bfloat16.  CUDA's autocast additionally runs layer_norm, pow and norm
# in fp32 (torch/csrc/autocast_mode.cpp, the `fp32` cast-policy list), which the CPU policy does not: those three call sites are
# forced to fp32 here with explicit casts, so that the op policy is the one the reference's GPU run applies.  PARITY UNPINNED for
# that policy (no CUDA device to run the reference's autocast on); the ops themselves are the pinned fp32 restatement above.
def _fp32(fn, *args, **kw):
    with torch.autocast("cpu", enabled=False):
        return fn(*[a.float() if torch.is_tensor(a) and a.is_floating_point() else a for a in args], **kw)


def _ln_channels_first_autocast(x, w, b, eps=1e-6):
    """convnext.py:31-36 op by op: mean and (x - u) in the dtype of x, pow (an fp32 op under CUDA autocast) and everything after
    it in fp32"""
    u = x.mean(1, keepdim=True)
    d = x - u
    s = _fp32(lambda t: t.pow(2), d).mean(1, keepdim=True)
    xn = d / torch.sqrt(s + eps)
    return w[None, :, None, None] * xn + b[None, :, None, None]


def grn_autocast(x, gamma, beta):
    gx = _fp32(lambda t: torch.linalg.vector_norm(t, ord=2, dim=(1, 2), keepdim=True), x)  # torch.norm: fp32 under CUDA autocast
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    return gamma * (x * nx) + beta + x


def block_autocast(p, pre, x):
    C = x.shape[1]
    y = F.conv2d(x, p[pre + "dwconv.weight"], p[pre + "dwconv.bias"], padding=3, groups=C)
    y = y.permute(0, 2, 3, 1)
    y = _fp32(F.layer_norm, y, (C,), p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-6)
    y = F.gelu(F.linear(y, p[pre + "pwconv1.weight"], p[pre + "pwconv1.bias"]))
    y = grn_autocast(y, p[pre + "grn.gamma"], p[pre + "grn.beta"])
    y = F.linear(y, p[pre + "pwconv2.weight"], p[pre + "pwconv2.bias"])
    return x + y.permute(0, 3, 1, 2)


def forward_features_autocast(p, x, depths):
    """forward_features under torch.autocast(bfloat16) with CUDA's op policy; fp32 parameters, the dtypes of the results are
    autocast's (res2 fp32: the stem LayerNorm returns fp32 and stage 0 adds 16-bit branches to it; res3..res5 bfloat16)"""
    outs = {}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        for i in range(4):
            if i == 0:
                x = F.conv2d(x, p["downsample_layers.0.0.weight"], p["downsample_layers.0.0.bias"], stride=4)
                x = _ln_channels_first_autocast(x, p["downsample_layers.0.1.weight"], p["downsample_layers.0.1.bias"])
            else:
                x = _ln_channels_first_autocast(x, p[f"downsample_layers.{i}.0.weight"], p[f"downsample_layers.{i}.0.bias"])
                x = F.conv2d(x, p[f"downsample_layers.{i}.1.weight"], p[f"downsample_layers.{i}.1.bias"], stride=2)
            for j in range(depths[i]):
                x = block_autocast(p, f"stages.{i}.{j}.", x)
            outs[f"res{i+2}"] = x
    return outs
