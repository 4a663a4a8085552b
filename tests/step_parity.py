"""Shared helper of the whole-step parity tests (TEST INFRASTRUCTURE): run the CPU oracle of the distillation step with a
module's weights, in fp32 or - as the arbiter of what fp32 can deliver - in fp64, for every in-scope student family."""
import torch
import torch.nn.functional as F

RESNET_BLOCKS = {"resnet_50": [3, 4, 6, 3], "resnet_101": [3, 4, 23, 3]}
STDC_LAYERS = {"stdc_1": [2, 2, 2], "stdc_2": [4, 5, 3]}


def oracle_step(mod, cfg, x, dtype=torch.float32, teacher_bf16=False, autocast=False):
    """-> (logged-scalar dict, {student key: grad}, {loss-module key: grad}) of the oracle run with ``mod``'s weights.
    teacher_bf16: the frozen teacher (forward and block re-use) in the "bf16 operands, fp32 accumulate" arithmetic.
    autocast: the whole step under torch's autocast policy in bfloat16, as the reference's default precision=16 runs it - teacher
    forward (oracle/vit_oracle.py), ConvNeXt backbone and feature resize, ScaleKD projectors, re-used teacher blocks and the
    Linear-built DCT of the frequency loss: torch.autocast itself on the oracle's op sequences, CUDA's fp32 op list forced where
    the CPU policy differs."""
    from oracle import convnext_oracle as cno, resnet_oracle as ro, stdc_oracle as sto, step_oracle as so, swin_oracle as swo
    from oracle import vit_oracle as vo

    def cast(v, grad):
        v = v.detach().cpu().clone()
        if v.dtype.is_floating_point:
            v = v.to(dtype)
            if grad:
                v.requires_grad_(True)
        return v

    sp = {k[len("model.model."):]: cast(v, "running" not in k) for k, v in mod.student.state_dict().items()}
    lp = {k: cast(v, "running" not in k) for k, v in mod.losses.state_dict().items()}
    tp = {k: cast(v, False) for k, v in mod.teacher.model.state_dict().items()}
    tcfg = vo.VIT_CONFIGS[cfg.teacher.model_name]
    xd = x.to(dtype)
    with torch.no_grad():
        feat_t = vo.teacher_feature_map_autocast(tp, xd, tcfg) if autocast else vo.teacher_feature_map(tp, xd, tcfg, bf16=teacher_bf16)
    name, grid, keys = cfg.student.model_name, tuple(cfg.teacher.n_patches), cfg.student.student_keys
    if name.startswith("swin"):
        feats = swo.student_features(sp, xd, swo.SWIN_CONFIGS[name], grid, keys)
    elif name.startswith("convnext") and autocast:
        f = cno.forward_features_autocast(sp, xd, cno.CONVNEXT_CONFIGS[name]["depths"])
        feats = {k: F.interpolate(f[k], size=grid, mode="bilinear", align_corners=False) for k in keys}   # 16-bit in, 16-bit out
    elif name.startswith("convnext"):
        feats = cno.student_features(sp, xd, cno.CONVNEXT_CONFIGS[name]["depths"], grid, keys)
    elif name.startswith("resnet"):
        f = ro.forward(sp, xd, RESNET_BLOCKS[name])
        feats = {k: F.interpolate(f[k], size=grid, mode="bilinear", align_corners=False) for k in keys}
    elif name.startswith("stdc"):
        f = sto.forward(sp, xd, STDC_LAYERS[name], 64)
        feats = {k: F.interpolate(f[k], size=grid, mode="bilinear", align_corners=False) for k in keys}
    else:
        raise KeyError(name)
    cfgs, w = {}, {}
    for spec in cfg.loss.losses:
        kw = spec.kwargs
        cfgs[kw.name] = dict(alpha=kw.alpha, hw=tuple(kw.pos_hw), num_heads=kw.num_heads, softmax_scale=kw.softmax_scale)
        w[kw.name] = spec.weight
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = so.compute_losses(lp, cfgs, w, feats, feat_t, tp, tcfg["depth"], tcfg["num_heads"], policy=True)
    else:
        out = so.compute_losses(lp, cfgs, w, feats, feat_t, tp, tcfg["depth"], tcfg["num_heads"], teacher_bf16=teacher_bf16)
    out["loss"].backward()
    gs = {k: v.grad for k, v in sp.items() if v.dtype.is_floating_point and v.requires_grad and v.grad is not None}
    gl = {k: v.grad for k, v in lp.items() if v.dtype.is_floating_point and v.requires_grad and v.grad is not None}
    return {k: v.detach() for k, v in out.items()}, gs, gl


def close(a, b, what, rtol=1e-3):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite"
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= rtol * scale + 1e-7, f"{what}: max|diff| {err:.3e} > {rtol:g} * {scale:.3e}"


def flat(named_grads, names):
    return torch.cat([named_grads[k].detach().cpu().reshape(-1).double() for k in names])


def rel_l2(a, b):
    return ((a - b).norm() / b.norm()).item()


def module_grads(mod):
    """({student key: grad}, {loss key: grad}) of the HIP module after backward (student keys without 'model.model.')"""
    gs = {k[len("model.model."):]: p.grad for k, p in mod.student.named_parameters() if p.grad is not None}
    gl = {k: p.grad for k, p in mod.losses.named_parameters() if p.grad is not None}
    return gs, gl
