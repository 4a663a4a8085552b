#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE's own modules.

Run in the build container only (needs /root/reference; the GPU box has no reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from the reference:
    losses/scalekd.py                  -> ScaleKD, AttentionProjector   (forward + backward)
    train/distillation_module.py       -> DistillationModule.training_step with stand-in teacher/student
    models/backbones/convnext.py       -> D2ConvNextV2
    models/backbones/stdc.py           -> STDCNet (CatBottleneck path)
    models/backbones/swin.py           -> SwinTransformer (drop rate 0)
    models/backbones/resnet.py         -> BasicStem, BottleneckBlock, ResNet (fvcore.c2_msra_fill stand-in = its definition)
The DINOv2 teacher is NOT imported (external torch.hub dependency, fetch denied): the step fixture uses a
stand-in teacher whose ``.model.blocks`` are this repo's own oracle ViT blocks with seeded weights.

Packages absent from this image (torchvision, timm, lightning, termcolor) are replaced by minimal stand-in
modules below; none of them contributes arithmetic on the captured paths (torchvision.resize is only
referenced by dead code, DropPath is Identity at rate 0, trunc_normal_ is torch's own).

Only data (inputs, weights by state-dict name, outputs, gradients) is written to the .npz files.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


def _install_standins():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    def resize(*a, **k):  # only reachable from dead code (losses/scalekd.py:160)
        raise RuntimeError("torchvision stand-in: resize is not available")

    tvf.resize = resize
    tv.transforms = tvt
    tvt.functional = tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})

    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            assert p == 0.0

        def forward(self, x):
            return x

    tl.DropPath = DropPath
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    tl.to_2tuple = lambda v: (v, v) if not isinstance(v, (tuple, list)) else tuple(v)
    timm.models = tm
    tm.layers = tl
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})

    lightning = types.ModuleType("lightning")

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.logged = {}

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, name, value, **k):
            self.logged[name] = float(value.detach()) if torch.is_tensor(value) else float(value)

        def optimizers(self):
            return types.SimpleNamespace(param_groups=[{"lr": 1e-3}])

    lightning.LightningModule = LightningModule
    sys.modules["lightning"] = lightning

    tc = types.ModuleType("termcolor")
    tc.colored = lambda s, *a, **k: s
    sys.modules["termcolor"] = tc


_install_standins()


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32) if t.dtype.is_floating_point else t.detach().cpu().numpy()


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {len(arrays)} arrays, {os.path.getsize(path)/1024:.1f} KiB")


def _load_file(modname, relpath):
    import importlib.util
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# ------------------------------------------------------------------------------------------------
def gen_scalekd():
    skd = _load_file("ref_scalekd", "losses/scalekd.py")
    cases = {
        # name: (student_dims, teacher_dims, P, heads, self_query, window, with_queries)
        "scalekd_selfq": (40, 48, 4, 6, True, (1, 1), False),
        "scalekd_extq": (24, 48, 4, 6, False, (1, 1), True),
        "scalekd_win22": (24, 32, 4, 4, True, (2, 2), False),
        "scalekd_p6": (16, 32, 6, 2, True, (1, 1), False),
    }
    for idx, (name, (cs, d, P, heads, selfq, win, withq)) in enumerate(cases.items()):
        torch.manual_seed(7 + idx)
        m = skd.ScaleKD(name="scalekd_res5", alpha=[0.08, 0.06], student_dims=cs, teacher_dims=d,
                        query_hw=[P, P], pos_hw=[P, P], pos_dims=d, window_shapes=win, self_query=selfq,
                        softmax_scale=[5.0, 5.0], num_heads=heads)
        m.train()
        # make every parameter non-trivial (LN/BN affine default to 1/0, biases to small values)
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                if p_.dim() == 1:
                    p_.add_(0.1 * torch.randn_like(p_))
        B = 3
        arrays = {}
        for k, v in m.state_dict().items():
            arrays["param/" + k] = _np(v)
        x = torch.randn(B, cs, P, P, requires_grad=True)
        t = torch.randn(B, d, P, P)
        qs = qf = None
        if withq:
            qs = torch.randn(B, P * P, d, requires_grad=True)
            qf = torch.randn(B, P * P, d, requires_grad=True)
        out = m(x, t, query_s=qs, query_f=qf)
        out["loss"].backward()
        arrays["in/preds_S"] = _np(x)
        arrays["in/preds_T"] = _np(t)
        if withq:
            arrays["in/query_s"], arrays["in/query_f"] = _np(qs), _np(qf)
            arrays["grad/query_s"], arrays["grad/query_f"] = _np(qs.grad), _np(qf.grad)
        for k, v in out.items():
            arrays["out/" + k] = _np(v)
        arrays["grad/preds_S"] = _np(x.grad)
        for k, p_ in m.named_parameters():
            arrays["pgrad/" + k] = _np(p_.grad)
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                arrays["after/" + k] = _np(v)
        arrays["meta"] = np.array([cs, d, P, heads, int(selfq), win[0], win[1], B], dtype=np.int64)
        # projector tokens on their own (pins AttentionProjector.forward)
        with torch.no_grad():
            m2 = skd.ScaleKD(name="x", alpha=[0.08, 0.06], student_dims=cs, teacher_dims=d, query_hw=[P, P],
                             pos_hw=[P, P], pos_dims=d, window_shapes=win, self_query=selfq,
                             softmax_scale=[5.0, 5.0], num_heads=heads)
            m2.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in arrays.items() if k.startswith("param/")})
            m2.train()
            arrays["out/tokens_spat"] = _np(m2.project_feat_spat(x, query=qs))
        _save(name, **arrays)


def gen_loss_tail():
    skd = _load_file("ref_scalekd", "losses/scalekd.py")
    m = skd.ScaleKD(name="x", alpha=[0.08, 0.06], student_dims=8, teacher_dims=8, query_hw=[2, 2], pos_hw=[2, 2],
                    pos_dims=8, num_heads=2)
    for P, C, B in ((16, 24, 2), (37, 8, 2), (5, 16, 3)):
        torch.manual_seed(100 + P)
        s = torch.randn(B, P * P, C, requires_grad=True)
        t = torch.randn(B, C, P, P) + 0.3
        arrays = {"in/tok_s": _np(s), "in/feat_t": _np(t)}
        ls, ss = m.get_spat_loss(s, t)
        ls.backward()
        arrays["out/spat_loss"], arrays["out/spat_sim"], arrays["grad/spat"] = _np(ls), _np(ss), _np(s.grad)
        s.grad = None
        lf, sf = m.get_freq_loss(s, t)
        lf.backward()
        arrays["out/freq_loss"], arrays["out/freq_sim"], arrays["grad/freq"] = _np(lf), _np(sf), _np(s.grad)
        _save(f"loss_tail_p{P}", **arrays)


def gen_convnext():
    cn = _load_file("ref_convnext", "models/backbones/convnext.py")
    torch.manual_seed(11)
    depths, dims = [1, 1, 2, 1], [8, 16, 32, 64]
    net = cn.D2ConvNextV2(depths=depths, embed_dims=dims, drop_path_rate=0.0, out_features=["res4", "res5"])
    net.train()
    with torch.no_grad():  # reference init leaves GRN gamma/beta and all biases at zero: perturb so they matter
        for n_, p_ in net.named_parameters():
            if "grn" in n_ or p_.dim() == 1:
                p_.add_(0.2 * torch.randn_like(p_))
            else:
                p_.mul_(8.0)
    x = torch.randn(2, 3, 64, 64, requires_grad=True)
    feats = net.forward_features(x)
    arrays = {"in/x": _np(x), "meta/depths": np.array(depths), "meta/dims": np.array(dims)}
    for k, v in net.state_dict().items():
        arrays["param/" + k] = _np(v)
    loss = 0
    for k in ("res2", "res3", "res4", "res5"):
        arrays["out/" + k] = _np(feats[k])
    P = (5, 5)  # ModelWrapper resize (models/model_zoo.py:123-128)
    for k in ("res4", "res5"):
        r = torch.nn.functional.interpolate(feats[k], size=P, mode="bilinear", align_corners=False)
        ct = torch.randn_like(r)
        arrays["out/resized_" + k] = _np(r)
        arrays["in/cotangent_" + k] = _np(ct)
        loss = loss + (r * ct).sum()
    loss.backward()
    arrays["grad/x"] = _np(x.grad)
    for k, p_ in net.named_parameters():
        arrays["pgrad/" + k] = _np(p_.grad)
    _save("convnext_small", **arrays)


def gen_stdc():
    st = _load_file("ref_stdc", "models/backbones/stdc.py")
    torch.manual_seed(21)
    base, layers = 16, [2, 2, 2]
    net = st.STDCNet(base=base, layers=layers)
    net.init_params()
    net.train()
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.2 * torch.randn_like(p_))
    x = torch.randn(2, 3, 64, 64, requires_grad=True)
    arrays = {"in/x": _np(x), "meta": np.array([base] + layers, dtype=np.int64)}
    for k, v in net.state_dict().items():
        if k.startswith("features."):
            arrays["param/" + k] = _np(v)
    arrays["meta/n_state_keys"] = np.array(len(net.state_dict()))
    feats = net(x)
    loss = 0
    for k in ("res2", "res3", "res4", "res5"):
        arrays["out/" + k] = _np(feats[k])
    for k in ("res4", "res5"):
        ct = torch.randn_like(feats[k])
        arrays["in/cotangent_" + k] = _np(ct)
        loss = loss + (feats[k] * ct).sum()
    loss.backward()
    arrays["grad/x"] = _np(x.grad)
    for k, p_ in net.named_parameters():
        if k.startswith("features."):
            arrays["pgrad/" + k] = _np(p_.grad)
    for k, v in net.state_dict().items():
        if k.startswith("features.") and "running" in k:
            arrays["after/" + k] = _np(v)
    _save("stdc_small", **arrays)


def gen_resnet():
    fv = types.ModuleType("fvcore"); fvn = types.ModuleType("fvcore.nn"); fvw = types.ModuleType("fvcore.nn.weight_init")

    def c2_msra_fill(m):  # fvcore's definition: kaiming_normal_(fan_out, relu), zero bias
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    fvw.c2_msra_fill = c2_msra_fill
    fv.nn = fvn; fvn.weight_init = fvw
    sys.modules.update({"fvcore": fv, "fvcore.nn": fvn, "fvcore.nn.weight_init": fvw})
    rn = _load_file("ref_resnet", "models/backbones/resnet.py")
    torch.manual_seed(31)
    chans, nblk, strides = [8, 16, 32, 64, 128], [2, 1, 2, 1], [1, 2, 2, 2]
    stem = rn.BasicStem(in_channels=3, out_channels=chans[0], norm="BN")
    stages = [rn.ResNet.make_stage(rn.BottleneckBlock, n, stride_per_block=[s] + [1] * (n - 1), in_channels=chans[i],
                                   out_channels=chans[i + 1], bottleneck_channels=chans[i + 1] // 4, norm="BN")
              for i, (n, s) in enumerate(zip(nblk, strides))]
    net = rn.ResNet(stem, stages, num_classes=None, out_features=["res2", "res3", "res4", "res5"], freeze_at=0)
    net.train()
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.2 * torch.randn_like(p_))
    x = torch.randn(2, 3, 64, 64, requires_grad=True)
    arrays = {"in/x": _np(x), "meta/nblk": np.array(nblk), "meta/chans": np.array(chans)}
    for k, v in net.state_dict().items():
        arrays["param/" + k] = _np(v)
    feats = net(x)
    loss = 0
    for k in ("res2", "res3", "res4", "res5"):
        arrays["out/" + k] = _np(feats[k])
    for k in ("res4", "res5"):
        ct = torch.randn_like(feats[k])
        arrays["in/cotangent_" + k] = _np(ct)
        loss = loss + (feats[k] * ct).sum()
    loss.backward()
    arrays["grad/x"] = _np(x.grad)
    for k, p_ in net.named_parameters():
        arrays["pgrad/" + k] = _np(p_.grad)
    for k, v in net.state_dict().items():
        if "running" in k:
            arrays["after/" + k] = _np(v)
    _save("resnet_small", **arrays)


def gen_presnet():
    """the reference's own D2Presnet (models/backbones/presnet.py needs nothing but torch), depth 18 (BasicBlock) and 50
    (BottleNeck), variant "d", live BatchNorm, batch 4 at 98x98: the stages see 25 / 13 / 7 / 4 pixels a side, so two
    average-pool shortcuts clip their last window, and the deepest BatchNorm still averages over 64 samples (with 2 x 3 x 3
    the fp32 round-off of a 50-layer backward pass reaches 0.5 % of the first conv's gradient in any summation order).  The widths are
    hard-wired in the reference (presnet.py:345), so the weights come from tests/golden/param_recipe.py instead of the file."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import param_recipe as pr_
    pr = _load_file("ref_presnet", "models/backbones/presnet.py")
    for depth in (18, 50):
        net = pr.D2Presnet(depth=depth, variant="d", num_stages=4, act="relu", freeze_at=-1, freeze_norm=False, pretrained=False)
        state = pr_.seeded_state({k: v.shape for k, v in net.state_dict().items()}, seed=40 + depth)
        net.load_state_dict(state, strict=True)
        net.train()
        g = torch.Generator().manual_seed(depth)
        x = torch.randn(4, 3, 98, 98, generator=g).requires_grad_(True)
        arrays = {"in/x": _np(x), "meta/depth": np.array([depth]), "meta/seed": np.array([40 + depth]),
                  "meta/param_checksum": np.array([pr_.checksum(state)])}
        feats = net(x)
        loss = 0
        for k in ("res2", "res3"):  # large maps: digest only
            arrays["out_digest/" + k] = _np(pr_.digest(feats[k]))
        for k in ("res4", "res5"):
            arrays["out/" + k] = _np(feats[k])
        for k in ("res4", "res5"):
            ct = torch.randn(feats[k].shape, generator=g)
            arrays["in/cotangent_" + k] = _np(ct)
            loss = loss + (feats[k] * ct).sum()
        loss.backward()
        arrays["grad/x"] = _np(x.grad)
        for k, p_ in net.named_parameters():
            arrays["pgrad_digest/" + k] = _np(pr_.digest(p_.grad))
        for k, v in net.state_dict().items():
            if "running" in k:
                arrays["after/" + k] = _np(v)
        _save(f"presnet{depth}_small", **arrays)


def gen_mobilenet_v2():
    """the reference's own D2MobileNetV2 (models/backbones/mobilenet_v2.py needs nothing but torch), default configuration,
    batch 4 at 96x96 (res5 is 3x3: the deepest BatchNorm averages over 36 samples)"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import param_recipe as pr_
    mb = _load_file("ref_mobilenet_v2", "models/backbones/mobilenet_v2.py")
    for name, strides, dilations in (("mobilenet_v2_small", (1, 2, 2, 2, 1, 2, 1), (1, 1, 1, 1, 1, 1, 1)),
                                     ("mobilenet_v2_os8_small", (1, 2, 2, 1, 1, 1, 1), (1, 1, 1, 2, 2, 4, 4))):
        _gen_mobilenet_v2(mb, pr_, name, strides, dilations)


def _gen_mobilenet_v2(mb, pr_, name, strides, dilations):
    net = mb.D2MobileNetV2(widen_factor=1.0, strides=strides, dilations=dilations,
                           out_features=["res2", "res3", "res4", "res5"])
    state = pr_.seeded_state({k: v.shape for k, v in net.state_dict().items()}, seed=77)
    net.load_state_dict(state, strict=True)
    net.train()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(4, 3, 96, 96 if dilations[-1] == 1 else 48, generator=g).requires_grad_(True)  # (os8 keeps stride 8)
    arrays = {"in/x": _np(x), "meta/seed": np.array([77]), "meta/param_checksum": np.array([pr_.checksum(state)])}
    feats = net(x)
    loss = 0
    for k in ("res2", "res3"):
        arrays["out_digest/" + k] = _np(pr_.digest(feats[k]))
    for k in ("res4", "res5"):
        arrays["out/" + k] = _np(feats[k])
        ct = torch.randn(feats[k].shape, generator=g)
        arrays["in/cotangent_" + k] = _np(ct)
        loss = loss + (feats[k] * ct).sum()
    loss.backward()
    arrays["grad/x"] = _np(x.grad)
    for k, p_ in net.named_parameters():
        arrays["pgrad_digest/" + k] = _np(pr_.digest(p_.grad))
    for k, v in net.state_dict().items():
        if "running" in k:
            arrays["after/" + k] = _np(v)
    _save(name, **arrays)


def gen_darknet():
    """the reference's own DarkNet (models/backbones/darknet.py needs nothing but torch) at the "n" widths of
    darknet_wrapper.py:9 (C2f depths 1/2/2/1), batch 4 at 96x96: res5 is 3x3, so the deepest BatchNorm averages over 36 samples"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import param_recipe as pr_
    dk = _load_file("ref_darknet", "models/backbones/darknet.py")
    depth, width = [1, 2, 2, 1], [3, 16, 32, 64, 128, 256]
    net = dk.DarkNet(depth, width)
    state = pr_.seeded_state({k: v.shape for k, v in net.state_dict().items()}, seed=58)
    net.load_state_dict(state, strict=True)
    net.train()
    g = torch.Generator().manual_seed(58)
    x = torch.randn(4, 3, 96, 96, generator=g).requires_grad_(True)
    arrays = {"in/x": _np(x), "meta/seed": np.array([58]), "meta/param_checksum": np.array([pr_.checksum(state)]),
              "meta/depth": np.array(depth), "meta/width": np.array(width)}
    feats = net(x)
    loss = 0
    for k in ("res2", "res3"):
        arrays["out_digest/" + k] = _np(pr_.digest(feats[k]))
    for k in ("res4", "res5"):
        arrays["out/" + k] = _np(feats[k])
        ct = torch.randn(feats[k].shape, generator=g)
        arrays["in/cotangent_" + k] = _np(ct)
        loss = loss + (feats[k] * ct).sum()
    loss.backward()
    arrays["grad/x"] = _np(x.grad)
    for k, p_ in net.named_parameters():
        arrays["pgrad_digest/" + k] = _np(pr_.digest(p_.grad))
    for k, v in net.state_dict().items():
        if "running" in k:
            arrays["after/" + k] = _np(v)
    _save("darknet_n_small", **arrays)


def gen_mit():
    """the reference's own MultiscaleImageTransformer, configured as MITWrapper("mit_b0") does (mit_wrapper.py:76-86): heads
    1/2/4/8 (head dims 32, 32, 40, 32), sr ratios 8/4/2/1, every dropout rate 0.  timm is absent: DropPath (identity at rate 0),
    to_2tuple and trunc_normal_ are the stand-ins of _install_standins()."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import param_recipe as pr_
    _install_standins()
    mit = _load_file("ref_mit", "models/backbones/mit.py")
    depths, dims = [2, 2, 2, 2], [32, 64, 160, 256]
    net = mit.MultiscaleImageTransformer(embed_dims=dims, depths=depths, num_heads=[1, 2, 4, 8], mlp_ratios=[4, 4, 4, 4],
                                         qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0)
    state = pr_.seeded_state({k: v.shape for k, v in net.state_dict().items()}, seed=91)
    net.load_state_dict(state, strict=True)
    net.train()
    g = torch.Generator().manual_seed(91)
    x = torch.randn(2, 3, 128, 96, generator=g).requires_grad_(True)   # stage grids 32x24, 16x12, 8x6, 4x3
    arrays = {"in/x": _np(x), "meta/seed": np.array([91]), "meta/param_checksum": np.array([pr_.checksum(state)]),
              "meta/depths": np.array(depths), "meta/dims": np.array(dims)}
    feats = net(x)
    loss = 0
    for k in ("res2", "res3"):
        arrays["out_digest/" + k] = _np(pr_.digest(feats[k]))
    for k in ("res4", "res5"):
        arrays["out/" + k] = _np(feats[k])
        ct = torch.randn(feats[k].shape, generator=g)
        arrays["in/cotangent_" + k] = _np(ct)
        loss = loss + (feats[k] * ct).sum()
    loss.backward()
    arrays["grad/x"] = _np(x.grad)
    for k, p_ in net.named_parameters():
        if p_.grad is not None:
            arrays["pgrad_digest/" + k] = _np(pr_.digest(p_.grad))
    _save("mit_b0_small", **arrays)


def gen_swin():
    """SwinTransformer (models/backbones/swin.py) at drop rate 0: odd grids, window padding, shifted masks, patch merging."""
    sw = _load_file("ref_swin", "models/backbones/swin.py")
    torch.manual_seed(41)
    dim, depths, heads, ws = 16, [2, 2, 2, 2], [1, 1, 2, 4], 7
    net = sw.SwinTransformer(pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=dim, depths=depths, num_heads=heads,
                             window_size=ws, mlp_ratio=[4, 4, 4, 4], qkv_bias=True, qk_scale=None, drop_rate=0.0,
                             attn_drop_rate=0.0, drop_path_rate=0.0)
    net.train()
    with torch.no_grad():  # default init leaves LN at (1,0) and the bias table at 0.02: perturb so they matter
        for n_, p_ in net.named_parameters():
            if not p_.requires_grad:
                continue
            if "relative_position_bias_table" in n_:
                p_.mul_(25.0)
            elif p_.dim() == 1:
                p_.add_(0.2 * torch.randn_like(p_))
    x = torch.randn(2, 3, 86, 90, requires_grad=True)  # 86,90 -> padded 88,92 -> grids 22x23, 11x12, 6x6, 3x3
    arrays = {"in/x": _np(x), "meta/depths": np.array(depths), "meta/heads": np.array(heads), "meta/dim": np.array(dim),
              "meta/window": np.array(ws)}
    for k, v in net.state_dict().items():
        arrays["param/" + k] = _np(v)
    feats = net(x)
    loss = 0
    for k in ("res2", "res3", "res4", "res5"):
        arrays["out/" + k] = _np(feats[k])
    P = (5, 5)  # ModelWrapper resize (models/model_zoo.py:123-128)
    for k in ("res4", "res5"):
        r = torch.nn.functional.interpolate(feats[k], size=P, mode="bilinear", align_corners=False)
        ct = torch.randn_like(r)
        arrays["out/resized_" + k] = _np(r)
        arrays["in/cotangent_" + k] = _np(ct)
        loss = loss + (r * ct).sum()
    loss.backward()
    arrays["grad/x"] = _np(x.grad)
    for k, p_ in net.named_parameters():
        if p_.requires_grad and p_.grad is not None:
            arrays["pgrad/" + k] = _np(p_.grad)
    _save("swin_small", **arrays)


def gen_step():
    """DistillationModule.training_step (train/distillation_module.py:247-276) with stand-ins."""
    from oracle import vit_oracle as vit
    import train.distillation_module as dm  # the reference's module (imports reference losses/ and utils/)

    D, heads, depth, P = 48, 3, 12, 4
    c4, c5, B = 24, 40, 2
    torch.manual_seed(5)
    tcfg = dict(embed_dim=D, depth=depth, num_heads=heads)
    tp = vit.init_vit_params(tcfg, seed=3)
    with torch.no_grad():
        for k in tp:  # non-trivial LayerScale/bias so the block arithmetic is exercised
            if k.endswith("gamma"):
                tp[k] = 0.5 + torch.rand_like(tp[k])
            elif k.endswith("bias"):
                tp[k] = 0.05 * torch.randn_like(tp[k])
            elif k.endswith("weight") and tp[k].dim() == 2:
                tp[k] = tp[k] * 5

    class Blk(nn.Module):
        def __init__(self, i):
            super().__init__()
            self.i = i

        def forward(self, x):
            return vit.vit_block(tp, f"blocks.{self.i}.", x, heads)

    feat_t = torch.randn(B, P * P, D).reshape(B, P, P, D).permute(0, 3, 1, 2)

    class Teacher(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = nn.Module()
            self.model.blocks = nn.ModuleList([Blk(i) for i in range(depth)])

        def forward(self, x):
            return {"feature_map": feat_t}

    s4 = torch.randn(B, c4, P, P, requires_grad=True)
    s5 = torch.randn(B, c5, P, P, requires_grad=True)

    class Student(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = types.SimpleNamespace(ckpt_dir="unused.pth")

        def forward(self, x):
            return {"res4": s4, "res5": s5}

    class Cfg(dict):
        __getattr__ = dict.__getitem__

    def loss_spec(name, heads_, selfq, cs):
        return Cfg(type="scalekd", weight=1.0 if name.endswith("5") else 0.7, kwargs=Cfg(
            alpha=[0.08, 0.06], window_shapes=[1, 1], self_query=selfq, softmax_scale=[5.0, 5.0],
            num_heads=heads_, name=name, teacher_dims=D, pos_dims=D, pos_hw=[P, P], query_hw=[P, P], student_dims=cs))

    cfg = Cfg(student=Cfg(model_name="convnext_tiny", checkpoint_path=None),
              teacher=Cfg(model_name="dinov2_vits14", teacher_key="feature_map"),
              loss=Cfg(losses=[loss_spec("scalekd_res4", 6, True, c4), loss_spec("scalekd_res5", 3, False, c5)]),
              optimizer=Cfg(type="AdamW", kwargs=Cfg(lr=1e-3)))
    dm.DistillationModule._load_student_checkpoint = lambda self, path: None  # no checkpoint files exist
    mod = dm.DistillationModule(student=Student(), teacher=Teacher(), cfg=cfg)
    mod.train()
    mod.teacher.eval()
    with torch.no_grad():
        for n_, p_ in mod.losses.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    arrays = {}
    for k, v in mod.losses.state_dict().items():
        arrays["param/" + k] = _np(v)
    for k, v in tp.items():
        if k.startswith("blocks.9.") or k.startswith("blocks.10."):
            arrays["teacher/" + k] = _np(v)
    loss = mod.training_step(torch.zeros(B, 3, 56, 56), 0)
    loss.backward()
    for k, v in mod.logged.items():
        arrays["log/" + k] = np.float32(v)
    arrays["in/res4"], arrays["in/res5"], arrays["in/feat_t"] = _np(s4), _np(s5), _np(feat_t)
    arrays["grad/res4"], arrays["grad/res5"] = _np(s4.grad), _np(s5.grad)
    for k, p_ in mod.losses.named_parameters():
        arrays["pgrad/" + k] = _np(p_.grad)
    arrays["meta"] = np.array([D, heads, depth, P, c4, c5, B], dtype=np.int64)
    _save("step_res45", **arrays)


if __name__ == "__main__":
    which = sys.argv[1:] or ["scalekd", "loss_tail", "convnext", "stdc", "resnet", "presnet", "mobilenet_v2", "darknet", "mit", "swin", "step"]
    for w in which:
        globals()["gen_" + w]()
