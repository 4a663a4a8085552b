"""CPU: host-side logic of the path - reference-compatible module surface / state-dict names, config derivation,
block-range logic, error behaviour, and the data-parallel gradient exchange (gloo, world_size 2)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_names_match_reference_golden(golden_dir):
    """the reference's own state-dict keys (captured in the golden files) load into our modules with strict=True"""
    import numpy as np
    from dinov2_distillation_amd.losses import ScaleKD
    from dinov2_distillation_amd.models.convnext import D2ConvNextV2
    z = np.load(os.path.join(golden_dir, "scalekd_selfq.npz"))
    cs, d, P, heads, selfq, wh, ww, B = [int(v) for v in z["meta"]]
    m = ScaleKD(name="x", alpha=[0.08, 0.06], student_dims=cs, teacher_dims=d, query_hw=[P, P], pos_hw=[P, P], pos_dims=d,
                self_query=True, softmax_scale=[5.0, 5.0], num_heads=heads)
    m.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}, strict=True)
    z = np.load(os.path.join(golden_dir, "convnext_small.npz"))
    net = D2ConvNextV2([int(v) for v in z["meta/depths"]], [int(v) for v in z["meta/dims"]])
    net.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}, strict=True)


def test_module_surface_and_param_counts():
    from dinov2_distillation_amd.config import make_config, set_student_dims
    from dinov2_distillation_amd.models import DINOv2ViT, ModelWrapper
    from dinov2_distillation_amd.train import LOSS_REGISTRY, DistillationModule
    cfg = make_config()
    s = ModelWrapper("convnext_tiny", cfg.teacher.n_patches, cfg.student.student_keys)
    assert s.feature_channels == {"res2": 96, "res3": 192, "res4": 384, "res5": 768}
    assert s.model.ckpt_dir.endswith("convnextv2_tiny_1k_224_ema.pkl")
    set_student_dims(cfg, s)
    t = DINOv2ViT("dinov2_vits14")
    mod = DistillationModule(s, t, cfg)
    assert set(LOSS_REGISTRY) == {"scalekd"}
    assert sorted(mod.losses) == ["scalekd_res4", "scalekd_res5"]
    assert abs(sum(p.numel() for p in s.parameters()) / 1e6 - 27.86) < 0.01
    assert abs(sum(p.numel() for p in mod.losses.parameters()) / 1e6 - 8.58) < 0.01
    assert abs(sum(p.numel() for p in t.parameters()) / 1e6 - 22.06) < 0.01
    assert all(not p.requires_grad for p in t.parameters()) and not t.training
    mod.train()
    assert not mod.teacher.training and mod.student.training  # teacher stays in eval during fit
    keys = list(s.state_dict())
    assert keys[0] == "model.model.downsample_layers.0.0.weight" and "model.model.stages.2.8.grn.gamma" in keys
    assert len(t.model.blocks) == 12 and "blocks.11.ls2.gamma" in t.model.state_dict()


def test_error_behaviour_matches_reference():
    from dinov2_distillation_amd.losses.scalekd import AttentionProjector
    from dinov2_distillation_amd.models import ModelWrapper
    p = AttentionProjector(8, 16, (2, 2), 16, self_query=False, num_heads=2)
    with pytest.raises(NotImplementedError, match="There is no query"):   # losses/scalekd.py:236
        p(torch.zeros(1, 8, 2, 2))
    with pytest.raises(ValueError):   # 24 does not divide 1024
        AttentionProjector(8, 1024, (2, 2), 1024, num_heads=24)
    with pytest.raises(NotImplementedError):
        ModelWrapper("efficientnet_b0", [16, 16])   # a family without an MI355X backbone fails loudly
    dk = ModelWrapper("darknet_s", [16, 16])   # counts from the reference's DarkNet(*versions["s"]) in this container
    assert sum(p.numel() for p in dk.parameters()) == 4_422_816 and len(dk.model.model.state_dict()) == 150
    assert dk.feature_channels == {"res5": 512, "res4": 256} and dk.model.ckpt_dir.endswith("yolov8s.pkl")
    assert "model.model.p3.1.m.1.cv2.bn.running_var" in dk.state_dict() and dk.model.model.p2[0].bn.momentum == 0.03
    assert sum(p.numel() for p in ModelWrapper("darknet_m", [16, 16]).parameters()) == 11_024_688
    with pytest.raises(NotImplementedError, match="DarkNet config"):
        ModelWrapper("darknet_q", [16, 16])
    mit = ModelWrapper("mit_b0", [16, 16])    # counts from the reference's MultiscaleImageTransformer in this container
    assert sum(p.numel() for p in mit.parameters()) == 3_319_392 and len(mit.model.model.state_dict()) == 176
    assert mit.feature_channels == {"res2": 32, "res3": 64, "res4": 160, "res5": 256}
    assert "model.model.block1.0.attn.sr.weight" in mit.state_dict() and "model.model.block4.0.attn.sr.weight" not in mit.state_dict()
    assert sum(p.numel() for p in ModelWrapper("mit_b2", [16, 16]).parameters()) == 24_196_288
    with pytest.raises(NotImplementedError, match="Swin config"):   # models/wrappers/swin_wrapper.py:39-41
        ModelWrapper("swin_huge", [16, 16])
    w = ModelWrapper("swin_tiny", [37, 37])
    assert sum(p.numel() for p in w.parameters() if p.requires_grad) == 27_520_698 and w.feature_channels["res4"] == 384
    sd = w.state_dict()
    assert sd["model.model.layers.2.blocks.5.attn.relative_position_index"].dtype == torch.int64  # swin.py:139-141
    assert "model.model.layers.0.downsample.reduction.weight" in sd and "model.model.norm3.bias" in sd
    r = ModelWrapper("resnet_50", [16, 16])
    assert abs(sum(p.numel() for p in r.parameters()) / 1e6 - 23.51) < 0.01 and r.feature_channels["res5"] == 2048
    with pytest.raises(NotImplementedError, match="PResNet config"):   # models/wrappers/presnet_wrapper.py:24-29
        ModelWrapper("presnet_20", [16, 16])
    pr = ModelWrapper("presnet_50", [16, 16])   # counts taken from the reference's D2Presnet(50, variant="d") in this container
    assert sum(p.numel() for p in pr.parameters()) == 23_527_264 and len(pr.model.model.state_dict()) == 330
    assert pr.feature_channels == {"res2": 256, "res3": 512, "res4": 1024, "res5": 2048}
    assert "model.model.res_layers.1.blocks.0.short.conv.conv.weight" in pr.state_dict()   # "d" shortcut after the pool
    assert "model.model.res_layers.0.blocks.0.short.conv.weight" in pr.state_dict()        # stage 2: plain 1x1
    p18 = ModelWrapper("presnet_18", [16, 16])
    assert sum(p.numel() for p in p18.parameters()) == 11_199_968 and p18.feature_channels["res5"] == 512
    mb = ModelWrapper("mobilenet_v2", [16, 16])   # counts from the reference's D2MobileNetV2() in this container
    assert sum(p.numel() for p in mb.parameters()) == 1_811_712 and len(mb.model.model.state_dict()) == 306
    assert mb.feature_channels == {"res2": 24, "res3": 32, "res4": 96, "res5": 320}
    assert "model.model.layer1.0.conv.1.norm.running_var" in mb.state_dict()   # expand_ratio 1: two convs only
    assert ModelWrapper("mobilenet_v2_os8", [16, 16]).model.model.layer4[0].conv[1].dilation == 2   # mobilenet_v2_wrapper.py:13-16
    with pytest.raises(NotImplementedError, match="MobileNet config"):
        ModelWrapper("mobilenet_v2_os4", [16, 16])
    # MobileNetV3 = torchvision's class (models/backbones/mobilenet_v3.py:10): its PUBLISHED parameter counts, 2 542 856 (small) and
    # 5 483 032 (large) at 1000 classes, minus 999 rows of the last Linear for the reference's num_classes=1
    m3s, m3l = ModelWrapper("mobilenet_v3_small", [16, 16]), ModelWrapper("mobilenet_v3_large", [16, 16])
    assert sum(p.numel() for p in m3s.parameters()) == 2_542_856 - 999 * 1025 and len(m3s.model.model.state_dict()) == 244
    assert sum(p.numel() for p in m3l.parameters()) == 5_483_032 - 999 * 1281 and len(m3l.model.model.state_dict()) == 312
    assert m3s.feature_channels == {"res5": 576, "res4": 96} and m3l.feature_channels == {"res5": 960, "res4": 160}  # :31-48
    sd3 = m3s.state_dict()
    assert "model.model.features.1.block.1.fc1.bias" in sd3 and "model.model.features.12.1.running_var" in sd3
    assert "model.model.classifier.3.weight" in sd3 and tuple(sd3["model.model.features.4.block.1.0.weight"].shape) == (96, 1, 5, 5)
    os8 = ModelWrapper("mobilenet_v3_large_os8", [16, 16]).model.model.features
    assert os8[13].block[1][0].dilation == (2, 2) and os8[13].block[1][0].stride == (1, 1) and os8[12].block[1][0].dilation == (1, 1)
    with pytest.raises(NotImplementedError, match="MobileNetV3 config"):
        ModelWrapper("mobilenet_v3_medium", [16, 16])
    s = ModelWrapper("stdc_2", [16, 16])
    assert abs(sum(p.numel() for p in s.parameters()) / 1e6 - 9.35) < 0.01 and len(s.state_dict()) == 636


def test_stage_block_range_logic():
    from dinov2_distillation_amd.train import DistillationModule

    class T(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.model = torch.nn.Module()
            self.calls = []
            self.model.blocks = [(lambda x, i=i: (self.calls.append(i), x)[1]) for i in range(n)]

    for n, expect in ((12, [9, 10]), (24, [18, 19, 20, 21, 22]), (40, list(range(30, 39)))):
        t = T(n)
        fake = type("M", (), {"teacher": t})()
        DistillationModule._forward_specific_stage(fake, torch.zeros(1), "res4")
        assert t.calls == expect
        t.calls.clear()
        DistillationModule._forward_specific_stage(fake, torch.zeros(1), "res2")
        DistillationModule._forward_specific_stage(fake, torch.zeros(1), "res3")
        assert t.calls == []


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from dinov2_distillation_amd.ddp import GradBucketReducer, allreduce_metrics, init_process_group_from_env
    os.environ["NCCL_P2P_DISABLE"] = "1"  # must be dropped, never inherited (train.py:23)
    init_process_group_from_env("gloo")
    assert "NCCL_P2P_DISABLE" not in os.environ
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 70), torch.nn.Linear(70, 5))
    from dinov2_distillation_amd import ops

    class DirectLinear(torch.autograd.Function):
        """the product's "direct" gradient path in miniature: backward writes dW/db into the arena views itself, announces
        them through ops.grad_done and returns None (autograd still runs the post-accumulate hook for such a None)."""
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x)
            ctx.params = (w, b)
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, dy):
            (x,) = ctx.saved_tensors
            w, b = ctx.params
            dx = dy @ w
            w.grad.copy_(dy.t() @ x)
            b.grad.copy_(dy.sum(0))
            ops.grad_done(w, b)
            return dx, None, None

    def forward_direct(x):
        h = torch.relu(net[0](x))
        h = DirectLinear.apply(h, net[2].weight, net[2].bias)
        return net[3](h)

    class FlatOpt:  # the arena layout FusedAdamW builds, on CPU tensors
        def __init__(self, params):
            ps = [p for p in params]
            offs, tot = [], 0
            for p in ps:
                offs.append(tot)
                tot += (p.numel() + 63) // 64 * 64
            self.flat_p, self.flat_g = torch.zeros(tot), torch.zeros(tot)
            self.layout = []
            for p, o in zip(ps, offs):
                self.flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + p.numel()].view(p.shape)
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
                self.layout.append((p, o, p.numel()))

    opt = FlatOpt(net.parameters())
    if rank == 1:
        with torch.no_grad():
            opt.flat_p.add_(1.0)  # diverge on purpose: broadcast must restore rank 0's weights
    red = GradBucketReducer(opt, bucket_bytes=33600 * 4)  # -> [w0 b0 w2 b2] [w3 b3]: the direct-written pair shares a bucket
    assert [b[2] for b in red.buckets] == [4, 2]
    red.broadcast_parameters(0)
    x = torch.randn(8, 40, generator=torch.Generator().manual_seed(100 + rank))
    for it in range(3):  # several steps: hooks re-arm; the last one mixes autograd-accumulated and directly written gradients
        opt.flat_g.zero_()
        (forward_direct(x) if it == 2 else net(x)).pow(2).sum().backward()
        assert any(red.launched), "buckets must be launched from the grad hooks during backward"
        red.finish()
    # reference: average of the per-rank gradients computed in one process
    ref_net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 70), torch.nn.Linear(70, 5))
    torch.manual_seed(0)
    ref_net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 70), torch.nn.Linear(70, 5))
    tot = [torch.zeros_like(p) for p in ref_net.parameters()]
    for r in range(world):
        ref_net.zero_grad()
        xr = torch.randn(8, 40, generator=torch.Generator().manual_seed(100 + r))
        ref_net(xr).pow(2).sum().backward()
        for t, p in zip(tot, ref_net.parameters()):
            t += p.grad / world
    err = max((p.grad - t).abs().max().item() for p, t in zip(net.parameters(), tot))
    m = allreduce_metrics({"a": torch.tensor(float(rank)), "b": torch.tensor(2.0)})
    q.put((rank, err, float(m["a"]), float(m["b"])))
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, a, b in res:
        assert err < 1e-5, (rank, err)
        assert abs(a - 0.5) < 1e-6 and abs(b - 2.0) < 1e-6


def test_student_checkpoint_warm_start_and_export_keys(tmp_path):
    """train/distillation_module.py:381-438 (per-family key remap of .pkl / .pth student checkpoints, numpy -> tensor,
    strict=False) and the key surface scripts/convert_to_anyma.py:22-31 exports from a trained module's state dict."""
    import pickle
    import numpy as np
    from dinov2_distillation_amd.config import make_config, set_student_dims
    from dinov2_distillation_amd.models import DINOv2ViT, ModelWrapper
    from dinov2_distillation_amd.train import DistillationModule

    def module(student_name, ckpt):
        cfg = make_config(student=student_name)
        cfg.student.checkpoint_path = str(ckpt) if ckpt else None
        student = ModelWrapper(student_name, cfg.teacher.n_patches, cfg.student.student_keys)
        set_student_dims(cfg, student)
        teacher = DINOv2ViT("dinov2_vits14", seed=0)
        teacher.model.blocks = teacher.model.blocks[:1]  # construction only: keep the CPU test light
        return DistillationModule(student, teacher, cfg)

    # STDC: detectron-style .pkl, numpy arrays, keys prefixed 'cp.backbone.' (distillation_module.py:413-415)
    ref = ModelWrapper("stdc_1", [16, 16])
    sd = {k[len("model.model."):]: v for k, v in ref.state_dict().items()}
    k0 = "features.0.conv.weight"
    w0 = np.full(tuple(sd[k0].shape), 0.25, dtype=np.float32)
    pkl = tmp_path / "STDCNet1.pkl"
    with open(pkl, "wb") as f:
        pickle.dump({"model": {"cp.backbone." + k0: w0, "cp.backbone.not_a_key": np.zeros(3, dtype=np.float32)}}, f)
    m = module("stdc_1", pkl)
    assert torch.equal(m.student.state_dict()["model.model." + k0], torch.from_numpy(w0))
    # ConvNeXt: .pth of tensors, keys un-prefixed (the final else branch, :422-424)
    cref = ModelWrapper("convnext_atto", [16, 16])
    ck = "downsample_layers.0.0.bias"
    pth = tmp_path / "convnext.pth"
    torch.save({ck: torch.full_like(cref.state_dict()["model.model." + ck], 3.0)}, pth)
    m2 = module("convnext_atto", pth)
    assert float(m2.student.state_dict()["model.model." + ck].min()) == 3.0
    with pytest.raises(ValueError, match="Unsupported checkpoint format"):
        bad = tmp_path / "weights.bin"
        bad.write_bytes(b"x")
        module("convnext_atto", bad)
    # a missing file ends construction as in the reference (:90-93 loads unconditionally, open() / torch.load raise) unless the
    # config opts out (make_config does, for synthetic runs)
    cfg_strict = make_config(student="convnext_atto")
    cfg_strict.student.allow_missing_checkpoint = False
    st_ = ModelWrapper("convnext_atto", cfg_strict.teacher.n_patches, cfg_strict.student.student_keys)
    set_student_dims(cfg_strict, st_)
    t_ = DINOv2ViT("dinov2_vits14", seed=0)
    t_.model.blocks = t_.model.blocks[:1]
    with pytest.raises(FileNotFoundError, match="student checkpoint"):
        DistillationModule(st_, t_, cfg_strict)              # falls back to the wrapper's ckpt_dir, which does not exist here
    cfg_strict.student.checkpoint_path = str(tmp_path / "nope.pth")
    with pytest.raises(FileNotFoundError, match="nope.pth"):
        DistillationModule(st_, t_, cfg_strict)
    # MobileNetV3: keys prefixed 'backbone.', the classification head is dropped before loading (distillation_module.py:419-421)
    m3 = ModelWrapper("mobilenet_v3_small", [16, 16])
    k3, kc = "features.0.0.weight", "classifier.3.bias"
    before = m3.state_dict()["model.model." + kc].clone()
    pth3 = tmp_path / "mobilenet_v3_small.pth"
    torch.save({"backbone." + k3: torch.full_like(m3.state_dict()["model.model." + k3], 0.5),
                "backbone." + kc: torch.full_like(before, 9.0)}, pth3)
    mm3 = module("mobilenet_v3_small", pth3)
    assert float(mm3.student.state_dict()["model.model." + k3].min()) == 0.5
    assert float(mm3.student.state_dict()["model.model." + kc]) != 9.0
    # export surface (convert_to_anyma.py:23-29): student backbone keys, projector / loss keys filtered out
    w_student = {k: v for k, v in m2.state_dict().items() if "student" in k and "scalekd" not in k and "feature_matchers" not in k}
    exported = {k.replace("student.model.model.", "backbone."): v for k, v in w_student.items()}
    assert exported and all(k.startswith("backbone.") for k in exported)
    assert "backbone.stages.3.1.pwconv2.weight" in exported and not any("projector" in k for k in exported)
    assert any(k.startswith("losses.scalekd_res4.projector_0") for k in m2.state_dict())


# ---- round 2: the Lightning branch, optimiser state, teacher broadcast, in-place gradient rules ------------------------------
def test_lightning_branch_drives_log_optimizers_and_fit_order():
    """train/distillation_module.py's ``L.LightningModule`` branch, driven the way /root/reference/train.py:209-213,256-288 drives
    the reference's module (stand-in ``lightning`` package, tests/lightning_standin.py; subprocess because the stand-in must be
    importable before the package is): configure_optimizers' dict layout, one optimizer with the reference's three parameter
    groups (distillation_module.py:457-479), the epoch-interval cosine scheduler, every metric name of
    ``_log_training_metrics`` with sync_dist=True plus the ``lr`` log that reads ``self.optimizers()``."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "lightning_standin.py"), "cpu"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["groups"] == ["student", "loss_scalekd_res4", "loss_scalekd_res5"]
    assert out["n_log_calls"] == 8 * 12 and out["sync_dist_all"]  # 8 micro-batches x (11 scalars + lr)
    assert "lr" in out["logged_names"] and "train_loss" in out["logged_names"]
    assert sum(n.startswith("train_scalekd_res") for n in out["logged_names"]) == 10
    assert out["global_step"] == 4  # 2 epochs x 4 micro-batches / accumulate 2
    assert out["lr_logged"][0] == pytest.approx(1e-3) and out["lr_logged"][-1] < 1e-3  # the scheduler stepped after epoch 0
    from oracle.step_oracle import cosine_lr
    assert out["final_lr"] == pytest.approx(cosine_lr(1e-3, 2), rel=1e-6)


def test_fused_adamw_state_dict_is_torch_adamw_layout_and_round_trips():
    """Lightning ModelCheckpoint / resume (train.py:83,284) go through optimizer.state_dict(): per-parameter step / exp_avg /
    exp_avg_sq in torch.optim.AdamW's layout, so a reference run's optimizer state loads into the fused optimiser and back."""
    from dinov2_distillation_amd.optim import FusedAdamW
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 10), torch.nn.Linear(10, 3))
    net[0].bias.requires_grad_(False)  # frozen parameters stay in the group but own no arena slot (distillation_module.py:457-461)
    ref_net = torch.nn.Sequential(torch.nn.Linear(6, 10), torch.nn.Linear(10, 3))
    ref_net.load_state_dict(net.state_dict())
    ref_net[0].bias.requires_grad_(False)
    ref = torch.optim.AdamW([{"params": list(ref_net.parameters())}], lr=1e-3, weight_decay=0.01)
    for i in range(3):  # a few real torch steps: non-trivial moments and step counters
        ref.zero_grad()
        ref_net(torch.randn(4, 6, generator=torch.Generator().manual_seed(i))).pow(2).sum().backward()
        ref.step()
    opt = FusedAdamW([{"params": list(net.parameters()), "name": "student"}], lr=1e-3)
    opt.load_state_dict(ref.state_dict())
    assert opt._steps == 3
    rs = ref.state_dict()["state"]
    for i, p in enumerate(net.parameters()):
        if not p.requires_grad:
            continue
        assert torch.equal(opt.state[p]["exp_avg"], rs[i]["exp_avg"]) and torch.equal(opt.state[p]["exp_avg_sq"], rs[i]["exp_avg_sq"])
        assert opt.state[p]["exp_avg"].data_ptr() >= opt.flat_m.data_ptr()  # views of the arena, not copies
    sd = opt.state_dict()
    assert set(sd["state"]) == set(rs) and all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in sd["state"].values())
    assert all(float(v["step"]) == 3.0 for v in sd["state"].values())
    assert "_range" not in sd["param_groups"][0] and sd["param_groups"][0]["params"] == [0, 1, 2, 3]
    opt.flat_m.add_(1.0)  # the exported state is a snapshot, not an alias of the live arena
    assert torch.equal(sd["state"][0]["exp_avg"], rs[0]["exp_avg"])
    opt2 = FusedAdamW([{"params": list(net.parameters()), "name": "student"}], lr=1e-3)
    opt2.load_state_dict(sd)
    assert opt2._steps == 3 and torch.equal(opt2.flat_v, opt.flat_v) and not torch.equal(opt2.flat_m, opt.flat_m)
    # torch's own AdamW accepts the export as well
    back = torch.optim.AdamW([{"params": list(ref_net.parameters())}], lr=1e-3)
    back.load_state_dict(sd)
    assert torch.equal(back.state_dict()["state"][2]["exp_avg_sq"], rs[2]["exp_avg_sq"])
    with pytest.raises(Exception, match="GPU"):
        opt.step()  # no CPU path: loud


def test_foreign_gradients_are_adopted_not_ignored():
    """``module.zero_grad()`` (set_to_none) detaches the arena views; a later backward allocates fresh .grad tensors.  The fused
    update must adopt them into the arena (and re-attach the views) instead of stepping on stale zeros."""
    from dinov2_distillation_amd.optim import FusedAdamW
    net = torch.nn.Linear(4, 4)
    opt = FusedAdamW(net.parameters(), lr=1e-3)
    net.zero_grad(set_to_none=True)
    assert net.weight.grad is None
    net(torch.ones(2, 4)).sum().backward()
    foreign = net.weight.grad.clone()
    assert net.weight.grad.data_ptr() != opt._grad_views[0].data_ptr()
    opt._reattach(adopt=True)
    assert net.weight.grad is opt._grad_views[0] and torch.equal(opt.flat_g[:16].view(4, 4), foreign)


def test_inplace_gradient_rules():
    """ops.can_async: in-place arena writes only (a) once per parameter between two zero_grad() calls - a second backward pass
    (accumulate_grad_batches, train.py:269) goes back to autograd, which ADDS - and (b) when the gradient exchange is aware
    of them (one rank, or ddp.GradBucketReducer registered as listener; never under a foreign DDP wrapper)."""
    from dinov2_distillation_amd import ops
    w = torch.nn.Parameter(torch.zeros(4, 4))
    w.grad = torch.zeros(4, 4)
    was = ops.ASYNC_WGRAD
    try:
        ops.ASYNC_WGRAD = True
        ops.mark_grads_fresh()
        assert ops.can_async(w)
        ops.claim(w)
        assert not ops.can_async(w)          # second write since zero_grad: accumulate through autograd
        ops.mark_grads_fresh()
        assert ops.can_async(w)
        ops.ASYNC_WGRAD = False
        assert not ops.can_async(w)
    finally:
        ops.ASYNC_WGRAD = was
        ops.mark_grads_fresh()


def _bcast_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo")
    from dinov2_distillation_amd import ops
    from dinov2_distillation_amd.ddp import GradBucketReducer
    from dinov2_distillation_amd.optim import FusedAdamW
    torch.manual_seed(0)

    class Step(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.teacher = torch.nn.Sequential(torch.nn.Linear(300, 1200), torch.nn.Linear(1200, 8))  # > 4 MiB? no: coalesced path
            self.big = torch.nn.Linear(1100, 1100)                                                     # 4.8 MB: direct path
            self.student = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.BatchNorm1d(8))
            for p in list(self.teacher.parameters()) + list(self.big.parameters()):
                p.requires_grad = False

    m = Step()
    opt = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    if rank == 1:  # a rank whose teacher checkpoint / BatchNorm statistics / student weights differ
        with torch.no_grad():
            for t in list(m.parameters()) + list(m.buffers()):
                t.add_(1) if t.dtype.is_floating_point else t.add_(5)
    red = GradBucketReducer(opt)
    # without a listener a multi-rank job must not write gradients in place (foreign DDP wrapper case)
    ops.ASYNC_WGRAD = True
    ops.remove_grad_listener(red._on_grad)
    no_listener = ops.exchange_ok()
    ops.add_grad_listener(red._on_grad)
    with_listener = ops.exchange_ok()
    ops.ASYNC_WGRAD = False
    red.broadcast_parameters(0, m)
    flat = torch.cat([t.detach().double().reshape(-1) for t in list(m.parameters()) + list(m.buffers())])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    q.put((rank, bool(torch.equal(gathered[0], gathered[1])), no_listener, with_listener,
           int(m.student[1].num_batches_tracked)))
    dist.destroy_process_group()


def test_teacher_and_buffers_are_broadcast_once_gloo_world2():
    """north_star: "teacher frozen, broadcast once".  Rank 1 starts with different teacher weights, BatchNorm statistics and
    student weights; after broadcast_parameters(0, module) every tensor of the module equals rank 0's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, equal, no_listener, with_listener, nbt in res:
        assert equal, rank
        assert no_listener is False and with_listener is True
        assert nbt == 0  # rank 1's +5 was overwritten by rank 0's value
