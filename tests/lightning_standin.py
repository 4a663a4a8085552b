"""TEST INFRASTRUCTURE: a stand-in for the ``lightning`` package (absent from this image) that is just big enough to drive the
``L.LightningModule`` branch of dinov2_distillation_amd.train.distillation_module the way the reference's caller does
(/root/reference/train.py:209-213 constructs the module, :256-270 the Trainer: automatic optimisation, gradient_clip_val 1.0
by norm, accumulate_grad_batches, one optimizer + an epoch-interval scheduler from configure_optimizers).

    python tests/lightning_standin.py cpu|gpu      -> prints one JSON line (used by the tests through a subprocess, because the
                                                      stand-in must be in sys.modules BEFORE the package is imported)

The fit loop follows Lightning's documented order for automatic optimisation: training_step -> (loss / accumulate).backward()
-> on_after_backward -> [every accumulate-th batch] clip_grad_norm_ -> optimizer.step() -> optimizer.zero_grad();
lr_scheduler.step() at the end of an epoch.  Logged values arrive through ``self.log`` exactly as the module sends them.
"""
import json
import os
import sys
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def install():
    L = types.ModuleType("lightning")

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.trainer = None
            self.hparams_saved = None
            self.log_calls = []

        def save_hyperparameters(self, *a, **k):
            self.hparams_saved = a[0] if a else k

        def log(self, name, value, **kw):
            self.log_calls.append((name, float(value.detach()) if torch.is_tensor(value) else float(value), dict(kw)))

        def optimizers(self):
            return self.trainer.optimizer

        def on_after_backward(self):
            pass

    class Trainer:
        def __init__(self, max_epochs=1, gradient_clip_val=None, gradient_clip_algorithm="norm", accumulate_grad_batches=1, **kw):
            assert gradient_clip_algorithm == "norm"
            self.max_epochs, self.clip, self.accumulate = max_epochs, gradient_clip_val, accumulate_grad_batches
            self.optimizer = self.scheduler = None
            self.global_step = 0

        def fit(self, module, batches):
            module.trainer = self
            conf = module.configure_optimizers()
            if isinstance(conf, dict):
                self.optimizer = conf["optimizer"]
                sch = conf.get("lr_scheduler")
                self.scheduler = sch["scheduler"] if isinstance(sch, dict) else sch
            else:
                self.optimizer = conf
            module.train()
            self.optimizer.zero_grad()
            for _ in range(self.max_epochs):
                for i, batch in enumerate(batches):
                    loss = module.training_step(batch, i)
                    (loss / self.accumulate).backward()
                    module.on_after_backward()
                    if (i + 1) % self.accumulate == 0:
                        if self.clip is not None:
                            torch.nn.utils.clip_grad_norm_([p for g in self.optimizer.param_groups for p in g["params"]
                                                            if p.grad is not None], self.clip)
                        self.optimizer.step()
                        self.optimizer.zero_grad()
                        self.global_step += 1
                if self.scheduler is not None:
                    self.scheduler.step()

    L.LightningModule, L.Trainer = LightningModule, Trainer
    sys.modules["lightning"] = L
    return L


def main(mode):
    sys.path.insert(0, ROOT)
    L = install()
    from dinov2_distillation_amd.config import make_config, set_student_dims
    from dinov2_distillation_amd.models import DINOv2ViT, ModelWrapper
    from dinov2_distillation_amd.train import distillation_module as dm
    assert dm._Base is L.LightningModule and issubclass(dm.DistillationModule, L.LightningModule)
    sname = "convnext_atto" if mode == "cpu" else "convnext_tiny"
    cfg = make_config(student=sname)
    cfg.optimizer.scheduler = dict(type="CosineAnnealingLR", kwargs=dict(T_max=50, eta_min=1e-5), monitor="val_loss",
                                   interval="epoch", frequency=1)  # config/config.yaml:31-38
    torch.manual_seed(1)
    student = ModelWrapper(sname, cfg.teacher.n_patches, cfg.student.student_keys)
    set_student_dims(cfg, student)
    teacher = DINOv2ViT("dinov2_vits14", seed=0)
    out = {"mode": mode}
    if mode == "cpu":
        teacher.model.blocks = teacher.model.blocks[:1]  # construction only
        cfg.optimizer.type = "Adam"  # FusedAdamW steps on the GPU only (it raises on CPU tensors); the gpu mode uses it
        mod = dm.DistillationModule(student, teacher, cfg)
        assert mod.hparams_saved is cfg
        # the hot path needs the GPU: stand in a differentiable CPU "step" that touches every trainable parameter and emits
        # the reference's 11 scalars, so that log / optimizers() / configure_optimizers / the fit order run for real
        names = ["loss"] + [f"scalekd_{l}_{k}" for l in ("res4", "res5") for k in
                            ("total_loss", "frequency_loss", "spatial_loss", "spatial_similarity", "frequency_similarity")]

        def fake_losses(features):
            tot = sum((p * p).sum() for p in mod.parameters() if p.requires_grad) * features
            return {k: tot for k in names}

        mod._compute_losses = fake_losses
        mod._extract_features = lambda batch: batch.mean()
        tr = L.Trainer(max_epochs=2, gradient_clip_val=1.0, accumulate_grad_batches=2)
        tr.fit(mod, [torch.full((2, 3), 0.5), torch.full((2, 3), 1.5), torch.full((2, 3), 1.0), torch.full((2, 3), 2.0)])
        out["n_log_calls"] = len(mod.log_calls)
        out["logged_names"] = sorted({n for n, _, _ in mod.log_calls})
        out["sync_dist_all"] = all(kw.get("sync_dist") for _, _, kw in mod.log_calls)
        out["lr_logged"] = [v for n, v, _ in mod.log_calls if n == "lr"]
        out["optimizer"] = type(tr.optimizer).__name__
        out["groups"] = [g["name"] for g in tr.optimizer.param_groups]
        out["global_step"] = tr.global_step
        out["final_lr"] = tr.optimizer.param_groups[0]["lr"]
    else:
        from dinov2_distillation_amd import ops
        from dinov2_distillation_amd.trainer import DistillationStepper
        torch.cuda.set_device(0)
        mod = dm.DistillationModule(student, teacher, cfg).cuda()
        xs = [torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(40 + i)).cuda() for i in range(4)]
        tr = L.Trainer(max_epochs=1, gradient_clip_val=1.0, accumulate_grad_batches=2)
        tr.fit(mod, xs)
        torch.cuda.synchronize()
        out["losses"] = [v for n, v, _ in mod.log_calls if n == "train_loss"]
        out["global_step"] = tr.global_step
        # the same four micro-batches through the build's own stepper (fused clip + AdamW, accumulate 2)
        torch.manual_seed(1)
        st = DistillationStepper(student=sname, device="cuda:0", accumulate_grad_batches=2)
        ref_losses = [float(st.step(x).detach()) for x in xs]
        torch.cuda.synchronize()
        a = torch.cat([p.detach().reshape(-1) for p in mod.student.parameters()] + [p.detach().reshape(-1) for p in mod.losses.parameters()])
        b = torch.cat([p.detach().reshape(-1) for p in st.module.student.parameters()] +
                      [p.detach().reshape(-1) for p in st.module.losses.parameters()])
        out["ref_losses"] = ref_losses
        out["param_rel_diff"] = ((a - b).norm() / b.norm()).item()
        out["async_wgrad"] = bool(ops.ASYNC_WGRAD)
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cpu")
