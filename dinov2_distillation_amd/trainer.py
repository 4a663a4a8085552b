"""Minimal step driver for the hot path (what Lightning's fit loop does around ``training_step`` in the reference:
zero_grad -> training_step -> backward (+ DDP exchange) -> clip -> optimizer.step; train.py:256-288)."""
import torch
import torch.distributed as dist

from .config import make_config, set_student_dims
from .ddp import GradBucketReducer
from .models import DINOv2ViT, ModelWrapper
from .train import DistillationModule


class DistillationStepper:
    def __init__(self, teacher="dinov2_vits14", student="convnext_tiny", crop=224, device="cuda", heads=(16, 24),
                 clip=1.0, lr=1e-3, seed=1, bucket_bytes=32 << 20, teacher_precision=None):
        self.cfg = make_config(teacher, student, crop, heads=heads, lr=lr)
        torch.manual_seed(seed)  # student seed 1, ScaleKD follows; identical on every rank
        self.student = ModelWrapper(student, self.cfg.teacher.n_patches, self.cfg.student.student_keys)
        set_student_dims(self.cfg, self.student)
        self.teacher = DINOv2ViT(teacher, seed=0, precision=teacher_precision)
        self.module = DistillationModule(self.student, self.teacher, self.cfg).to(device).train()
        self.optimizer = self.module.configure_optimizers()
        if isinstance(self.optimizer, dict):
            self.optimizer = self.optimizer["optimizer"]
        self.reducer = GradBucketReducer(self.optimizer, bucket_bytes) if dist.is_initialized() else None
        if self.reducer is not None:
            self.reducer.broadcast_parameters(0)
        self.clip = clip
        self.step_idx = 0

    def step(self, batch):
        self.optimizer.zero_grad()
        loss = self.module.training_step(batch, self.step_idx)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.step(clip_grad_norm=self.clip)
        self.step_idx += 1
        return loss
