from .distillation_module import LOSS_REGISTRY, DistillationModule

__all__ = ["DistillationModule", "LOSS_REGISTRY"]
