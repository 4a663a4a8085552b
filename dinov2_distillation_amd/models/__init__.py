from .dinov2 import DINOv2ViT
from .model_zoo import ModelWrapper

__all__ = ["DINOv2ViT", "ModelWrapper"]
