from .augment import DataAugmentationDINOGpu, make_plan  # noqa: F401
