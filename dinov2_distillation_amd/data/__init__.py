from .augment import DataAugmentationDINOGpu, make_plan  # noqa: F401
from .jpeg import CustomDatasetFiles, DecodedBatch, GpuInputPipeline, JpegDecoderGpu, UnsupportedJpeg, parse_jpeg  # noqa: F401
