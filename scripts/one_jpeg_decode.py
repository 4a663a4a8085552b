"""A few decodes of one batch of 64 synthetic 500 x 375 photographs (no restart markers) - the target of scripts/pmc_jpeg.sh"""
import io
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dinov2_distillation_amd.data import JpegDecoderGpu  # noqa: E402

rng = np.random.default_rng(0)
files = []
for i in range(8):
    yy, xx = np.mgrid[0:375, 0:500]
    img = np.stack([127 + 100 * np.sin(xx / (9.0 + i) + yy / 17.0), 127 + 100 * np.cos(xx / 23.0 - i), 127 + 90 * np.sin(yy / 7.0 + xx / 31.0)], 2)
    buf = io.BytesIO()
    Image.fromarray(np.clip(img + rng.normal(0, 10.0, img.shape), 0, 255).astype(np.uint8)).save(buf, format="JPEG", quality=90)
    files.append(buf.getvalue())
files = files * 8
dec = JpegDecoderGpu()
for _ in range(3):
    dec.decode(files)
torch.cuda.synchronize()
print("bytes per file", sum(map(len, files)) // len(files))
