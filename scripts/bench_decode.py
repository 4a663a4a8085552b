"""Host JPEG decode rate (SURVEY 8f-3's first word: /root/reference/datasets/CustomDataset.py:64 `Image.open(path).convert('RGB')`), the
part of the input pipeline that is NOT on the GPU here: images/s per host core, next to the step rates it has to feed.
    python scripts/bench_decode.py [seconds]
Synthetic ImageNet-like files (500 x 375, quality 90, natural-image-like spectrum: ~110 KB each) are encoded once in memory; the timed
loop is exactly the reference's decode call on one core.  Also timed: PIL's DCT-domain `draft` down-scaling (decode at >= 2x the 224 crop),
which the GPU augment's RandomResizedCrop could sit behind."""
import io
import os
import sys
import time

import numpy as np
from PIL import Image

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
rng = np.random.default_rng(0)


def natural_like(h, w):
    """1/f-spectrum noise + smooth gradients: compresses like a photograph"""
    f = np.fft.rfft2(rng.standard_normal((3, h, w)))
    ky, kx = np.meshgrid(np.fft.fftfreq(h), np.fft.rfftfreq(w), indexing="ij")
    f /= np.maximum(np.sqrt(ky ** 2 + kx ** 2), 1.0 / max(h, w)) ** 1.1
    img = np.fft.irfft2(f, s=(h, w))
    img = (img - img.min()) / (img.max() - img.min())
    return (img.transpose(1, 2, 0) * 255).astype(np.uint8)


files = []
for _ in range(32):
    buf = io.BytesIO()
    Image.fromarray(natural_like(375, 500)).save(buf, format="JPEG", quality=90)
    files.append(buf.getvalue())
kb = sum(len(f) for f in files) / len(files) / 1024


def rate(fn):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for f in files:
            fn(f)
        n += len(files)
    return n / (time.perf_counter() - t0)


def full(f):
    return Image.open(io.BytesIO(f)).convert("RGB").load()


def draft(f):
    im = Image.open(io.BytesIO(f))
    im.draft("RGB", (320, 320))  # DCT-domain down-scaling to the smallest size >= the request (here 1/1: the image is under 2x)
    return im.convert("RGB").load()


r_full, r_draft = rate(full), rate(draft)
cores = len(os.sched_getaffinity(0))
print(f"JPEG 500x375 q90, {kb:.0f} KB/file, one core ({cores} in this process's affinity mask):")
print(f"  Image.open(...).convert('RGB')           {r_full:7.1f} images/s per core")
print(f"  ... with draft('RGB', (320, 320)) first   {r_draft:7.1f} images/s per core")
for name, need in (("fp32 step (bench.py value)", 1784.0), ("whole step under the autocast policy", 3228.0)):
    print(f"  {name}: {need:.0f} images/s -> {need / r_full:.1f} decode cores")
