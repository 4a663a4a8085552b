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

# ---- the same files through the GPU decoder (csrc/jpeg.hip) ----------------------------------------------------------------------
try:
    import torch
    have_gpu = torch.cuda.is_available()
except Exception:
    have_gpu = False
if have_gpu:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dinov2_distillation_amd.data.jpeg import JpegDecoderGpu, parse_jpeg
    batch = (files * 2)[:64]
    for f, im in zip(batch[:4], JpegDecoderGpu().decode(batch[:4]).images()):
        assert np.array_equal(im.cpu().numpy(), np.asarray(Image.open(io.BytesIO(f)).convert("RGB")))
    t0 = time.perf_counter()
    for f in batch:
        parse_jpeg(f)
    t_parse = (time.perf_counter() - t0) / len(batch)
    print(f"GPU decode, batch of {len(batch)} (bit-exact with Pillow on the first four):")
    print(f"  host side, marker walk only               {t_parse * 1e6:7.1f} us per file on one core")
    for label, fs in (("no restart markers: one wavefront per image", batch),):
        dec = JpegDecoderGpu()
        dec.decode(fs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n, t0 = 0, time.perf_counter()
        e0.record()
        while time.perf_counter() - t0 < secs:
            dec.decode(fs)
            n += 1
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n
        print(f"  {label}: {wall * 1e3:6.2f} ms per batch end to end ({len(fs) / wall:7.0f} images/s; GPU timeline "
              f"{e0.elapsed_time(e1) / n:6.2f} ms per batch)")
    # the same pictures re-encoded with one restart interval per MCU row: one wavefront per row
    rows = []
    for f in batch:
        buf = io.BytesIO()
        Image.open(io.BytesIO(f)).save(buf, format="JPEG", quality=90, restart_marker_rows=1)
        rows.append(buf.getvalue())
    dec = JpegDecoderGpu()
    dec.decode(rows)
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        dec.decode(rows)
        n += 1
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    print(f"  one restart interval per MCU row ({dec.stats['segments'] // dec.stats['images']} wavefronts per image): {wall * 1e3:6.2f} ms per batch "
          f"({len(rows) / wall:7.0f} images/s)")
