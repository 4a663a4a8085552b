import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_augment_cpu import _images
from dinov2_distillation_amd.data import DataAugmentationDINOGpu
from oracle import augment_oracle as O
aug = DataAugmentationDINOGpu()
images = _images(0)
for op in ("AutoContrast", "Equalize", "Contrast"):
    plans = [dict(crop=(0, 0, im.shape[0], im.shape[1]), flip=False, ops=[(op, 0.27 if op == "Contrast" else 0.0)], erase=None) for im in images]
    out, stages = aug(images, plans=plans, return_stages=True)
    d = aug._debug
    for b, im in enumerate(images):
        ref, rs = O.run_plan(im, plans[b], return_stages=True)
        src = rs[0]
        ok0 = np.array_equal(stages[0][b, :, :, :3].cpu().numpy(), src)
        hist = np.stack([np.bincount(src[..., c].ravel(), minlength=256) for c in range(3)])
        gh = d["hist"][b].cpu().numpy().reshape(3, 256)
        print(op, "image", b, "resize ok", ok0, "hist equal", np.array_equal(hist, gh), "stage equal", np.array_equal(stages[1][b, :, :, :3].cpu().numpy(), rs[1]))
        if op == "AutoContrast":
            for c in range(3):
                h = hist[c]; lo = int(np.nonzero(h)[0][0]); hi = int(np.nonzero(h)[0][-1])
                scale = 255.0 / (hi - lo); offset = -lo * scale
                lut = np.array([min(255, max(0, int(ix * scale + offset))) for ix in range(256)])
                gl = d["lut"][b].cpu().numpy().reshape(3, 256)[c]
                bad = np.nonzero(lut != gl)[0]
                print("   channel", c, "lo hi", lo, hi, "lut mismatches", len(bad), [(int(i), int(lut[i]), int(gl[i]), repr(i * scale + offset)) for i in bad[:6]])
