"""Host side of the GPU input pipeline and its oracle, on the CPU: the planner draws what the oracle's restatement
of torchvision's transforms draws, and the tabulated resize coefficients are Pillow's (checked through Pillow's own output)."""
import numpy as np
import torch
from PIL import Image


def _images(seed=0):
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, size=(97, 131, 3), dtype=np.uint8)
    big = rng.integers(0, 256, size=(40, 52, 3), dtype=np.uint8)
    smooth = np.asarray(Image.fromarray(big).resize((400, 300), Image.BICUBIC))
    tall = np.asarray(Image.fromarray(big).resize((180, 500), Image.BILINEAR))
    return [noise, smooth, tall]


def test_planner_draws_what_the_oracle_draws():
    from dinov2_distillation_amd.data import augment as A
    from oracle import augment_oracle as O
    seen = set()
    for seed in range(12):
        g1, g2 = torch.Generator().manual_seed(seed), torch.Generator().manual_seed(seed)
        for (h, w) in [(300, 400), (97, 131), (640, 480), (20, 900)]:
            p1, p2 = A.make_plan(g1, h, w), O.sample_plan(g2, h, w)
            assert p1 == p2
            assert torch.equal(g1.get_state(), g2.get_state())  # same number of draws consumed
            for op, mag in p1["ops"]:
                seen.add(op)
                code, f = A._op_params(op, mag, 224)
                if code == A.AUG_AFFINE:
                    assert list(f[:6]) == O.affine_matrix_of(op, mag, 224, 224)
    assert seen == set(O.OPS)  # every op of the RandAugment space was drawn at least once
    assert A.OP_NAMES == O.OPS and abs(A._magnitude("ShearX", 9, 31, 224) - 0.09) < 1e-6 and A._magnitude("Posterize", 9, 31, 224) == 7.0


def test_resize_coefficient_tables_reproduce_pillow_bit_for_bit():
    """Pillow's bicubic resize = a horizontal then a vertical pass of 22-bit fixed-point taps (libImaging/Resample.c); with the
    tables of data.augment.resample_coeffs a plain integer resampler must return Pillow's image exactly"""
    from dinov2_distillation_amd.data.augment import PRECISION_BITS, resample_coeffs
    rng = np.random.default_rng(3)
    for (H, W, S) in [(300, 400, 224), (97, 131, 224), (224, 224, 224), (50, 1000, 518), (640, 480, 37)]:
        img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        kh, bh, _ = resample_coeffs(W, S)
        kv, bv, _ = resample_coeffs(H, S)
        src = img.astype(np.int64)
        tmp = np.zeros((H, S, 3), np.int64)
        for xx in range(S):
            x0, n = bh[xx]
            tmp[:, xx] = np.clip(((1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n] * kh[xx, :n, None]).sum(1)) >> PRECISION_BITS, 0, 255)
        out = np.zeros((S, S, 3), np.uint8)
        for yy in range(S):
            y0, n = bv[yy]
            out[yy] = np.clip(((1 << (PRECISION_BITS - 1)) + (tmp[y0:y0 + n] * kv[yy, :n, None, None]).sum(0)) >> PRECISION_BITS, 0, 255)
        assert np.array_equal(out, np.asarray(Image.fromarray(img).resize((S, S), Image.BICUBIC))), (H, W, S)


def test_batched_coefficient_tables_equal_the_scalar_ones():
    from dinov2_distillation_amd.data.augment import resample_coeffs, resample_coeffs_batch
    sizes = [400, 131, 1000, 224, 37, 500, 97, 3]
    kb, bb, K = resample_coeffs_batch(sizes, 224)
    for i, n in enumerate(sizes):
        k, b, ks = resample_coeffs(n, 224)
        assert ks <= K and np.array_equal(kb[i, :, :ks], k) and np.array_equal(bb[i], b) and not kb[i, :, ks:].any()


def test_oracle_pipeline_runs_every_stage():
    from oracle import augment_oracle as O
    gen = torch.Generator().manual_seed(5)
    for img in _images():
        plan = O.sample_plan(gen, img.shape[0], img.shape[1])
        t, stages = O.run_plan(img, plan, return_stages=True)
        assert t.shape == (3, 224, 224) and t.dtype == torch.float32 and len(stages) == 10
        if plan["erase"] is not None:
            i, j, h, w = plan["erase"]
            assert float(t[:, i:i + h, j:j + w].abs().max()) == 0.0
    # an identity plan is ToTensor + Normalize of the resized crop
    img = _images()[1]
    plan = dict(crop=(0, 0, img.shape[0], img.shape[1]), flip=False, ops=[("Identity", 0.0)], erase=None)
    t = O.run_plan(img, plan)
    ref = torch.from_numpy(np.asarray(Image.fromarray(img).resize((224, 224), Image.BICUBIC)).copy()).permute(2, 0, 1).float() / 255
    mean, std = torch.tensor(O.IMAGENET_MEAN)[:, None, None], torch.tensor(O.IMAGENET_STD)[:, None, None]
    assert torch.equal(t, (ref - mean) / std)
