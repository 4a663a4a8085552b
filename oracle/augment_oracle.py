"""Oracle (TEST INFRASTRUCTURE): the reference's training-time input pipeline, CPU restatement on PIL.

Restates /root/reference/datasets/augmentations.py:23-78 (``DataAugmentationDINO``: RandomResizedCrop(bicubic) -> horizontal
flip -> RandAugment(num_ops 9, magnitude 9, 31 bins, bilinear) -> ToTensor -> Normalize(ImageNet) -> RandomErasing(p .25)) and the
``Image.open(...).convert('RGB')`` hand-over of datasets/CustomDataset.py:53-69.

Those transforms live in torchvision (third-party, not vendored under /root/reference, absent from this image; the reference
pins no version, the formulas below are those of torchvision 0.15-0.20's ``transforms`` / ``transforms.functional`` /
``transforms.autoaugment``, which have not changed across those releases).  torchvision's PIL branch is a thin layer over
Pillow calls, and Pillow IS installed here, so the split is:
  * PIXEL ARITHMETIC - every resize / affine / enhance / LUT step is executed by Pillow itself through exactly the call
    torchvision makes (``img.crop().resize(BICUBIC)``, ``img.transform(AFFINE, BILINEAR)``, ``img.rotate``,
    ``ImageEnhance.*``, ``ImageOps.*``): pinned by construction, the GPU pipeline is held to bit equality against it;
  * PARAMETER SAMPLING - which crop box / op / sign / erase box is drawn from which RNG call is restated from torchvision's
    published source (``RandomResizedCrop.get_params``, ``RandAugment.forward`` + ``_augmentation_space``,
    ``RandomErasing.get_params``), driven by an explicit ``torch.Generator``: PARITY UNPINNED for the draw order (no
    torchvision to run), deterministic and shared with the product's planner, which is tested against this file draw by draw.
"""
import math

import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageOps

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # augmentations.py:12-13
IMAGENET_STD = (0.229, 0.224, 0.225)
OPS = ("Identity", "ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate", "Brightness", "Color", "Contrast", "Sharpness",
       "Posterize", "Solarize", "AutoContrast", "Equalize")  # RandAugment._augmentation_space, in dict order
SIGNED = {"ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate", "Brightness", "Color", "Contrast", "Sharpness"}


def _uniform(gen, lo, hi):
    return torch.empty(1).uniform_(lo, hi, generator=gen).item()


def _randint(gen, n):
    return int(torch.randint(n, (1,), generator=gen).item())


def rrc_params(gen, height, width, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """RandomResizedCrop.get_params -> (top, left, h, w)"""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * _uniform(gen, scale[0], scale[1])
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1], generator=gen)).item()
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = _randint(gen, height - h + 1)
            j = _randint(gen, width - w + 1)
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def magnitude_of(op, magnitude_bin, num_bins, height, width):
    """_augmentation_space(num_bins, image_size)[op][0][magnitude]: float32 linspace values read back as Python floats"""
    lin = lambda a, b: float(torch.linspace(a, b, num_bins)[magnitude_bin].item())
    if op in ("ShearX", "ShearY"):
        return lin(0.0, 0.3)
    if op == "TranslateX":
        return lin(0.0, 150.0 / 331.0 * width)
    if op == "TranslateY":
        return lin(0.0, 150.0 / 331.0 * height)
    if op == "Rotate":
        return lin(0.0, 30.0)
    if op in ("Brightness", "Color", "Contrast", "Sharpness"):
        return lin(0.0, 0.9)
    if op == "Posterize":
        return float((8 - (torch.arange(num_bins) / ((num_bins - 1) / 4)).round().int())[magnitude_bin].item())
    if op == "Solarize":
        return lin(255.0, 0.0)
    return 0.0


def randaugment_plan(gen, height, width, num_ops=9, magnitude=9, num_bins=31):
    """RandAugment.forward's draws: [(op name, signed magnitude)] * num_ops"""
    plan = []
    for _ in range(num_ops):
        op = OPS[_randint(gen, len(OPS))]
        mag = magnitude_of(op, magnitude, num_bins, height, width)
        if op in SIGNED and _randint(gen, 2):
            mag *= -1.0
        plan.append((op, mag))
    return plan


def inverse_affine_matrix(center, angle, translate, scale, shear):
    """transforms.functional._get_inverse_affine_matrix"""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    matrix = [d, -b, 0.0, -c, a, 0.0]
    matrix = [x / scale for x in matrix]
    matrix[2] += matrix[0] * (-cx - tx) + matrix[1] * (-cy - ty)
    matrix[5] += matrix[3] * (-cx - tx) + matrix[4] * (-cy - ty)
    matrix[2] += cx
    matrix[5] += cy
    return matrix


def pil_rotate_matrix(width, height, angle):
    """the matrix Image.rotate(angle, expand=False, center=None) hands to Image.transform (Pillow's Image.py)"""
    angle = angle % 360.0
    cx, cy = width / 2.0, height / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return m


def affine_matrix_of(op, mag, width, height):
    """the inverse-map matrix of a geometric RandAugment op (autoaugment._apply_op -> F.affine / F.rotate, PIL branch)"""
    if op == "ShearX":
        return inverse_affine_matrix([0, 0], 0.0, [0, 0], 1.0, [math.degrees(math.atan(mag)), 0.0])
    if op == "ShearY":
        return inverse_affine_matrix([0, 0], 0.0, [0, 0], 1.0, [0.0, math.degrees(math.atan(mag))])
    if op == "TranslateX":
        return inverse_affine_matrix([width * 0.5, height * 0.5], 0.0, [int(mag), 0], 1.0, [0.0, 0.0])
    if op == "TranslateY":
        return inverse_affine_matrix([width * 0.5, height * 0.5], 0.0, [0, int(mag)], 1.0, [0.0, 0.0])
    if op == "Rotate":
        if mag % 360.0 in (0.0, 180.0, 90.0, 270.0):  # Image.rotate's exact fast paths (never hit at magnitude 9)
            raise NotImplementedError("right-angle rotation")
        return pil_rotate_matrix(width, height, mag)
    raise KeyError(op)


def apply_op(img, op, mag):
    """autoaugment._apply_op on a PIL image: the Pillow call torchvision's PIL branch makes for each op"""
    w, h = img.size
    if op in ("ShearX", "ShearY", "TranslateX", "TranslateY"):
        return img.transform((w, h), Image.AFFINE, affine_matrix_of(op, mag, w, h), Image.BILINEAR)
    if op == "Rotate":
        return img.rotate(mag, Image.BILINEAR)
    if op == "Brightness":
        return ImageEnhance.Brightness(img).enhance(1.0 + mag)
    if op == "Color":
        return ImageEnhance.Color(img).enhance(1.0 + mag)
    if op == "Contrast":
        return ImageEnhance.Contrast(img).enhance(1.0 + mag)
    if op == "Sharpness":
        return ImageEnhance.Sharpness(img).enhance(1.0 + mag)
    if op == "Posterize":
        return ImageOps.posterize(img, int(mag))
    if op == "Solarize":
        return ImageOps.solarize(img, mag)
    if op == "AutoContrast":
        return ImageOps.autocontrast(img)
    if op == "Equalize":
        return ImageOps.equalize(img)
    return img


def erase_params(gen, height, width, p=0.25, scale=(0.02, 1.0 / 3.0), ratio=(0.3, 3.3)):
    """RandomErasing.forward + get_params (value = 0) -> (top, left, h, w) or None"""
    if not (torch.rand(1, generator=gen).item() < p):
        return None
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        erase_area = area * _uniform(gen, scale[0], scale[1])
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1], generator=gen)).item()
        h = int(round(math.sqrt(erase_area * aspect_ratio)))
        w = int(round(math.sqrt(erase_area / aspect_ratio)))
        if not (h < height and w < width):
            continue
        i = _randint(gen, height - h + 1)
        j = _randint(gen, width - w + 1)
        return i, j, h, w
    return None


def sample_plan(gen, height, width, size=224, scale=(0.08, 1.0)):
    """every random decision DataAugmentationDINO.__call__ takes for ONE image, in the order the transforms run"""
    crop = rrc_params(gen, height, width, scale)
    flip = bool(torch.rand(1, generator=gen).item() < 0.5)
    ops = randaugment_plan(gen, size, size)
    erase = erase_params(gen, size, size)
    return dict(crop=crop, flip=flip, ops=ops, erase=erase)


def run_plan(img_u8, plan, size=224, return_stages=False):
    """img_u8: [H, W, 3] uint8 array (an RGB image as ``Image.open().convert('RGB')`` delivers it) -> float32 [3, size, size]"""
    img = Image.fromarray(np.ascontiguousarray(img_u8))
    i, j, h, w = plan["crop"]
    img = img.crop((j, i, j + w, i + h)).resize((size, size), Image.BICUBIC)  # F.resized_crop, PIL branch
    if plan["flip"]:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    stages = [np.asarray(img).copy()]
    for op, mag in plan["ops"]:
        img = apply_op(img, op, mag)
        stages.append(np.asarray(img).copy())
    t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).to(torch.float32).div(255)  # ToTensor
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32)[:, None, None]
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32)[:, None, None]
    t = t.sub(mean).div(std)  # Normalize
    if plan["erase"] is not None:
        ei, ej, eh, ew = plan["erase"]
        t[:, ei:ei + eh, ej:ej + ew] = 0.0
    return (t, stages) if return_stages else t
