"""Seeded parameter recipe shared by tests/golden/make_golden.py (which loads the result into the REFERENCE model) and the
tests (which load it into the oracle / the MI355X model): full-width backbones would make the fixtures hundreds of MB if
their weights were stored, so the fixture keeps inputs, outputs, gradient digests and a checksum of these parameters instead.
Every tensor has its own generator (seed + crc32 of its state-dict key): independent of iteration order."""
import zlib

import torch


def seeded_state(shapes, seed):
    """shapes: {state-dict key: shape}.  conv weights ~ N(0, 2/fan_in); norm.weight 1 + 0.2 N; norm.bias 0.2 N;
    running_mean 0.1 N; running_var 1 + 0.1 U; num_batches_tracked 0; Linear weights ~ N(0, 1/fan_in); every other 1-D
    weight 1 + 0.2 N and bias 0.2 N (LayerNorm, Linear / conv biases)."""
    out = {}
    for k, shape in shapes.items():
        g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(k.encode()))
        shape = tuple(shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shape, dtype=torch.long)
        elif k.endswith("running_mean"):
            out[k] = 0.1 * torch.randn(shape, generator=g)
        elif k.endswith("running_var"):
            out[k] = 1.0 + 0.1 * torch.rand(shape, generator=g)
        elif k.endswith("norm.weight"):
            out[k] = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif k.endswith("norm.bias"):
            out[k] = 0.2 * torch.randn(shape, generator=g)
        elif len(shape) == 4:  # (depthwise weights are [C, 1, k, k]: fan_in = k * k)
            fan_in = shape[1] * shape[2] * shape[3]
            out[k] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif len(shape) == 2:  # Linear weight [out, in]
            out[k] = torch.randn(shape, generator=g) * (1.0 / shape[1]) ** 0.5
        elif len(shape) == 1 and k.endswith("weight"):  # LayerNorm weight (keys like norm1.weight, attn.norm.weight)
            out[k] = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif len(shape) == 1 and k.endswith("bias"):
            out[k] = 0.2 * torch.randn(shape, generator=g)
        else:
            raise KeyError(f"no recipe for {k} {shape}")
    return out


def checksum(state):
    return float(sum(v.double().sum() for v in state.values() if v.dtype.is_floating_point))


def digest(t):
    """what the fixture keeps of a large gradient: its L2 norm, its sum and its first 64 values"""
    f = t.detach().reshape(-1).double()
    return torch.cat([f.norm().reshape(1), f.sum().reshape(1), f[:64]]).float()
