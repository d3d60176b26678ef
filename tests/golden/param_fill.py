"""Deterministic parameter/buffer fill shared by the golden generator and the tests.

Weights are not stored in the fixtures: the reference model (in make_golden.py), the CPU
oracle and the HIP modules all fill every state_dict entry from a seed derived from the
entry's NAME, so equal names => equal values on every side (names are the reference's
checkpoint contract).
"""
import zlib

import torch


def fill_state(module, seed=0, scale=1.0):
    sd = module.state_dict()
    out = {}
    for name, t in sd.items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros_like(t)
        elif name.endswith("running_var"):
            out[name] = torch.rand(t.shape, generator=g) * 0.5 + 0.75
        elif name.endswith("running_mean"):
            out[name] = torch.randn(t.shape, generator=g) * 0.1
        elif t.dim() >= 2:  # weight matrices: fan-in scaled
            fan_in = t.shape[1]
            out[name] = torch.randn(t.shape, generator=g) * (scale / fan_in ** 0.5)
        elif name.endswith("weight"):  # norm gains
            out[name] = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
        else:  # biases
            out[name] = 0.1 * torch.randn(t.shape, generator=g)
    module.load_state_dict(out, strict=True)
    return module


def unit_cloud(B, N, seed, C=3):
    """uniform(-1,1) cloud, centred and scaled into the unit sphere (mirrors pc_normalize,
    dataset/ModelNetDataLoader.py:12-17)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, N, C, generator=g) * 2 - 1
    x = x - x.mean(1, keepdim=True)
    return (x / x.norm(dim=-1).max(dim=1)[0].view(B, 1, 1)).contiguous()


def randn(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale
