"""Coordinate conversions of the RepSurf front-end -- mirror of the reference's modules/polar_utils.py
(xyz2sphere :10-31, xyz2cylind :34-54).  Plain tensor code (any device); the umbrella constructor's
hot path uses the fused kernel behind ops.umbrella_features instead."""
import math

import torch


def xyz2sphere(xyz, normalize=True):
    """[..., 3] -> (rho, theta, phi); theta = 0 where rho = 0; normalised to [0, 1] if asked."""
    rho = xyz.pow(2).sum(-1, keepdim=True).sqrt().clamp(min=0)
    theta = torch.acos(xyz[..., 2:3] / rho)
    phi = torch.atan2(xyz[..., 1:2], xyz[..., 0:1])
    theta = torch.where(rho == 0, torch.zeros_like(theta), theta)
    if normalize:
        theta = theta / math.pi
        phi = phi / (2 * math.pi) + .5
    return torch.cat([rho, theta, phi], dim=-1)


def xyz2cylind(xyz, normalize=True):
    """[..., 3] -> (rho, phi, z) with rho, z clamped to the unit ranges."""
    rho = xyz[..., :2].pow(2).sum(-1, keepdim=True).sqrt().clamp(0, 1)
    phi = torch.atan2(xyz[..., 1:2], xyz[..., 0:1])
    z = xyz[..., 2:3].clamp(-1, 1)
    if normalize:
        phi = phi / (2 * math.pi) + .5
        z = (z + 1.) / 2.
    return torch.cat([rho, phi, z], dim=-1)
