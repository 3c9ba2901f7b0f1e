"""modules.polar_utils — coordinate conversions with the reference's signatures
(classification/modules/polar_utils.py).  On the hot path these conversions are fused into the
HIP kernels (rs_umbrella_features, rs_group_features); the functions here serve callers that
use the utilities directly and run as ordinary tensor ops on the input's device."""
import math

import torch


def xyz2sphere(xyz, normalize=True):
    """(..., 3) -> (..., 3) = (rho, theta, phi); theta in [0,1], phi in [0,1] when normalised
    (reference :10-31; theta is 0 where rho == 0)."""
    x, y, z = xyz.unbind(-1)
    rho = xyz.pow(2).sum(-1).sqrt()
    theta = torch.where(rho == 0, torch.zeros_like(rho), torch.acos(z / rho))
    phi = torch.atan2(y, x)
    if normalize:
        theta = theta / math.pi
        phi = phi / (2 * math.pi) + .5
    return torch.stack([rho, theta, phi], dim=-1)


def xyz2cylind(xyz, normalize=True):
    """(..., 3) -> (..., 3) = (rho, phi, z) cylindrical (reference :34-54)."""
    x, y, z = xyz.unbind(-1)
    rho = (x * x + y * y).sqrt().clamp(0, 1)
    phi = torch.atan2(y, x)
    z = z.clamp(-1, 1)
    if normalize:
        phi = phi / (2 * math.pi) + .5
        z = (z + 1.) / 2.
    return torch.stack([rho, phi, z], dim=-1)
