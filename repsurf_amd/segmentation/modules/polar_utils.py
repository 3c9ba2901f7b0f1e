"""modules.polar_utils (segmentation/modules/polar_utils.py): coordinate-system helpers as tensor ops.
The hot path does not call these (the HIP kernels compute the polar channels in registers); they are kept
for API parity and for callers outside the fused kernels."""
import numpy as np
import torch


def xyz2sphere(xyz, normalize=True):
    """(..., 3) -> (rho, theta, phi), theta / pi and phi / 2pi + 0.5 when normalize (reference :10-31)."""
    rho = torch.sqrt(torch.sum(xyz * xyz, dim=-1, keepdim=True)).clamp(min=0)
    theta = torch.acos(xyz[..., 2:3] / rho)
    phi = torch.atan2(xyz[..., 1:2], xyz[..., 0:1])
    theta = torch.where(rho == 0, torch.zeros_like(theta), theta)
    if normalize:
        theta = theta / np.pi
        phi = phi / (2 * np.pi) + .5
    return torch.cat([rho, theta, phi], dim=-1)


def xyz2cylind(xyz, normalize=True):
    """(..., 3) -> (rho, phi, z) (reference :34-55)."""
    rho = torch.sqrt(torch.sum(xyz[..., :2] * xyz[..., :2], dim=-1, keepdim=True)).clamp(0, 1)
    phi = torch.atan2(xyz[..., 1:2], xyz[..., 0:1])
    z = xyz[..., 2:3].clamp(-1, 1)
    if normalize:
        phi = phi / (2 * np.pi) + .5
        z = (z + 1.) / 2.
    return torch.cat([rho, phi, z], dim=-1)
