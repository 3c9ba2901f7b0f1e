"""modules.recons_utils — triangle-reconstruction helpers with the reference's signatures
(classification/modules/recons_utils.py:27-57, 82-90, 108-124, 152-176).

The shipped models never call these directly any more: UmbrellaSurfaceConstructor runs the
fused HIP kernel rs_umbrella_features.  They remain for code that composes
`group_by_umbrella` by hand; they are ordinary tensor ops on the input's device.
"""
import torch


def cal_normal(group_xyz, random_inv=False, is_group=False):
    """Unit normal of each triangle (..., 3 vertices, 3) -> (..., 3).  Sign rule: x component
    positive (per triangle, or per fan from its first triangle when is_group); random_inv flips a
    whole cloud with probability 1/2, drawn from the CPU generator like the reference (:50)."""
    v0 = group_xyz[..., 0, :]
    n = torch.cross(group_xyz[..., 1, :] - v0, group_xyz[..., 2, :] - v0, dim=-1)
    n = n / n.norm(dim=-1, keepdim=True)
    lead = n[..., 0:1, 0:1] if is_group else n[..., 0:1]
    n = n * ((lead > 0).to(n.dtype) * 2 - 1)
    if random_inv:
        flip = (torch.randint(0, 2, (group_xyz.size(0), 1, 1)).float() * 2 - 1).to(n.device)
        n = n * (flip.unsqueeze(-1) if is_group else flip)
    return n


def cal_center(group_xyz):
    """Centroid of the vertices (..., K, 3) -> (..., 3) (reference :82-90)."""
    return group_xyz.mean(dim=-2)


def cal_const(normal, center, is_normalize=True):
    """Plane constant <n, c> (/ sqrt(3) when normalised) -> (..., 1) (reference :108-124)."""
    const = (normal * center).sum(dim=-1, keepdim=True)
    return const / torch.sqrt(torch.tensor([3.0], device=normal.device)) if is_normalize else const


def _patch_nan(first_of, *tensors):
    """Rows whose normal is NaN take the values of the first valid row along dim -2."""
    normal = tensors[0]
    bad = torch.isnan(normal).any(dim=-1)                      # (..., G)
    first = (~bad).int().argmax(dim=-1, keepdim=True)           # 0 when none is valid
    out = []
    for t in tensors:
        donor = torch.gather(t, -2, first.unsqueeze(-1).expand(*first.shape, t.shape[-1]))
        out.append(torch.where(bad.unsqueeze(-1), donor.expand_as(t), t))
    return out


def check_nan(normal, center, pos=None):
    """(B,N,3) variant (reference :127-149): NaN rows take the first valid row of the cloud."""
    ts = [normal, center] + ([pos] if pos is not None else [])
    res = _patch_nan(None, *ts)
    return tuple(res)


def check_nan_umb(normal, center, pos=None):
    """(B,N,G,3) variant (reference :152-176): per point, degenerate fan triangles take the
    values of that point's first valid triangle."""
    ts = [normal, center] + ([pos] if pos is not None else [])
    res = _patch_nan(None, *ts)
    return tuple(res)
