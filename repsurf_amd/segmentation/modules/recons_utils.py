"""modules.recons_utils (segmentation/modules/recons_utils.py): triangle reconstruction helpers as tensor ops.
The shipped constructor runs the fused HIP kernel (repsurf_amd.ops.umbrella_fan_offset) instead; these are the
same functions for callers that assemble the pieces themselves.  `random_flips` is the one piece the hot path
uses: the per-cloud normal inversion drawn from the numpy global generator exactly like the reference."""
import numpy as np
import torch


def random_flips(num_clouds):
    """+-1 per cloud: `np.random.rand(B) < 0.5` keeps the normal, otherwise inverts it (reference :29-35)."""
    keep = np.random.rand(num_clouds) < 0.5
    return np.where(keep, 1.0, -1.0).astype(np.float32)


def _per_point_sign(offset, signs, device):
    ends = [int(v) for v in offset.tolist()]
    lens = np.diff(np.concatenate([[0], ends]))
    return torch.from_numpy(np.repeat(signs, lens)).to(device).unsqueeze(-1)


def cal_normal(group_xyz, offset, random_inv=False, is_group=False):
    """Unit normal of each triangle, first component of the (first) triangle positive, optional per-cloud
    random inversion (reference :10-45).  group_xyz (N,3,3) / (N,G,3,3)."""
    e1 = group_xyz[..., 1, :] - group_xyz[..., 0, :]
    e2 = group_xyz[..., 2, :] - group_xyz[..., 0, :]
    nor = torch.cross(e1, e2, dim=-1)
    unit = nor / torch.norm(nor, dim=-1, keepdim=True)
    if not is_group:
        pos_mask = (unit[..., 0] > 0).float() * 2. - 1.
    else:
        pos_mask = (unit[..., 0:1, 0] > 0).float() * 2. - 1.
    unit = unit * pos_mask.unsqueeze(-1)
    if random_inv:
        mask = _per_point_sign(offset, random_flips(offset.shape[0]), unit.device)
        unit = unit * (mask if not is_group else mask.unsqueeze(-1))
    return unit


def cal_center(group_xyz):
    """Centroid of each triangle (reference :48-57)."""
    return torch.mean(group_xyz, dim=-2)


def cal_const(normal, center, is_normalize=True):
    """<normal, center> (/ sqrt(3)) (reference :84-100)."""
    const = torch.sum(normal * center, dim=-1, keepdim=True)
    return const / np.float32(np.sqrt(np.float32(3))) if is_normalize else const


def check_nan(normal, center, pos=None):
    """Rows with a NaN normal take the first valid row's values (reference :103-125)."""
    mask = torch.isnan(normal).any(dim=-1)
    first = torch.argmax((~mask).int(), dim=-1)
    normal = torch.where(mask.unsqueeze(-1), normal[first].unsqueeze(0), normal)
    center = torch.where(mask.unsqueeze(-1), center[first].unsqueeze(0), center)
    if pos is not None:
        pos = torch.where(mask.unsqueeze(-1), pos[first].unsqueeze(0), pos)
        return normal, center, pos
    return normal, center


def check_nan_umb(normal, center, pos=None):
    """Per point: fan triangles with a NaN normal take the first valid triangle's normal / centroid / const
    (reference :128-151).  normal, center (N,G,3); pos (N,G,1)."""
    n = normal.shape[0]
    mask = torch.isnan(normal).any(dim=-1)
    first = torch.argmax((~mask).int(), dim=-1)
    rows = torch.arange(n, device=normal.device)
    m = mask.unsqueeze(-1)
    normal = torch.where(m, normal[rows, first].unsqueeze(1), normal)
    center = torch.where(m, center[rows, first].unsqueeze(1), center)
    if pos is not None:
        pos = torch.where(m, pos[rows, first].unsqueeze(1), pos)
        return normal, center, pos
    return normal, center
