"""modules.pointnet2_utils — the PointNet++ building blocks of the segmentation sub-project
(segmentation/modules/pointnet2_utils.py:13-135) with the reference's names, signatures, parameter names and list-based
calling convention, over the same HIP kernels as modules.repsurface_utils: `models/pointnet2/pointnet2_ssg.py` of the
reference (its PointNet++ baseline over the same `pointops` boundary) imports and runs over this file unmodified
(tests/test_dropin.py).

Packed batches: rows of all clouds concatenated, `offset` (B,) int32 running row ends, channels last."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from repsurf_amd import mlp as _mlp
from repsurf_amd import ops
from modules.pointops.functions import pointops


def sample_and_group(stride, nsample, xyz, points, offset, return_idx=False, num_sector=1):
    """xyz (N,3), points (N,C)|None, offset (B,) -> new_xyz (M,3), new_points (M,nsample,3(+C)), new_offset (B,)
    [, group_idx (M,nsample)]  (reference :13-46).  FPS / sectorized FPS / kNN are the packed-batch HIP kernels; the
    grouped rows [neighbour - centre, points[neighbour]] leave one gather launch (differentiable w.r.t. points)."""
    if stride > 1:
        new_offset = ops.strided_offset(offset, stride)            # reference :17-22 (two .item() per cloud there)
        if num_sector > 1:
            fps_idx = pointops.sectorized_fps(xyz, offset, new_offset, num_sector)
        else:
            fps_idx = pointops.furthestsampling(xyz, offset, new_offset)
        new_xyz = ops.gather_rows(xyz.unsqueeze(0), fps_idx.unsqueeze(0)).squeeze(0)
    else:
        new_xyz, new_offset = xyz, offset
    m = new_xyz.shape[0]
    group_idx, _ = ops.knnquery_offset(nsample, xyz, new_xyz, offset, new_offset)
    if points is not None and not return_idx:
        # rs_group_features with the point features in the place of the normal channels: rows = [offset(3), points[idx]]
        rows = ops.group_features(xyz.unsqueeze(0), new_xyz.unsqueeze(0), points.unsqueeze(0), None,
                                  group_idx.unsqueeze(0), polar=False)
        new_points = rows.view(m, nsample, -1)
    else:
        new_points = ops.gather_rows(xyz.unsqueeze(0), group_idx.reshape(1, -1)).view(m, nsample, 3) - new_xyz.unsqueeze(1)
    if return_idx:
        return new_xyz, new_points, new_offset, group_idx
    return new_xyz, new_points, new_offset


class PointNetSetAbstraction(nn.Module):
    """PointNet++ set abstraction (reference :49-83): sample_and_group -> [Conv1d(1x1), BatchNorm1d, ReLU]* -> max over
    the nsample neighbours, on the fused shared-MLP kernels."""

    def __init__(self, stride, nsample, in_channel, mlp, num_sector=1):
        super().__init__()
        self.stride, self.nsample, self.num_sector = stride, nsample, num_sector
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for width in mlp:
            self.mlp_convs.append(nn.Conv1d(last, width, 1))
            self.mlp_bns.append(nn.BatchNorm1d(width))
            last = width

    def forward(self, pos_feat_off):
        xyz, points, offset = pos_feat_off                         # (N,3), (N,C), (B,)
        new_xyz, grouped, new_offset = sample_and_group(self.stride, self.nsample, xyz, points, offset,
                                                        num_sector=self.num_sector)
        m, ns, c = grouped.shape
        pooled = _mlp.sa_mlp_plain(grouped.reshape(m * ns, c), self.mlp_convs, self.mlp_bns, ns)
        return [new_xyz, pooled, new_offset]


class PointNetFeaturePropagation(nn.Module):
    """PointNet++ feature propagation (reference :86-135): inverse-distance interpolation of the coarse features over the
    3 nearest coarse rows of the same cloud, concatenated behind the skip features, then [Linear, BatchNorm1d, ReLU]*."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for width in mlp:
            self.mlp_convs.append(nn.Linear(last, width))
            self.mlp_bns.append(nn.BatchNorm1d(width))
            last = width

    def forward(self, pos_feat_off1, pos_feat_off2):
        xyz1, points1, offset1 = pos_feat_off1                     # fine:   (N,3), (N,C1)|None, (B,)
        xyz2, points2, offset2 = pos_feat_off2                     # coarse: (M,3), (M,C2), (B,)
        idx, d2 = ops.knnquery_offset(3, xyz2, xyz1, offset2, offset1)
        weight = ops.interp_weights(d2)                            # 1 / (dist + 1e-8), normalised (reference :110-112)
        interpolated = ops.three_interpolate(points2.unsqueeze(0), idx.unsqueeze(0), weight.unsqueeze(0)).squeeze(0)
        new_points = interpolated if points1 is None else torch.cat([points1, interpolated], dim=1)
        return _mlp.sa_mlp_plain(new_points, self.mlp_convs, self.mlp_bns, 1, True)
