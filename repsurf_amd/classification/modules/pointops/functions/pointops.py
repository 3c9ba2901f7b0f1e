"""modules.pointops.functions.pointops — the reference's L1 operator names
(classification/modules/pointops/functions/pointops.py:54-354) over librepsurf_hip.

Tensor layouts follow the reference operators: features are channels-first (b, c, n) here,
and each wrapper converts to the library's channels-last layout.  Code that wants to avoid the
transposes should call repsurf_amd.ops directly (as modules.pointnet2_utils does).
Index-producing operators are not differentiable, like the reference (`backward` returns None).
"""
import torch

from repsurf_amd import ops


def furthestsampling(xyz, m):
    """xyz (b,n,3) -> idx (b,m) int32, first pick = index 0 (the CUDA operator's rule,
    sampling_cuda_kernel.cu:72-74); arithmetic and tie rule as farthest_point_sample(cuda=False)."""
    return ops.furthestsampling(xyz, m, None)


def gathering(features, idx):
    """features (b,c,n), idx (b,m) -> (b,c,m)"""
    return ops.gather_rows(features.transpose(1, 2), idx).transpose(1, 2)


def grouping(features, idx):
    """features (b,c,n), idx (b,m,nsample) -> (b,c,m,nsample)"""
    return ops.gather_rows(features.transpose(1, 2), idx).permute(0, 3, 1, 2)


def ballquery(radius, nsample, xyz, new_xyz):
    """-> (b,m,nsample) int32"""
    return ops.ballquery(radius, nsample, xyz, new_xyz)


def knnquery(nsample, xyz, new_xyz=None):
    """-> (b,m,nsample) int32"""
    return ops.knnquery(nsample, xyz, new_xyz)


knnquery_heap = knnquery


def nearestneighbor(unknown, known):
    """unknown (b,n,3), known (b,m,3) -> dist (b,n,3) (sqrt applied, like the reference :102), idx (b,n,3)"""
    d2, idx = ops.three_nn(unknown, known)
    return torch.sqrt(d2), idx


def interpolation(features, idx, weight):
    """features (b,c,m), idx/weight (b,n,3) -> (b,c,n)"""
    return ops.three_interpolate(features.transpose(1, 2), idx, weight).transpose(1, 2)


def grouping_int(features, idx):
    """features (b,c,n) integer payload, idx (b,m,nsample) -> (b,c,m,nsample) int64 (reference :183-203): the gather kernel on
    4-byte words (an int64 channel travels as two of them); not differentiable."""
    words = features.to(torch.int64).permute(0, 2, 1).contiguous().view(torch.float32)          # (b, n, 2c) words
    return ops.gather_rows(words, idx).contiguous().view(torch.int64).permute(0, 3, 1, 2)      # (b, m, nsample, 2c) words -> (b, c, m, nsample)


def knnquery_naive(nsample, xyz, new_xyz=None):
    """The reference's sort-based kNN (:252-291: direct-difference distances, full sort, first nsample): the same neighbour
    sets as `knnquery` -- this package has one kNN kernel (lowest index first among equal distances, where the reference's
    unstable 1024-wide sort is arbitrary)."""
    return knnquery(nsample, xyz, new_xyz)


class QueryAndGroup(torch.nn.Module):
    """Ball query (radius) or kNN (radius=None) + grouping (reference :357-410): xyz (b,n,3), new_xyz (b,m,3), features (b,c,n)
    -> new_features (b, 3+c | c | 3, m, nsample), grouped_xyz (b,3,m,nsample)[, idx (b,m,nsample) int64]."""

    def __init__(self, radius=None, nsample=32, use_xyz=True, return_idx=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.return_idx = return_idx

    def forward(self, xyz, new_xyz=None, features=None, idx=None):
        if new_xyz is None:
            new_xyz = xyz
        if idx is None:
            idx = ballquery(self.radius, self.nsample, xyz, new_xyz) if self.radius is not None else knnquery_heap(self.nsample, xyz, new_xyz)
        grouped_xyz = grouping(xyz.transpose(1, 2), idx)                         # (b, 3, m, nsample)
        diff = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped = grouping(features, idx)
            new_features = torch.cat([diff, grouped], dim=1) if self.use_xyz else grouped
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = diff
        return (new_features, grouped_xyz, idx.long()) if self.return_idx else (new_features, grouped_xyz)
