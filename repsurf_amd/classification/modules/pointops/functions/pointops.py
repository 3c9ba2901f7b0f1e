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
