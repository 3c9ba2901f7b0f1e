"""modules.pointops.functions.pointops — the reference's packed-batch operator names
(segmentation/modules/pointops/functions/pointops.py) over librepsurf_hip.

xyz (n,3) rows of all clouds concatenated; offset (b,) int32 running row ends.  Index-producing operators are
not differentiable, like the reference.  The Point-Transformer operators of the reference file (`subtraction`,
`aggregation`, :181-253) belong to a different model family and are not provided.
"""
import torch

from repsurf_amd import ops


def furthestsampling(xyz, offset, new_offset):
    """-> idx (m,) int32 global rows; each cloud starts from its first row (reference :31-49)."""
    return ops.furthestsampling_offset(xyz, offset, new_offset)


def sectorized_fps(xyz, offset, new_offset, num_sectors, min_points=10000):
    """Farthest point sampling inside angular sectors (reference :52-111): clouds of at least `min_points` points are cut
    into `num_sectors` sectors of atan2(x, y), each sector gets new_size // num_sectors picks (the last one also the
    remainder) from an independent FPS -- num_sectors times more workgroups, each with a num_sectors times shorter
    dependency chain.  Runs entirely on the device (repsurf_amd.ops.sectorized_fps: sectorize kernel, FPS over the
    sectors, index remap); the reference's per-cloud host loop and its read-backs are gone."""
    return ops.sectorized_fps(xyz, offset, new_offset, num_sectors, min_points)


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """-> idx (m,nsample) int32 global rows, dist (m,nsample) (sqrt applied, reference :114-130)."""
    if new_xyz is None:
        new_xyz = xyz
    idx, d2 = ops.knnquery_offset(nsample, xyz, new_xyz, offset, new_offset)
    return idx, torch.sqrt(d2)


def grouping(input, idx):
    """input (n,c), idx (m,nsample) -> (m,nsample,c); differentiable w.r.t. input (reference :133-164)."""
    return ops.gather_rows(input.unsqueeze(0), idx.unsqueeze(0)).squeeze(0)


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True):
    """-> (m, nsample, 3+c) = [neighbour - query, feat[neighbour]] (reference :167-188)."""
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        idx, _ = knnquery(nsample, xyz, new_xyz, offset, new_offset)
    grouped_feat = grouping(feat, idx)
    if not use_xyz:
        return grouped_feat
    grouped_xyz = grouping(xyz, idx) - new_xyz.unsqueeze(1)
    return torch.cat((grouped_xyz, grouped_feat), -1)


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """feat (m,c) on xyz (m,3) -> (n,c) on new_xyz (n,3): inverse-distance weighted mean of the 3 nearest
    rows of the same cloud (reference :256-270); differentiable w.r.t. feat."""
    assert k == 3, "the interpolation kernels are built for the three nearest neighbours"
    idx, d2 = ops.knnquery_offset(3, xyz, new_xyz, offset, new_offset)
    weight = ops.interp_weights(d2)
    return ops.three_interpolate(feat.unsqueeze(0), idx.unsqueeze(0), weight.unsqueeze(0)).squeeze(0)


interpolation2 = interpolation
