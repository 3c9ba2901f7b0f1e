"""modules.repsurface_utils — RepSurf-U segmentation building blocks with the reference's names, signatures,
parameter names and list-based calling convention (segmentation/modules/repsurface_utils.py), running on
hand-written HIP kernels.

Data are packed batches: rows of all clouds concatenated, `offset` (B,) int32 running row ends; everything is
channels-last already, so no layout changes happen between modules.  Offsets are read back to the host at
most once per forward (repsurf_amd.ops.host_offsets) instead of 2 `.item()` calls per cloud per stage
(reference :17-22).

RNG: the per-cloud normal inversion is drawn from numpy's global generator with the reference's call
(`np.random.rand(B) < 0.5`, recons_utils.py:29), so `np.random.seed` reproduces the reference's flips.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from repsurf_amd import mlp as _mlp
from repsurf_amd import ops, rng
from modules.pointops.functions import pointops
from modules.polar_utils import xyz2sphere


# The interpolation's backward as a gather over ops.inverse_index (no atomics, bit-reproducible gradients): measured level with the
# atomic scatter inside the step (3.515 against 3.510 ms: every masked-gradient row is read three times) -- an option, off by default
INTERP_GATHER = os.environ.get("REPSURF_INTERP_GATHER", "0") != "0"


class StageGeometry:
    """The coordinate-only part of one sample_and_group call: sampled rows, their coordinates, the running ends of the
    sampled clouds and the kNN lists.  `stage_geometry` computes it; a training loop that has its next batch in hand
    can do that while the previous batch is still training (repsurf_amd.graph.PipelinedStep)."""
    __slots__ = ("fps_idx", "new_center", "new_offset", "group_idx", "csr")

    def __init__(self, fps_idx, new_center, new_offset, group_idx, csr=None):
        """csr (round 4, optional): ops.inverse_index of group_idx -- (csr_off, csr_edges), the grouped rows that read each source
        row: the grouping's backward then gathers (no atomics, no fill)"""
        self.fps_idx, self.new_center, self.new_offset, self.group_idx, self.csr = fps_idx, new_center, new_offset, group_idx, csr


def stage_geometry(stride, nsample, center, offset, num_sector=1, training=True):
    """FPS (reference :17-31) + kNN grouping indices (:33) of one stage: reads coordinates only."""
    if stride > 1:
        new_offset = ops.strided_offset(offset, stride)
        if num_sector > 1 and training:
            fps_idx = pointops.sectorized_fps(center, offset, new_offset, num_sector)
        else:
            fps_idx = pointops.furthestsampling(center, offset, new_offset)
        new_center = ops.gather_rows(center.unsqueeze(0), fps_idx.unsqueeze(0)).squeeze(0)
    else:
        fps_idx, new_center, new_offset = None, center, offset
    group_idx, _ = ops.knnquery_offset(nsample, center, new_center, offset, new_offset)
    csr = ops.inverse_index(group_idx, nsample, new_offset, offset) if (training and center.is_cuda) else None
    return StageGeometry(fps_idx, new_center, new_offset, group_idx, csr)


def sample_and_group(stride, nsample, center, normal, feature, offset, return_polar=False, num_sector=1,
                     training=True, aligned=False, geometry=None):
    """center (N,3), normal (N,Cn), feature (N,C)|None, offset (B,) ->
    new_center (M,3), new_normal (M,Cn), new_feature (M,nsample,3(+3)+Cn+C), new_offset (B,)  (reference :15-51).
    aligned=True (internal use by the SA modules): rows in the padded layout of ops.group_features(aligned=True).
    geometry: a StageGeometry computed ahead of time for these coordinates (optional)."""
    g = geometry if geometry is not None else stage_geometry(stride, nsample, center, offset, num_sector, training)
    new_center, new_offset, group_idx = g.new_center, g.new_offset, g.group_idx
    new_normal = normal if g.fps_idx is None else ops.gather_rows(normal.unsqueeze(0), g.fps_idx.unsqueeze(0)).squeeze(0)
    m = new_center.shape[0]
    rows = ops.group_features(center.unsqueeze(0), new_center.unsqueeze(0), normal.unsqueeze(0),
                              None if feature is None else feature.unsqueeze(0), group_idx.unsqueeze(0),
                              polar=return_polar, aligned=aligned, csr=g.csr)
    return new_center, new_normal, rows.view(m, nsample, -1), new_offset


def resort_points(points, idx):
    """points (N,G,C), idx (N,G) -> points re-ordered along G (reference :54-68)."""
    return torch.gather(points, 1, idx.long().unsqueeze(-1).expand(-1, -1, points.shape[-1]))


def _fixed_rotate(xyz):
    """y-axis 45 deg then z-axis 45 deg (reference :71-74)."""
    rot = xyz.new_tensor([[0.5, -0.5, 0.7071], [0.7071, 0.7071, 0.], [-0.5, 0.5, 0.7071]])
    return xyz @ rot


def _umbrella_ring(xyz, new_xyz, offset, new_offset, k, rotate):
    idx, _ = ops.knnquery_offset(k, xyz, new_xyz, offset, new_offset)
    ring = ops.gather_rows(xyz.unsqueeze(0), idx.unsqueeze(0)).squeeze(0) - new_xyz.unsqueeze(-2)
    key = xyz2sphere(_fixed_rotate(ring) if rotate else ring)[..., 2]
    ring = resort_points(ring, key.argsort(dim=-1)).unsqueeze(-2)
    return torch.cat([torch.zeros_like(ring), ring, torch.roll(ring, -1, dims=-3)], dim=-2)


def group_by_umbrella_v2(xyz, new_xyz, offset, new_offset, k=9):
    """-> (N',k,3,3) fan triangles (origin, p_i, p_{i+1}), neighbours ordered by azimuth after the fixed
    rotation; the query itself stays in the ring (reference :77-98).  The shipped constructor uses the fused
    kernel instead; kNN and gather are HIP here, the k-element ordering runs as tensor ops."""
    return _umbrella_ring(xyz, new_xyz, offset, new_offset, k, True)


def group_by_umbrella(xyz, new_xyz, offset, new_offset, k=9):
    """Same without the rotation (reference :101-122)."""
    return _umbrella_ring(xyz, new_xyz, offset, new_offset, k, False)


def sort_factory(s_type):
    if s_type is None:
        return group_by_umbrella
    elif s_type == 'fix':
        return group_by_umbrella_v2
    raise Exception('No such sorting method')


class SurfaceAbstraction(nn.Module):
    """Set abstraction over surface features, single-branch first layer (reference :135-173)."""

    def __init__(self, stride, nsample, in_channel, mlp, return_polar=True, num_sector=1):
        super().__init__()
        self.stride, self.nsample, self.num_sector = stride, nsample, num_sector
        self.return_polar = return_polar
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for width in mlp:
            self.mlp_convs.append(nn.Conv1d(last, width, 1))
            self.mlp_bns.append(nn.BatchNorm1d(width))
            last = width

    def forward(self, pos_nor_feat_off):
        center, normal, feature, offset = pos_nor_feat_off
        new_center, new_normal, grouped, new_offset = sample_and_group(
            self.stride, self.nsample, center, normal, feature, offset, return_polar=self.return_polar,
            num_sector=self.num_sector, training=self.training)
        m, ns, c = grouped.shape
        pooled = _mlp.sa_mlp_plain(grouped.reshape(m * ns, c), self.mlp_convs, self.mlp_bns, ns)
        return [new_center, new_normal, pooled, new_offset]


class SurfaceAbstractionCD(nn.Module):
    """Set abstraction with the channel-de-differentiated first layer (reference :176-230): position channels
    and feature channels get their own 1x1 conv + BatchNorm, summed before the ReLU; then [conv, BN, ReLU]*,
    max over the nsample neighbours.  Parameter names (mlp_l0, mlp_f0, bn_l0, bn_f0, mlp_convs.i, mlp_bns.i)
    and shapes (Conv1d) match the reference."""

    def __init__(self, stride, nsample, feat_channel, pos_channel, mlp, return_normal=True, return_polar=False,
                 num_sector=1):
        super().__init__()
        self.stride, self.nsample, self.num_sector = stride, nsample, num_sector
        self.return_normal, self.return_polar = return_normal, return_polar
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        self.pos_channel = pos_channel
        self.mlp_l0 = nn.Conv1d(self.pos_channel, mlp[0], 1)
        self.mlp_f0 = nn.Conv1d(feat_channel, mlp[0], 1)
        self.bn_l0 = nn.BatchNorm1d(mlp[0])
        self.bn_f0 = nn.BatchNorm1d(mlp[0])
        last = mlp[0]
        for width in mlp[1:]:
            self.mlp_convs.append(nn.Conv1d(last, width, 1))
            self.mlp_bns.append(nn.BatchNorm1d(width))
            last = width

    def geometry(self, center, offset):
        return stage_geometry(self.stride, self.nsample, center, offset, self.num_sector, self.training)

    def forward(self, pos_nor_feat_off, geometry=None):
        center, normal, feature, offset = pos_nor_feat_off
        new_center, new_normal, grouped, new_offset = sample_and_group(
            self.stride, self.nsample, center, normal, feature, offset, return_polar=self.return_polar,
            num_sector=self.num_sector, training=self.training, aligned=True, geometry=geometry)
        m, ns, c = grouped.shape
        # rows are [offset(3|6), pad, normal, feature, pad]: the feature branch starts on a float4 boundary, so the
        # first-layer GEMM and weight gradient read it with vector loads (77 = 3 + 10 + 64 channels would not)
        foff, fk, _ = ops.aligned_layout(self.return_polar, normal.shape[1], 0 if feature is None else feature.shape[1])
        pooled = _mlp.sa_mlp_cd(grouped.reshape(m * ns, c), self.pos_channel, self.mlp_l0, self.bn_l0, self.mlp_f0,
                                self.bn_f0, self.mlp_convs, self.mlp_bns, ns, feat_off=foff, feat_k=fk)
        return [new_center, new_normal, pooled, new_offset]


def row_mlp(x, linears, bns, relu_last=True, lazy_out=False):
    """[Linear, BatchNorm1d, ReLU]* on ungrouped rows (reference :281-283; the classifier's first block,
    segmentation/models/repsurf/repsurf_umb_ssg.py:38-41).  The chain runs on the fused
    shared-MLP kernels as a stack of groups of ONE row (GEMM + BatchNorm sums, finalize, BN + ReLU pass; one
    weight-gradient and one data-gradient GEMM per layer in backward) -- the framework's route costs a
    batch-norm statistics pass of ~50 us per layer each way at 65 536 rows, and its weight gradient picks a
    32 x 32-tile library GEMM over the 65 536-deep reduction (178 us against ~25 us here)."""
    if len(linears) == 0:
        return x
    return _mlp.sa_mlp_plain(x, linears, bns, 1, relu_last, lazy_out=lazy_out)


class SurfaceFeaturePropagationCD(nn.Module):
    """Feature propagation with the channel-de-differentiated first layer (reference :233-284): coarse features
    go through Linear+BatchNorm, are interpolated onto the fine points with inverse-distance weights over the
    3 nearest coarse points of the same cloud, added to Linear+BatchNorm of the skip features, ReLU, then
    [Linear, BN, ReLU]*.  3-NN search, weights and the gather-interpolation (+ its backward) are HIP kernels;
    the Linear/BatchNorm1d(/ReLU) layers on ungrouped rows run on the fused shared-MLP kernels (`row_mlp`), in training
    and in eval mode (running statistics folded into the operand prologue)."""

    def __init__(self, prev_channel, skip_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        self.skip = skip_channel is not None
        self.mlp_f0 = nn.Linear(prev_channel, mlp[0])
        self.norm_f0 = nn.BatchNorm1d(mlp[0])
        if skip_channel is not None:
            self.mlp_s0 = nn.Linear(skip_channel, mlp[0])
            self.norm_s0 = nn.BatchNorm1d(mlp[0])
        last = mlp[0]
        for width in mlp[1:]:
            self.mlp_convs.append(nn.Linear(last, width))
            self.mlp_bns.append(nn.BatchNorm1d(width))
            last = width

    @staticmethod
    def geometry(xyz1, offset1, xyz2, offset2, training=False):
        """3 nearest coarse rows of every fine row + inverse-distance weights (reference :261-265): coordinates only.
        training: also the inverse of the index (ops.inverse_index: the fine rows that read each coarse row) -- the interpolation's
        backward then gathers; third element of the result (None when it cannot be built)."""
        idx, d2 = ops.knnquery_offset(3, xyz2, xyz1, offset2, offset1)
        csr = ops.inverse_index(idx, 3, offset1, offset2) if (training and xyz1.is_cuda and INTERP_GATHER) else None
        return idx, ops.interp_weights(d2), csr

    def forward(self, pos_feat_off1, pos_feat_off2, geometry=None, lazy_out=False):
        """lazy_out (round 4): return the last layer's raw output + BatchNorm coefficients (mlp_hip.LazyRows) instead of the activated
        rows -- for a consumer that applies them in its first GEMM's operand prologue (the next stage, the classifier); points2 may
        be such an object."""
        xyz1, points1, offset1 = pos_feat_off1      # fine:   (N,3), (N,C)|None, (B,)
        xyz2, points2, offset2 = pos_feat_off2      # coarse: (M,3), (M,C) | LazyRows, (B,)
        geometry = geometry if geometry is not None else self.geometry(xyz1, offset1, xyz2, offset2, self.training and torch.is_grad_enabled())
        idx, weight = geometry[0], geometry[1]
        csr = geometry[2] if len(geometry) > 2 else None
        if self.skip and _mlp.fp_front_usable(self.mlp_f0, self.norm_f0, self.mlp_s0, self.norm_s0):
            # both Linear + BatchNorm pairs, the interpolation, the skip connection and the ReLU as one node: the BatchNorms are
            # applied inside the interpolation launch (round 4)
            new_points = _mlp.fp_front(points2, points1, idx, weight, self.mlp_f0, self.norm_f0, self.mlp_s0, self.norm_s0, csr=csr)
            return row_mlp(new_points, self.mlp_convs, self.mlp_bns, lazy_out=lazy_out)
        points2 = row_mlp(points2, [self.mlp_f0], [self.norm_f0], relu_last=False)
        skip = row_mlp(points1, [self.mlp_s0], [self.norm_s0], relu_last=False).unsqueeze(0) if self.skip else None
        # interpolation + skip connection + ReLU (reference :266-270) in one launch forward, one backward
        new_points = ops.three_interpolate_add_relu(points2.unsqueeze(0), idx.unsqueeze(0), weight.unsqueeze(0), skip, csr=csr).squeeze(0)
        return row_mlp(new_points, self.mlp_convs, self.mlp_bns, lazy_out=lazy_out)


class UmbrellaSurfaceConstructor(nn.Module):
    """Umbrella RepSurf for packed batches (reference :287-329): per point, a fan of k triangles over its k
    nearest neighbours (itself included) -> 10 geometric channels per triangle [polar, normal, const, centroid]
    -> conv-BN-ReLU-conv -> sum over the fan.  `mlps` has the reference's Sequential layout (indices 0,1,3)."""

    def __init__(self, k, in_channel, out_channel, random_inv=True, sort='fix'):
        super().__init__()
        self.k = k
        self.random_inv = random_inv
        self.mlps = nn.Sequential(
            nn.Conv1d(in_channel, out_channel, 1, bias=True),
            nn.BatchNorm1d(out_channel),
            nn.ReLU(True),
            nn.Conv1d(out_channel, out_channel, 1, bias=True),
        )
        self.sort_func = sort_factory(sort)
        self._rotate = sort == 'fix'

    def features(self, center, offset, flip=None):
        """kNN + fan features (N,k,10): everything of the forward that reads coordinates only."""
        if self.random_inv and flip is None:      # numpy global generator, same call as recons_utils.py:29
            flip = rng.draw("npflip", offset.shape[0], 2, center.device)
        idx, _ = ops.knnquery_offset(self.k, center, center, offset, offset)
        return ops.umbrella_fan_offset(center, center, idx, offset, flip, self._rotate)      # (N,k,10)

    def forward(self, center, offset, flip=None, feat=None, moments=None):
        """moments: repsurf_amd.mlp.umbrella_moments of `feat` when both were computed ahead of time (geometry stage)"""
        n = center.shape[0]
        if feat is None:
            feat = self.features(center, offset, flip)
            moments = None
        return _mlp.umbrella_mlp2(feat.reshape(n * self.k, 10), self.mlps, self.k, moments=moments)
