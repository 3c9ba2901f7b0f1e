"""modules.repsurface_utils — RepSurf-U building blocks with the reference's names, signatures,
parameter names and tensor layouts (classification/modules/repsurface_utils.py), running on
hand-written HIP kernels.

Module API layout is the reference's channels-first (B,C,N); internally everything is
channels-last (B,N,C) and the (B,C,N) tensors handed back are permuted *views*, so chaining the
modules costs no transposes.  The `cuda` constructor flag is kept (and stored as `self.cuda`,
like the reference) but both values run the HIP path — see modules.pointnet2_utils.

RNG: like the reference's CPU path, the random normal inversion (recons_utils.py:50) and the
FPS start index (pointnet2_utils.py:66) are drawn from the CPU default generator, in the same
order (constructor, then each SA stage), so a fixed `torch.manual_seed` gives the same picks.
"""
import torch
import torch.nn as nn

from repsurf_amd import mlp as _mlp
from repsurf_amd import ops, rng
from modules.pointnet2_utils import farthest_point_sample, index_points, query_knn_point, query_ball_point
from modules.polar_utils import xyz2sphere


def sample_and_group(npoint, radius, nsample, center, normal, feature, return_normal=True, return_polar=False,
                     cuda=False, geometry=None):
    """center (B,N,3), normal (B,N,Cn), feature (B,N,C)|None ->
    new_center (B,S,3), new_normal (B,S,Cn), new_feature (B,S,nsample,C')   (reference :15-59).
    geometry: optional precomputed (fps_idx, new_center, idx) of this stage (repsurf_amd.geometry)."""
    if geometry is None:
        fps_idx = farthest_point_sample(center, npoint)
        new_center = index_points(center, fps_idx)
        idx = query_ball_point(radius, nsample, center, new_center)
    else:
        fps_idx, new_center, idx = geometry.fps_idx, geometry.new_center, geometry.idx
    new_normal = index_points(normal, fps_idx)
    b, s = fps_idx.shape
    if return_normal:
        rows = ops.group_features(center, new_center, normal, feature, idx, polar=return_polar)
    else:
        empty = normal.new_zeros((b, normal.shape[1], 0))
        rows = ops.group_features(center, new_center, empty, feature, idx, polar=return_polar)
    return new_center, new_normal, rows.view(b, s, nsample, -1)


_zero_centers = {}


def _zero_center(b, device):
    """the (B,1,3) zeros sample_and_group_all returns as new_center / new_normal: a constant, filled once per (B, device)"""
    key = (b, str(device))
    if key not in _zero_centers:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():      # (a capture's allocations belong to its graph)
            return torch.zeros((b, 1, 3), dtype=torch.float32, device=device)
        _zero_centers[key] = torch.zeros((b, 1, 3), dtype=torch.float32, device=device)
    return _zero_centers[key]


def sample_and_group_all(center, normal, feature, return_normal=True, return_polar=False):
    """-> new_center zeros (B,1,3), new_normal = new_center, new_feature (B,1,N,C') (reference :62-88)."""
    b, n, _ = center.shape
    new_center = _zero_center(b, center.device)
    src_normal = normal if return_normal else normal.new_zeros((b, n, 0))
    rows = ops.group_all_features(center, src_normal, feature, polar=return_polar)
    return new_center, new_center, rows.view(b, 1, n, -1)


def resort_points(points, idx):
    """points (B,N,G,C), idx (B,N,G) -> points re-ordered along G (reference :91-109)."""
    return torch.gather(points, 2, idx.long().unsqueeze(-1).expand(-1, -1, -1, points.shape[-1]))


def group_by_umbrella(xyz, new_xyz, k=9, cuda=False):
    """-> (B,N',k-1,3,3): fan triangles (origin, p_i, p_{i+1}) around each query, neighbours in
    counter-clockwise azimuth order (reference :112-132).  The shipped constructor does not call
    this (it uses the fused kernel); kNN and the gather are HIP, the 8-element ordering runs as
    tensor ops."""
    idx = query_knn_point(k, xyz, new_xyz)
    offsets = index_points(xyz, idx)[:, :, 1:] - new_xyz.unsqueeze(-2)
    order = xyz2sphere(offsets)[..., 2].argsort(dim=-1)
    ring = resort_points(offsets, order).unsqueeze(-2)
    return torch.cat([torch.zeros_like(ring), ring, torch.roll(ring, -1, dims=-3)], dim=-2)


class SurfaceAbstraction(nn.Module):
    """Set abstraction over surface features, single-branch first layer (reference :135-183)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, return_polar=True, return_normal=True,
                 cuda=False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.return_normal, self.return_polar = return_normal, return_polar
        self.cuda = cuda
        self.group_all = group_all
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for width in mlp:
            self.mlp_convs.append(nn.Conv2d(last, width, 1))
            self.mlp_bns.append(nn.BatchNorm2d(width))
            last = width

    def forward(self, center, normal, feature):
        center, normal = center.permute(0, 2, 1), normal.permute(0, 2, 1)
        feature = None if feature is None else feature.permute(0, 2, 1)
        if self.group_all:
            new_center, new_normal, grouped = sample_and_group_all(
                center, normal, feature, return_polar=self.return_polar, return_normal=self.return_normal)
        else:
            new_center, new_normal, grouped = sample_and_group(
                self.npoint, self.radius, self.nsample, center, normal, feature,
                return_polar=self.return_polar, return_normal=self.return_normal)
        b, s, ns, c = grouped.shape
        pooled = _mlp.sa_mlp_plain(grouped.reshape(b * s * ns, c), self.mlp_convs, self.mlp_bns, ns)
        return new_center.permute(0, 2, 1), new_normal.permute(0, 2, 1), pooled.view(b, s, -1).permute(0, 2, 1)


class SurfaceAbstractionCD(nn.Module):
    """Set abstraction with the channel-de-differentiated first layer: position channels and
    feature channels get their own 1x1 conv + BatchNorm, summed before the ReLU (reference :186-249).
    Parameter names (mlp_l0, mlp_f0, bn_l0, bn_f0, mlp_convs.i, mlp_bns.i) match the reference."""

    def __init__(self, npoint, radius, nsample, feat_channel, pos_channel, mlp, group_all,
                 return_normal=True, return_polar=False, cuda=False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.return_normal, self.return_polar = return_normal, return_polar
        self.cuda = cuda
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        self.pos_channel = pos_channel
        self.group_all = group_all

        self.mlp_l0 = nn.Conv2d(self.pos_channel, mlp[0], 1)
        self.mlp_f0 = nn.Conv2d(feat_channel, mlp[0], 1)
        self.bn_l0 = nn.BatchNorm2d(mlp[0])
        self.bn_f0 = nn.BatchNorm2d(mlp[0])
        last = mlp[0]
        for width in mlp[1:]:
            self.mlp_convs.append(nn.Conv2d(last, width, 1))
            self.mlp_bns.append(nn.BatchNorm2d(width))
            last = width

    def forward(self, center, normal, feature, geometry=None):
        center, normal = center.permute(0, 2, 1), normal.permute(0, 2, 1)
        feature = None if feature is None else feature.permute(0, 2, 1)
        if (not self.group_all and self.return_normal and _mlp.COMPACT_GROUPS
                and self.bn_l0.training):
            return self._forward_compact(center, normal, feature, geometry)
        if self.group_all:
            new_center, new_normal, grouped = sample_and_group_all(
                center, normal, feature, return_normal=self.return_normal, return_polar=self.return_polar)
        else:
            new_center, new_normal, grouped = sample_and_group(
                self.npoint, self.radius, self.nsample, center, normal, feature,
                return_normal=self.return_normal, return_polar=self.return_polar, geometry=geometry)
        b, s, ns, c = grouped.shape
        pooled = _mlp.sa_mlp_cd(grouped.reshape(b * s * ns, c), self.pos_channel, self.mlp_l0, self.bn_l0,
                                self.mlp_f0, self.bn_f0, self.mlp_convs, self.mlp_bns, ns)
        return new_center.permute(0, 2, 1), new_normal.permute(0, 2, 1), pooled.view(b, s, -1).permute(0, 2, 1)


def _sa_forward_compact(self, center, normal, feature, geometry):
    """sample_and_group + shared MLP on the distinct ball-query slots only (ops.CompactGroups)."""
    if geometry is None:
        fps_idx = farthest_point_sample(center, self.npoint)
        new_center = index_points(center, fps_idx)
        idx, cnt = ops.ballquery(self.radius, self.nsample, center, new_center, return_count=True)
    else:
        fps_idx, new_center, idx, cnt = geometry.fps_idx, geometry.new_center, geometry.idx, geometry.cnt
        if getattr(geometry, "center", None) is not None:
            center = geometry.center           # the geometry's channels-last copy: no transpose of the (B, 3, N) view here
    # the centres' own normal rows (index_points(normal, fps_idx), reference :31) come out of the grouping launches
    groups, new_normal = ops.group_features_compact(center, new_center, normal, feature, idx, cnt, polar=self.return_polar,
                                                    index=getattr(geometry, "index", None), fps_idx=fps_idx)
    pooled = _mlp.sa_mlp_cd(groups.x, self.pos_channel, self.mlp_l0, self.bn_l0, self.mlp_f0, self.bn_f0,
                            self.mlp_convs, self.mlp_bns, self.nsample, compact=groups)
    b, s = fps_idx.shape
    return new_center.permute(0, 2, 1), new_normal.permute(0, 2, 1), pooled.view(b, s, -1).permute(0, 2, 1)


SurfaceAbstractionCD._forward_compact = _sa_forward_compact


class UmbrellaSurfaceConstructor(nn.Module):
    """Umbrella RepSurf: per point, a fan of k-1 triangles over its kNN ring -> 10 geometric
    channels per triangle -> shared MLP -> pooled over the fan (reference :252-307).
    `mlps` has the reference's Sequential layout (indices 0,1,3,4,6 carry parameters)."""

    def __init__(self, k, in_channel, aggr_type='sum', return_dist=False, random_inv=True, cuda=False):
        super().__init__()
        self.k = k
        self.return_dist = return_dist
        self.random_inv = random_inv
        self.aggr_type = aggr_type
        self.cuda = cuda
        self.mlps = nn.Sequential(
            nn.Conv2d(in_channel, in_channel, 1, bias=False),
            nn.BatchNorm2d(in_channel),
            nn.ReLU(True),
            nn.Conv2d(in_channel, in_channel, 1, bias=True),
            nn.BatchNorm2d(in_channel),
            nn.ReLU(True),
            nn.Conv2d(in_channel, in_channel, 1, bias=True),
        )

    def features(self, center, flip=None):
        """The weight-free part: kNN ring, fan triangles, 10 geometric channels -> (B,N,k-1,10) (:276-293)."""
        xyz = center.permute(0, 2, 1).contiguous()
        b = xyz.shape[0]
        if self.random_inv and flip is None:   # per-cloud sign, CPU generator, same call as recons_utils.py:50
            flip = rng.draw("flip", b, 2, xyz.device)
        return ops.umbrella_features(xyz, self.k, flip)           # [centre, polar, normal, pos]

    def moments(self, feat):
        """First / second moments of the fan features (geometry-only): what BatchNorm 0 of `mlps` is computed from on the
        matrix-pipe path (csrc/umbrella_mfma.hip).  None where that path does not apply (9-channel features)."""
        if not self.return_dist or not feat.is_cuda:
            return None
        return _mlp.umbrella_moments(feat.reshape(-1, feat.shape[-1]))

    def forward(self, center, flip=None, feat=None, moments=None):
        """feat: the output of `features` for these points when it was computed ahead of time (moments: of `moments`)."""
        if feat is None:
            feat = self.features(center, flip)
            moments = None
        b, n = feat.shape[0], feat.shape[1]
        if not self.return_dist:
            feat = feat[..., :9]
            moments = None
        g = self.k - 1
        pooled = _mlp.umbrella_mlp(feat.reshape(b * n * g, feat.shape[-1]), self.mlps, g, self.aggr_type, moments=moments)
        return pooled.view(b, n, -1).permute(0, 2, 1)
