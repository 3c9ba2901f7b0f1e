"""Shared constructor for the RepSurf-U ScanObjectNN classifiers.

Both shipped variants are: UmbrellaSurfaceConstructor -> a ladder of SurfaceAbstractionCD stages
(the last one `group_all`) -> a 3-layer MLP head with log-softmax.  The attribute names
(`surface_constructor`, `sa1..saN`, `classfier` [sic]) and the Sequential indices of the head
are the reference's, so its checkpoints load
(classification/models/repsurf/repsurf_ssg_umb.py:11-57, repsurf_ssg_umb_2x.py:11-61).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from modules.repsurface_utils import SurfaceAbstractionCD, UmbrellaSurfaceConstructor
from repsurf_amd import head as _head
from repsurf_amd import mlp as _mlp
from repsurf_amd import rng
from repsurf_amd.geometry import GeometryPlan

REPSURF_CHANNEL = 10


class GeoState:
    """What a forward needs that depends on the coordinates only -- the constructor's fan features and, per sampling
    stage, FPS picks, their coordinates, ball-query indices and distinct-neighbour counts.  `UmbrellaClassifier.geometry`
    computes it; a training loop that knows its next batch can do that while the previous batch is still in backward
    (repsurf_amd.graph.PipelinedStep)."""
    __slots__ = ("feat", "stages", "xyz", "moments")

    def __init__(self, feat, stages, xyz=None, moments=None):
        """xyz: the channels-last (B, N, 3) copy of the coordinates the geometry worked on (the stages group from it);
        moments: (11, 16) fp64 first / second moments of the fan features (repsurf_amd.mlp.umbrella_moments): BatchNorm 0 of the
        constructor MLP follows from them, and they are geometry"""
        self.feat, self.stages, self.xyz, self.moments = feat, stages, xyz, moments
        for i, g in enumerate(stages):       # every stage samples from the previous stage's centres
            g.center = (xyz if i == 0 else stages[i - 1].new_center)

    def tensors(self):
        out = [self.feat] + ([self.xyz] if self.xyz is not None else []) + ([self.moments] if self.moments is not None else [])
        for g in self.stages:
            out += g.tensors()
        return out

    def clone(self):
        return GeoState(self.feat.clone(), [g.clone() for g in self.stages], None if self.xyz is None else self.xyz.clone(),
                        None if self.moments is None else self.moments.clone())

    def copy_(self, other):
        dst, src = self.tensors(), other.tensors()
        if len(dst) != len(src) or any(d is None or s_ is None or d.dtype not in (torch.float32, torch.int32, torch.float64) or d.dtype != s_.dtype
                                       for d, s_ in zip(dst, src)):
            # (an exception, not an assert: `python -O` must not turn a tensor of another dtype into a silently skipped copy)
            raise RuntimeError("GeoState.copy_: the two states do not hold the same float32 / int32 / float64 tensors")
        for dt in (torch.float32, torch.int32, torch.float64):          # one multi-tensor launch per dtype instead of one copy per tensor
            pairs = [(d, s_) for d, s_ in zip(dst, src) if d.dtype == dt]
            if pairs:
                torch._foreach_copy_([p[0] for p in pairs], [p[1] for p in pairs])
        return self


class UmbrellaClassifier(nn.Module):
    def __init__(self, args, stages, head_in):
        """stages: list of dicts(npoint, radius, nsample, mlp); feature widths chain automatically."""
        super().__init__()
        pos_channel = (6 if args.return_polar else 3) if args.return_center else 0
        self.init_nsample = args.num_point
        self.return_dist = args.return_dist
        self.surface_constructor = UmbrellaSurfaceConstructor(
            args.group_size + 1, REPSURF_CHANNEL, return_dist=args.return_dist, aggr_type=args.umb_pool,
            cuda=args.cuda_ops)
        width = 0
        self._stage_names = []
        self._sampling = [(st["npoint"], st["radius"], st["nsample"]) for st in stages[:-1]]
        self.overlap_geometry = True      # FPS / ball query of every stage on a side stream (repsurf_amd.geometry)
        for i, st in enumerate(stages, 1):
            last = i == len(stages)
            sa = SurfaceAbstractionCD(
                npoint=None if last else st["npoint"], radius=None if last else st["radius"],
                nsample=None if last else st["nsample"], feat_channel=width + REPSURF_CHANNEL,
                pos_channel=pos_channel, mlp=st["mlp"], group_all=last,
                return_polar=args.return_polar, cuda=args.cuda_ops)
            setattr(self, f"sa{i}", sa)
            self._stage_names.append(f"sa{i}")
            width = st["mlp"][-1]
        self.head_in = head_in
        self.classfier = nn.Sequential(
            nn.Linear(head_in, 512), nn.BatchNorm1d(512), nn.ReLU(True), nn.Dropout(0.4),
            nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(True), nn.Dropout(0.4),
            nn.Linear(256, args.num_class))

    def forward(self, points, geo=None):
        """points (B, C, N) -> log-probabilities; geo: `self.geometry(points)` computed ahead of time (optional)."""
        with _mlp.deferred_counters():
            return self._forward(points, geo)

    def geometry(self, points, fork=True):
        """The coordinate-only work of a forward (no learned parameter is read): constructor kNN + fan features on the
        current stream, FPS / ball query of every stage on the side stream.  Draws the reference's CPU-generator numbers
        in the reference's order (constructor flip, then one FPS start per stage)."""
        center = points[:, :3, :]
        sc = self.surface_constructor
        flip = rng.draw("flip", center.shape[0], 2, center.device) if sc.random_inv else None
        xyz = center.permute(0, 2, 1).contiguous()
        if fork:        # in-step form: the FPS chain goes to the side stream first, the kNN hides it
            plan = GeometryPlan(xyz, self._sampling, fork=True, compact=self._compact())
            feat = sc.features(center, flip)
        else:           # one serial branch next to another batch's network: full-chip kNN first (under the light
            feat = sc.features(center, flip)    # head of that forward), the 32-workgroup FPS chains afterwards (1.98 vs 2.00 ms)
            plan = GeometryPlan(xyz, self._sampling, fork=False, compact=self._compact())
        moments = sc.moments(feat) if (self.training and _mlp.umbrella_moments_wanted(3)) else None
        return GeoState(feat, [plan.stage(i) for i in range(len(self._sampling))], xyz, moments)

    def early_gradient_modules(self):
        """The modules whose parameter gradients are complete first in backward (the last SA stage, then the head): bucket 0
        of the sharded step's gradient all-reduce (repsurf_amd.graph.PipelinedStep, REPSURF_GRAD_BUCKETS=2).  The first entry
        is the one whose INPUT gradient marks that moment."""
        return [getattr(self, self._stage_names[-1]), self.classfier]

    def _compact(self):
        """the SA stages take the compacted-groups path (training mode): their bookkeeping belongs to the geometry"""
        return bool(_mlp.COMPACT_GROUPS and self.training)

    def _sa_convs(self):
        out = []
        for name in self._stage_names:
            sa = getattr(self, name)
            out += [sa.mlp_l0, sa.mlp_f0] + list(sa.mlp_convs)
        return out

    def _forward(self, points, geo=None):
        center = points[:, :3, :]
        plan = None
        if points.is_cuda and self.training:
            _mlp.prepack(self._sa_convs())      # padded / transposed weight copies of all stages: one launch
        if geo is not None:
            normal = self.surface_constructor(center, feat=geo.feat, moments=geo.moments)
        elif self.overlap_geometry:
            # same CPU-generator order as the reference: the constructor's flip first, then one FPS start per stage
            sc = self.surface_constructor
            flip = rng.draw("flip", center.shape[0], 2, center.device) if sc.random_inv else None
            plan = GeometryPlan(center.permute(0, 2, 1).contiguous(), self._sampling, compact=self._compact())
            normal = sc(center, flip=flip)
        else:
            normal = self.surface_constructor(center)
        feature = None
        for i, name in enumerate(self._stage_names):
            if i >= len(self._sampling):
                sg = None
            elif geo is not None:
                sg = geo.stages[i]
            else:
                sg = plan.stage(i) if plan is not None else None
            center, normal, feature = getattr(self, name)(center, normal, feature, geometry=sg)
        x = feature.reshape(-1, self.head_in)
        if _head.usable(self.classfier, x):            # training batches of <= 64 clouds: 3 fused launches
            return _head.classifier_logprobs(self.classfier, x)
        if _head.rows_usable(self.classfier, x):       # eval mode / more rows / SyncBatchNorm: row stacks of the shared-MLP kernels
            return _head.classifier_logprobs_rows(self.classfier, x)
        return F.log_softmax(self.classfier(x), -1)    # (a head that is not the reference's Sequential)
