"""models.repsurf.repsurf_umb_ssg — RepSurf-U (umbrella, single-scale grouping) segmentation network
(segmentation/models/repsurf/repsurf_umb_ssg.py): same constructor arguments, forward signature and
state-dict keys as the reference, built on the HIP-backed modules."""
import torch
import torch.nn as nn

from repsurf_amd import mlp as _mlp

from modules.repsurface_utils import UmbrellaSurfaceConstructor, SurfaceAbstractionCD, SurfaceFeaturePropagationCD, row_mlp


class Model(nn.Module):
    def __init__(self, args):
        super().__init__()
        center_channel = 6 if args.return_polar else 3
        repsurf_in_channel = 10
        repsurf_out_channel = 10

        self.sa1 = SurfaceAbstractionCD(4, 32, args.in_channel + repsurf_out_channel, center_channel, [32, 32, 64],
                                        True, args.return_polar, num_sector=4)
        self.sa2 = SurfaceAbstractionCD(4, 32, 64 + repsurf_out_channel, center_channel, [64, 64, 128],
                                        True, args.return_polar)
        self.sa3 = SurfaceAbstractionCD(4, 32, 128 + repsurf_out_channel, center_channel, [128, 128, 256],
                                        True, args.return_polar)
        self.sa4 = SurfaceAbstractionCD(4, 32, 256 + repsurf_out_channel, center_channel, [256, 256, 512],
                                        True, args.return_polar)

        self.fp4 = SurfaceFeaturePropagationCD(512, 256, [256, 256])
        self.fp3 = SurfaceFeaturePropagationCD(256, 128, [256, 256])
        self.fp2 = SurfaceFeaturePropagationCD(256, 64, [256, 128])
        self.fp1 = SurfaceFeaturePropagationCD(128, None, [128, 128, 128])

        self.classifier = nn.Sequential(
            nn.Linear(128, 128),
            nn.BatchNorm1d(128),
            nn.ReLU(True),
            nn.Dropout(0.5),
            nn.Linear(128, args.num_class),
        )
        self.surface_constructor = UmbrellaSurfaceConstructor(args.group_size + 1, repsurf_in_channel,
                                                              repsurf_out_channel)

    def _packed_layers(self):
        """every Conv1d / Linear the fused stacks of this step will ask a padded or transposed weight copy of"""
        out = []
        for sa in (self.sa1, self.sa2, self.sa3, self.sa4):
            out += [sa.mlp_l0, sa.mlp_f0] + list(sa.mlp_convs)
        for fp in (self.fp4, self.fp3, self.fp2, self.fp1):
            out += [fp.mlp_f0] + ([fp.mlp_s0] if fp.skip else []) + list(fp.mlp_convs)
        return out + [self.classifier[0], self.classifier[4]]

    def geometry(self, pos_feat_off0, fork=False):
        """Everything of a forward that reads coordinates only (no learned parameter): the constructor's kNN + fan
        features, FPS + kNN grouping of the four SA stages, the 3-NN + interpolation weights of the four FP stages.
        A training loop that has its next batch in hand runs it under the previous batch's network
        (repsurf_amd.graph.PipelinedStep); the numpy-generator flips are drawn in the reference's order."""
        coord, offset = pos_feat_off0[0], pos_feat_off0[2]
        sc = self.surface_constructor
        feat = sc.features(coord, offset)
        stages, centers, offsets = [], [coord], [offset]
        for sa in (self.sa1, self.sa2, self.sa3, self.sa4):
            g = sa.geometry(centers[-1], offsets[-1])
            stages.append(g)
            centers.append(g.new_center)
            offsets.append(g.new_offset)
        fps = []
        for fine, coarse in ((3, 4), (2, 3), (1, 2), (0, 1)):          # fp4, fp3, fp2, fp1
            fps.append(SurfaceFeaturePropagationCD.geometry(centers[fine], offsets[fine], centers[coarse], offsets[coarse], self.training))
        moments = _mlp.umbrella_moments(feat.reshape(-1, 10)) if (self.training and feat.is_cuda and _mlp.umbrella_moments_wanted(2)) else None
        return SegGeoState(feat, stages, fps, moments)

    def forward(self, pos_feat_off0, geo=None):
        with _mlp.deferred_counters():                 # num_batches_tracked += 1 of all 30 BatchNorms: one launch, not one per stack
            return self._forward(pos_feat_off0, geo)

    def _forward(self, pos_feat_off0, geo=None):
        coord, feat, offset = pos_feat_off0            # (N,3), (N,C_in-3), (B,) running row ends
        if coord.is_cuda and self.training and torch.is_grad_enabled():
            _mlp.prepack(self._packed_layers())        # ~23 weight-pack launches of the step in one
        sg = geo.stages if geo is not None else [None] * 4
        fg = geo.fps if geo is not None else [None] * 4
        normal = self.surface_constructor(coord, offset, feat=None if geo is None else geo.feat,
                                          moments=None if geo is None else geo.moments)
        level0 = [coord, normal, torch.cat([coord, feat], 1), offset]
        level1 = self.sa1(level0, geometry=sg[0])
        level2 = self.sa2(level1, geometry=sg[1])
        level3 = self.sa3(level2, geometry=sg[2])
        level4 = self.sa4(level3, geometry=sg[3])

        def pfo(level):                                # [center, normal, feature, offset] -> [center, feature, offset]
            return [level[0], level[2], level[3]]
        # The decoder is a chain of row stacks, each the only consumer of the one before: the last BatchNorm + ReLU of a stage is
        # applied in the operand prologue of the next stage's first GEMM (mlp_hip.LazyRows) instead of a pass over (rows, C) each way
        fps = (self.fp4, self.fp3, self.fp2, self.fp1)
        lazy = (coord.is_cuda and self.training and torch.is_grad_enabled() and all(fp.skip for fp in fps[:3])
                and all(_mlp.fp_front_usable(fp.mlp_f0, fp.norm_f0, fp.mlp_s0, fp.norm_s0) for fp in fps[:3])
                and _mlp.lazy_rows_usable([m_ for fp in fps for m_ in [fp.norm_f0] + list(fp.mlp_bns)] + [self.classifier[1]]))
        f3 = self.fp4(pfo(level3), pfo(level4), geometry=fg[0], lazy_out=lazy)
        f2 = self.fp3(pfo(level2), [level3[0], f3, level3[3]], geometry=fg[1], lazy_out=lazy)
        f1 = self.fp2(pfo(level1), [level2[0], f2, level2[3]], geometry=fg[2], lazy_out=lazy)
        f0 = self.fp1([coord, None, offset], [level1[0], f1, level1[3]], geometry=fg[3], lazy_out=lazy)
        cls = self.classifier                          # Linear-BN-ReLU on the fused kernels, Dropout, then the output Linear on the row GEMM
        return _mlp.row_linear(cls[3](row_mlp(f0, [cls[0]], [cls[1]])), cls[4])


class SegGeoState:
    """Tensors `Model.geometry` produced (int32 indices, coordinates, weights): what a forward needs besides the
    features and the parameters.  Offsets of the sampled levels are host-known constants of the batch shape."""
    __slots__ = ("feat", "stages", "fps", "moments")

    def __init__(self, feat, stages, fps, moments=None):
        """moments: (11, 16) fp64 moments of the fan features (repsurf_amd.mlp.umbrella_moments), geometry like the features"""
        self.feat, self.stages, self.fps, self.moments = feat, stages, fps, moments

    def tensors(self):
        out = [self.feat] + ([self.moments] if self.moments is not None else [])
        for g in self.stages:
            out += [t for t in (g.fps_idx, g.new_center, g.group_idx) if t is not None]
            if g.csr is not None:
                out += list(g.csr)
        for f in self.fps:
            out += [f[0], f[1]] + (list(f[2]) if (len(f) > 2 and f[2] is not None) else [])
        return out

    def clone(self):
        from modules.repsurface_utils import StageGeometry
        return SegGeoState(self.feat.clone(),
                           [StageGeometry(None if g.fps_idx is None else g.fps_idx.clone(), g.new_center.clone(), g.new_offset,
                                          g.group_idx.clone(), None if g.csr is None else tuple(t.clone() for t in g.csr)) for g in self.stages],
                           [(f[0].clone(), f[1].clone(), None if (len(f) < 3 or f[2] is None) else tuple(t.clone() for t in f[2])) for f in self.fps],
                           None if self.moments is None else self.moments.clone())

    def copy_(self, other):
        dst, src = self.tensors(), other.tensors()
        if len(dst) != len(src) or any(d.dtype != s_.dtype or d.shape != s_.shape for d, s_ in zip(dst, src)):
            raise RuntimeError("SegGeoState.copy_: the two states do not hold the same tensors")
        for dt in sorted({d.dtype for d in dst}, key=str):      # one multi-tensor launch per dtype (one blit per tensor was ~30 launches per step)
            pairs = [(d, s_) for d, s_ in zip(dst, src) if d.dtype == dt]
            torch._foreach_copy_([p[0] for p in pairs], [p[1] for p in pairs])
        return self
