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
        return out + [self.classifier[0]]

    def forward(self, pos_feat_off0):
        coord, feat, offset = pos_feat_off0            # (N,3), (N,C_in-3), (B,) running row ends
        if coord.is_cuda and self.training and torch.is_grad_enabled():
            _mlp.prepack(self._packed_layers())        # ~23 weight-pack launches of the step in one
        level0 = [coord, self.surface_constructor(coord, offset), torch.cat([coord, feat], 1), offset]
        level1 = self.sa1(level0)
        level2 = self.sa2(level1)
        level3 = self.sa3(level2)
        level4 = self.sa4(level3)

        def pfo(level):                                # [center, normal, feature, offset] -> [center, feature, offset]
            return [level[0], level[2], level[3]]
        f3 = self.fp4(pfo(level3), pfo(level4))
        f2 = self.fp3(pfo(level2), [level3[0], f3, level3[3]])
        f1 = self.fp2(pfo(level1), [level2[0], f2, level2[3]])
        f0 = self.fp1([coord, None, offset], [level1[0], f1, level1[3]])
        cls = self.classifier                          # Linear-BN-ReLU on the fused kernels, then Dropout and the output Linear
        return cls[4](cls[3](row_mlp(f0, [cls[0]], [cls[1]])))
