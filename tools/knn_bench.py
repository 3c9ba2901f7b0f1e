#!/usr/bin/env python3
"""rs_umbrella_features (kNN k=9 + fan features) at the step's shape, 30 launches back to back.  Run on the GPU box;
REPSURF_HIP_LIB selects a variant library (build_exp/librepsurf_knn<U>.so = U candidates per insertion round)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repsurf_amd import ops
torch.manual_seed(0)
xyz = (torch.rand(32, 1024, 3) * 2 - 1).cuda()
flip = torch.ones(32).cuda()
for _ in range(3):
    ops.umbrella_features(xyz, 9, flip)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    ops.umbrella_features(xyz, 9, flip)
e1.record()
torch.cuda.synchronize()
print(os.environ.get("REPSURF_HIP_LIB", "default"), f"{e0.elapsed_time(e1) / 30 * 1e3:.1f} us per launch")
