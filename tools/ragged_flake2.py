#!/usr/bin/env python3
"""Is the constructor's geometry (packed kNN-9 through the per-cloud grids + fan features) wrong when it runs on a side stream beside a replayed
network graph?  (GPU box)"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.test_seg_gpu import _seg_model, _ragged_batches
from tests.util import subproject
from repsurf_amd import ops
from repsurf_amd.graph import RaggedSegStep
from repsurf_amd.head import CrossEntropyLoss
layouts, batches, labels = _ragged_batches()
crit = CrossEntropyLoss(ignore_index=255)
with subproject("segmentation"):
    base = _seg_model()
    base.surface_constructor.random_inv = False
    sc = base.surface_constructor
    truth = []
    for b in batches:
        idx, d2 = ops.knnquery_offset(9, b[0], b[0], b[2], b[2])
        truth.append((idx.clone(), d2.clone(), sc.features(b[0], b[2]).clone()))
    torch.cuda.synchronize()
    step = RaggedSegStep(copy.deepcopy(base), crit, None, batches[0], labels[0], capacity=4096)
    side = torch.cuda.Stream()
    bad = {"idx": 0, "d2": 0, "feat": 0}
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for t in range(trials):
        b = t % 4
        with torch.cuda.stream(step.main):
            step.g_net[t % 2].replay()
        with torch.cuda.stream(side):
            idx, d2 = ops.knnquery_offset(9, batches[b][0], batches[b][0], batches[b][2], batches[b][2])
            feat = sc.features(batches[b][0], batches[b][2])
        torch.cuda.synchronize()
        e_idx, e_d2, e_feat = not torch.equal(idx, truth[b][0]), not torch.equal(d2, truth[b][1]), not torch.equal(feat, truth[b][2])
        bad["idx"] += e_idx; bad["d2"] += e_d2; bad["feat"] += e_feat
        if e_idx or e_d2 or e_feat:
            rows = torch.nonzero((idx != truth[b][0]).any(1)).flatten()
            frows = torch.nonzero((feat != truth[b][2]).flatten(1).any(1)).flatten()
            print(f"trial {t} batch {b}: idx rows differing {rows.numel()} {rows[:8].tolist()}  feat rows differing {frows.numel()} {frows[:8].tolist()}  (rows of this batch: {batches[b][0].shape[0]}, cloud ends {ops.host_offsets(batches[b][2])})")
    print("mismatches", bad, "of", trials)
