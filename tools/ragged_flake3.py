#!/usr/bin/env python3
"""Inside RaggedSegStep._prepare (side stream, beside a running network graph): is the constructor's kNN / fan-feature result reproducible when
computed again on the same stream?  Which of (coordinates read, kNN lists, fan features) differs when it is not?  (GPU box)"""
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
stats = {"calls": 0, "idx": 0, "feat_same_idx": 0, "coord": 0}
TRUTH = {}
for b_ in batches:
    i_, _ = ops.knnquery_offset(9, b_[0], b_[0], b_[2], b_[2])
    TRUTH[int(b_[0].shape[0])] = ops.umbrella_fan_offset(b_[0], b_[0], i_, b_[2], None, True).clone()
torch.cuda.synchronize()
with subproject("segmentation"):
    base = _seg_model()
    base.surface_constructor.random_inv = False
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
        model = copy.deepcopy(base)
        step = RaggedSegStep(model, crit, None, batches[0], labels[0], capacity=4096, overlap=os.environ.get('OVERLAP', '0') == '1')
        orig = step._prepare

        def checked(q, batch, label, first=False, _o=orig, _s=step):
            _o(q, batch, label, first)
            if os.environ.get("SERIAL", "0") == "1":
                torch.cuda.current_stream().wait_stream(_s.main)      # the checks below run with the network graph finished
            n0 = _s.counts[q][0]
            c = _s.coord[q][:n0]
            off = batch[2]
            i1, _ = ops.knnquery_offset(9, c, c, off, off)
            i2, _ = ops.knnquery_offset(9, c, c, off, off)
            f1 = ops.umbrella_fan_offset(c, c, i1, off, None, True)
            f2 = ops.umbrella_fan_offset(c, c, i1, off, None, True)
            stats["calls"] += 1
            stats["coord"] += int(not torch.equal(c, batch[0]))
            stats["idx"] += int(not torch.equal(i1, i2))
            stats["feat_same_idx"] += int(not torch.equal(f1, f2))
            if not torch.equal(f1, f2):
                truth = TRUTH[int(n0)]
                w = "f1" if not torch.equal(f1, truth) else "f2"
                fw = f1 if w == "f1" else f2
                rows = torch.nonzero((fw != truth).flatten(1).any(1)).flatten()
                r0 = int(rows[0])
                # is the wrong content the truth of ANOTHER batch at the same rows (stale memory), or of other rows of this batch?
                stale = [k for k, t in TRUTH.items() if k != int(n0) and t.shape[0] > int(rows[-1]) and torch.equal(t[rows], fw[rows])]
                nanc = int(torch.isnan(fw[rows]).sum())
                # which triangles (10 columns each) of the wrong rows differ, and is a wrong row the truth of ANOTHER row of this batch?
                tri = (fw[rows] != truth[rows]).reshape(rows.numel(), -1, 10).any(2).sum(1).tolist()
                moved = []
                for r_ in rows[:4].tolist():
                    hit = torch.nonzero((truth.flatten(1) == fw[r_].flatten()).all(1)).flatten().tolist()
                    moved.append((r_, hit[:2]))
                print(f"   triangles differing per wrong row {tri[:16]}; wrong row == truth of rows {moved}")
                print(f"   {w} wrong in rows {r0}..{int(rows[-1])} ({rows.numel()}); equals the features of a batch with {stale} rows at the same positions; nans {nanc}; "
                      f"wrong[0][:6] {fw[r0].flatten()[:6].tolist()} truth {truth[r0].flatten()[:6].tolist()}")
            if not torch.equal(_s.state[q].feat[:n0], f1):
                bad = torch.nonzero((_s.state[q].feat[:n0] != f1).flatten(1).any(1)).flatten()
                print(f"trial {trial}: state feat != recomputed in rows {bad[:4].tolist()}..{bad[-1:].tolist()} ({bad.numel()})")
        step._prepare = checked
        for s in range(8):
            step(batches[(s + 1) % 4], labels[(s + 1) % 4], sync=False)
        torch.cuda.synchronize()
        step.close()
print(stats)
