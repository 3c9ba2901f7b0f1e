#!/usr/bin/env python3
"""rs_umbrella_fan_offset (fan_packed_kernel<9, true>: one thread per point, no LDS, no atomics) run twice on the same inputs must give the same
bits.  (A) alone on one stream; (B) beside a replayed hipGraph of the segmentation network on another stream; (C) beside plain torch matmuls on
another stream.  Counts differing runs and the shape of the difference.  (GPU box)"""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
c, off = batches[3][0], batches[3][2]
idx, _ = ops.knnquery_offset(9, c, c, off, off)
ref = ops.umbrella_fan_offset(c, c, idx, off, None, True).clone()
torch.cuda.synchronize()


def run(tag, beside):
    side = torch.cuda.Stream()
    bad, shapes = 0, []
    for t in range(N):
        beside(t)
        with torch.cuda.stream(side):
            f = ops.umbrella_fan_offset(c, c, idx, off, None, True)
        if t % 8 == 7 or True:
            torch.cuda.synchronize()
            if not torch.equal(f, ref):
                bad += 1
                rows = torch.nonzero((f != ref).flatten(1).any(1)).flatten()
                shapes.append((int(rows[0]), int(rows.numel())))
    print(tag, "differing runs", bad, "of", N, shapes[:6])


run("A alone", lambda t: None)
a = torch.randn(2048, 2048, device="cuda")
main = torch.cuda.Stream()


def mm(t):
    with torch.cuda.stream(main):
        for _ in range(4):
            a @ a
run("C beside torch matmuls", mm)
crit = CrossEntropyLoss(ignore_index=255)
with subproject("segmentation"):
    base = _seg_model()
    base.surface_constructor.random_inv = False
    step = RaggedSegStep(copy.deepcopy(base), crit, None, batches[0], labels[0], capacity=4096)

    def rep(t):
        with torch.cuda.stream(step.main):
            step.g_net[t % 2].replay()
    run("B beside the replayed network graph", rep)
