#!/usr/bin/env python3
"""Gradient differences between runs of the capacity-sized segmentation network: eager vs eager (the float atomics' noise floor) and
captured vs eager (GPU box)."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.test_seg_gpu import _seg_model, _ragged_batches
from tests.util import subproject
from repsurf_amd.graph import RaggedSegStep
from repsurf_amd.head import CrossEntropyLoss
layouts, batches, labels = _ragged_batches()
crit = CrossEntropyLoss(ignore_index=255)
with subproject("segmentation"):
    base = _seg_model()
    base.surface_constructor.random_inv = False
    names = [n for n, _ in base.named_parameters()]
    runs = []
    for capture in (False, False, True):
        model = copy.deepcopy(base)
        step = RaggedSegStep(model, crit, None, batches[0], labels[0], capacity=4 * 1024, capture=capture)
        grads = []
        for s in range(6):
            par = step.parity
            step(batches[(s + 1) % 4], labels[(s + 1) % 4]).item()
            torch.cuda.synchronize()
            grads.append([g.detach().clone() for g in (step.grads[par] if capture else [p.grad for p in model.parameters()])])
        step.close()
        runs.append(grads)
    for tag, (ra, rb) in (("eager vs eager", (runs[0], runs[1])), ("graph vs eager", (runs[2], runs[0]))):
        print("==", tag)
        for s, (ga, gb) in enumerate(zip(ra, rb)):
            rels = sorted(((float((a.double() - b.double()).norm() / max(float(a.double().norm()), 1e-30)), float(a.norm()), n) for a, b, n in zip(ga, gb, names)), reverse=True)
            print(f"  call {s}: worst", [(f"{r:.1e}", f"{nm:.1e}", n) for r, nm, n in rels[:3]], " median", f"{rels[len(rels) // 2][0]:.1e}")
