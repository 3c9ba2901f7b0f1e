#!/usr/bin/env python3
"""Per-parameter gradient of the captured ragged segmentation step against the eager pass on the same batch (GPU box)."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.test_seg_gpu import _seg_model, packed_cloud, dev
from tests.util import subproject
from repsurf_amd import ops as _ops
from repsurf_amd.graph import RaggedSegStep
from repsurf_amd.head import CrossEntropyLoss
cuda = torch.device("cuda")
layouts = [[1024, 700, 513, 900], [600, 1024, 1024, 777], [512, 512, 900, 640], [1000, 333, 1024, 801]]
batches, labels = [], []
for seed, sizes in enumerate(layouts):
    xyz, _ = packed_cloud(20 + seed, sizes)
    r = np.random.RandomState(40 + seed)
    n = sum(sizes)
    batches.append([dev(xyz), dev(r.rand(n, 3).astype(np.float32)), _ops.offsets_tensor(np.cumsum(sizes).tolist(), cuda)])
    lab = r.randint(0, 13, n).astype(np.int64)
    lab[r.rand(n) < 0.05] = 255
    labels.append(dev(lab))
crit = CrossEntropyLoss(ignore_index=255)
with subproject("segmentation"):
    eager = _seg_model()
    eager.surface_constructor.random_inv = False
    twin = copy.deepcopy(eager)
    names = [n for n, _ in eager.named_parameters()]
    step = RaggedSegStep(twin, crit, None, batches[0], labels[0], capacity=4 * 1024)
    for s in range(3):
        b = s % 4
        par = step.parity
        loss = step(batches[(s + 1) % 4], labels[(s + 1) % 4]).item()
        torch.cuda.synchronize()
        for p in eager.parameters():
            p.grad = None
        le = crit(eager(batches[b]), labels[b])
        le.backward()
        torch.cuda.synchronize()
        bad = []
        for nm, pe, g in zip(names, eager.parameters(), step.grads[par]):
            a, c = pe.grad.double().flatten(), g.double().flatten()
            rel = float((a - c).norm() / max(float(a.norm()), 1e-12))
            if rel > 1e-4 or not np.isfinite(rel):
                bad.append((nm, f"{rel:.2e}", f"{float(a.norm()):.2e}"))
        print(f"call {s}: batch {b} parity {par} loss {loss:.7f} eager {le.item():.7f}  params off: {len(bad)}")
        for x in bad[:3]:
            print("    ", x)
