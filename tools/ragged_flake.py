#!/usr/bin/env python3
"""Hunt for the intermittent wrong loss of the captured RaggedSegStep (seen on the first runs on a fresh GPU box): many fresh steps, every
loss against the eager capacity-sized run; on a mismatch, which input of the graph differs from what it should hold."""
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
    ref = RaggedSegStep(copy.deepcopy(base), crit, None, batches[0], labels[0], capacity=4096, capture=False)
    want = [ref(batches[(s + 1) % 4], labels[(s + 1) % 4]).item() for s in range(8)]
    ref.close()
    nbad = 0
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        model = copy.deepcopy(base)
        step = RaggedSegStep(model, crit, None, batches[0], labels[0], capacity=4096, overlap=os.environ.get('OVERLAP', '0') == '1')
        for s in range(8):
            b = s % 4
            p = step.parity
            got = step(batches[(s + 1) % 4], labels[(s + 1) % 4]).item()
            torch.cuda.synchronize()
            if got != want[s]:
                nbad += 1
                n0 = sum(layouts[b])
                print(f"trial {trial} call {s} batch {b} parity {p}: loss {got} != {want[s]}  counts {step.counts[p]}  table {step.caps[p].table.tolist()}")
                print("   coord equal", bool(torch.equal(step.coord[p][:n0], batches[b][0])), " feat equal", bool(torch.equal(step.feat[p][:n0], batches[b][1])),
                      " label equal", bool(torch.equal(step.label[p][:n0], labels[b])), " label tail ignore", bool((step.label[p][n0:] == 255).all()))
                fresh = model.geometry([batches[b][0], batches[b][1], batches[b][2]])
                st = step.state[p]
                print("   feat", bool(torch.equal(st.feat[:n0], fresh.feat)), "moments", None if fresh.moments is None else bool(torch.equal(st.moments, fresh.moments)))
                dr = torch.nonzero((st.feat[:n0] != fresh.feat).flatten(1).any(1)).flatten()
                if dr.numel():
                    prev_b = (b - 2) % 4                       # the batch this parity held before
                    prev = model.geometry([batches[prev_b][0], batches[prev_b][1], batches[prev_b][2]]).feat
                    k = min(prev.shape[0], n0)
                    stale = int(((st.feat[:k] == prev[:k]).flatten(1).all(1) & (st.feat[:k] != fresh.feat[:k]).flatten(1).any(1)).sum())
                    print(f"   feat rows differing: {dr.numel()} of {n0}, first {dr[:6].tolist()} last {dr[-3:].tolist()}; of them equal to the PREVIOUS batch of this parity (batch {prev_b}): {stale};"
                          f" max abs diff {float((st.feat[:n0] - fresh.feat).abs().max()):.3e}; nan in state {bool(torch.isnan(st.feat[:n0]).any())}")
                    mom2 = None
                    from repsurf_amd import mlp as _mlp
                    print("   moments of the state's feat == state's moments:", bool(torch.equal(_mlp.umbrella_moments(st.feat[:n0].reshape(-1, 10)), st.moments)))
                for li, (a, g) in enumerate(zip(st.stages, fresh.stages)):
                    m = g.fps_idx.shape[0]
                    print(f"   stage {li}: fps {bool(torch.equal(a.fps_idx[:m], g.fps_idx))} centre {bool(torch.equal(a.new_center[:m], g.new_center))} idx {bool(torch.equal(a.group_idx[:m], g.group_idx))}"
                          f" csr_off {bool(torch.equal(a.csr[0][:g.csr[0].shape[0]], g.csr[0]))} csr_edges {bool(torch.equal(a.csr[1][:g.csr[1].shape[0]], g.csr[1]))}")
                for li, (a, g) in enumerate(zip(st.fps, fresh.fps)):
                    n = g[0].shape[0]
                    print(f"   fp {li}: idx {bool(torch.equal(a[0][:n], g[0]))} weight {bool(torch.equal(a[1][:n], g[1]))}")
                again = step.g_net[p].replay()
                torch.cuda.synchronize()
                print("   replayed again:", step.loss[p].item())
        step.close()
    print("mismatches:", nbad)
