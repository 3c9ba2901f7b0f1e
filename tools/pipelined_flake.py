#!/usr/bin/env python3
"""Does PipelinedStep (eager input copies + draw refill, then the geometry GRAPH, on the side stream beside the network GRAPH) ever train on
stale inputs?  Two different batches alternate, no optimizer: every loss must be one of two values, bit for bit.  (GPU box)"""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from repsurf_amd import rng
from repsurf_amd.graph import PipelinedStep
from util.utils import SmoothClsLoss
dev = torch.device("cuda")
Model = importlib.import_module("models.repsurf.repsurf_ssg_umb").Model
torch.manual_seed(0)
model = Model(bench.model_args()).to(dev).train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
crit = SmoothClsLoss()
b0 = bench.synthetic_batch(125, 32, 1024, dev)
b1 = bench.synthetic_batch(777, 32, 1024, dev)
rng._cpu_draw_orig = rng._cpu_draw
rng._cpu_draw = lambda kind, b, n: (torch.ones(b) if kind == "flip" else torch.zeros(b, dtype=torch.int32))      # fixed draws: the loss depends on the batch only
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
if len(sys.argv) > 2 and sys.argv[2] == "single":
    # ONE graph on one stream (GraphedStep): geometry and network in sequence, the inputs copied eagerly in front of each replay
    from repsurf_amd.graph import GraphedStep
    g = GraphedStep(model, crit, None, b0[0].clone(), b0[1].clone(), warmup=2)
    seen = {0: {}, 1: {}}
    for s in range(N):
        k = s % 2
        nb = b1 if k else b0
        g.points.copy_(nb[0]); g.label.copy_(nb[1])
        loss = g().item()
        seen[k][loss] = seen[k].get(loss, 0) + 1
    print("single graph, batch 0 losses", seen[0])
    print("single graph, batch 1 losses", seen[1])
    sys.exit(0)
step = PipelinedStep(model, crit, None, b0[0], b0[1], warmup=2)
seen = {0: {}, 1: {}}
cur = 0                                   # the batch the NEXT call trains on (b0 sits in both buffers at first)
ref_state, major = {}, {}
geo_bad = net_bad = 0
for s in range(N):
    nxt = (s + 1) % 2
    nb = b1 if nxt else b0
    p = step.parity
    loss = step(nb[0], nb[1]).item()
    torch.cuda.synchronize()
    seen[cur][loss] = seen[cur].get(loss, 0) + 1
    if s in (4, 5):                       # steady state: the geometry state the network of this parity just read
        ref_state[p] = [t.clone() for t in step.state[p].tensors()]
        major[p] = loss
    elif s > 5 and loss != major[p]:
        # state[p] is rewritten by the geometry graph that runs beside the NEXT network of the other parity: still intact here? it was
        # consumed by the network that just ran; the geometry replay enqueued in this call writes state[1 - p]
        diff = [(i, int((a != b).sum())) for i, (a, b) in enumerate(zip(step.state[p].tensors(), ref_state[p])) if not torch.equal(a, b)]
        if diff:
            geo_bad += 1
            if geo_bad <= 6:
                a, b = step.state[p].tensors()[0], ref_state[p][0]
                a2, b2 = a.reshape(-1, a.shape[-2], a.shape[-1]), b.reshape(-1, b.shape[-2], b.shape[-1])      # (points, fan, 10)
                pts_bad = torch.nonzero((a2 != b2).flatten(1).any(1)).flatten()
                print(f"   shape {tuple(a.shape)}; points differing {pts_bad.numel()}: {pts_bad[:20].tolist()}")
                q = int(pts_bad[0])
                ch = (a2[q] != b2[q])
                print(f"   point {q}: channels differing per triangle {ch.sum(1).tolist()}; columns differing {ch.any(0).tolist()}")
                print("     got ", [round(v, 5) for v in a2[q, 0].tolist()])
                print("     want", [round(v, 5) for v in b2[q, 0].tolist()])
                # is the wrong fan a permutation / other start of the right one?  compare the SET of centroids (columns 0..2 in the classification order)
                same_set = sorted(map(tuple, a2[q, :, 0:3].round(decimals=5).tolist())) == sorted(map(tuple, b2[q, :, 0:3].round(decimals=5).tolist()))
                print("     same set of triangle centroids (another order / sign only):", same_set)
        else:
            net_bad += 1
        if geo_bad + net_bad <= 12:
            print(f"step {s} parity {p}: loss {loss} (usual {major[p]}); geometry state tensors differing from the reference: {diff if diff else 'none -> the NETWORK graph computed something else'}")
    cur = nxt
print("deviating steps: geometry state differed", geo_bad, "; geometry state identical (network side)", net_bad)
print("batch 0 losses", seen[0])
print("batch 1 losses", seen[1])
