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
step = PipelinedStep(model, crit, None, b0[0], b0[1], warmup=2)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seen = {0: {}, 1: {}}
cur = 0                                   # the batch the NEXT call trains on (b0 sits in both buffers at first)
for s in range(N):
    nxt = (s + 1) % 2
    nb = b1 if nxt else b0
    loss = step(nb[0], nb[1]).item()
    seen[cur][loss] = seen[cur].get(loss, 0) + 1
    cur = nxt
print("batch 0 losses", seen[0])
print("batch 1 losses", seen[1])
