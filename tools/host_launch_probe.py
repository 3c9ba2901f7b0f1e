#!/usr/bin/env python3
"""Is the replayed step bound by the host?  Times the host side of N step() calls (enqueue only) against the wall time
including the final synchronize, for GraphedStep and PipelinedStep.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "repsurf_amd", "classification"))
import torch
import bench
from repsurf_amd.graph import GraphedStep, PipelinedStep
from repsurf_amd.optim import Adam
import importlib

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
Model = importlib.import_module("models.repsurf.repsurf_ssg_umb").Model
from util.utils import SmoothClsLoss
points, label = bench.synthetic_batch(125, 32, 1024, dev)
for name, cls in (("GraphedStep", GraphedStep), ("PipelinedStep", PipelinedStep)):
    torch.manual_seed(0)
    model = Model(bench.model_args()).to(dev).train()
    opt = Adam(model.parameters(), lr=1e-3)
    step = cls(model, SmoothClsLoss(), opt, points, label, warmup=2)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:14s} host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, wall {1e3 * (t2 - t0) / n:.3f} ms/step", flush=True)
    g = getattr(step, "graph", None) or step.g_net[0]
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:14s} bare graph.replay(): host {1e3 * (t1 - t0) / n:.3f} ms, wall {1e3 * (t2 - t0) / n:.3f} ms", flush=True)
