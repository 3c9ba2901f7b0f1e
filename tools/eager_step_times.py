#!/usr/bin/env python3
"""Per-step host time of the eagerly launched classification step (GPU box): does the eager loop reach a steady state, and when?
    python tools/eager_step_times.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from repsurf_amd.optim import Adam
from util.utils import SmoothClsLoss
import importlib
dev = torch.device("cuda")
Model = importlib.import_module("models.repsurf.repsurf_ssg_umb").Model
torch.manual_seed(0)
model = Model(bench.model_args()).to(dev).train()
crit = SmoothClsLoss()
opt = Adam(model.parameters(), lr=1e-3)
points, label = bench.synthetic_batch(125, 32, 1024, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ts = []
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad()
    loss = crit(model(points), label)
    loss.backward()
    opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("step: host issue ms / with device drained ms")
print(" ".join(f"{a:.1f}/{b:.1f}" for a, b in ts))
st = torch.cuda.memory_stats()
print("allocator: num_alloc_retries", st.get("num_alloc_retries"), "segments", st.get("segment.all.current"), "device mallocs", st.get("num_device_alloc"), "device frees", st.get("num_device_free"))
