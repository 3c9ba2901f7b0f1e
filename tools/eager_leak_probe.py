#!/usr/bin/env python3
"""Does the eagerly launched classification step keep memory alive from step to step (GPU box)?  Allocated / reserved bytes, live CUDA tensors
(gc) and the host time of every step; a cProfile of one late step.    python tools/eager_leak_probe.py [steps]"""
import cProfile, gc, os, pstats, sys, time, collections
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70


def step():
    opt.zero_grad()
    loss = crit(model(points), label)
    loss.backward()
    opt.step()


def live():
    c = collections.Counter()
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                c[(tuple(o.shape), str(o.dtype))] += 1
        except Exception:
            pass
    return c


prev = None
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if i == n - 3:
        pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
    else:
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    if i % 5 == 0 or i >= n - 3:
        st = torch.cuda.memory_stats()
        c = live()
        tot = sum(c.values())
        print(f"step {i}: {1e3 * (t1 - t0):.1f} ms  allocated {torch.cuda.memory_allocated() >> 20} MiB reserved {torch.cuda.memory_reserved() >> 20} MiB  "
              f"segments {st.get('segment.all.current')} live cuda tensors {tot}")
        if prev is not None:
            d = {k: v - prev.get(k, 0) for k, v in c.items() if v != prev.get(k, 0)}
            if d:
                print("   grew:", sorted(d.items(), key=lambda kv: -abs(kv[1]))[:8])
        prev = c
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
