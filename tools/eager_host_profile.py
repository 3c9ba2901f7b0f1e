#!/usr/bin/env python3
"""Where the HOST time of an eagerly launched step goes (GPU box): cProfile over a few steps of the reference-shaped loop body
(zero_grad / forward / loss / backward / optimizer.step), classification B=32 x 1024 or ragged segmentation batches.
    python tools/eager_host_profile.py [cls|seg] [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

what = sys.argv[1] if len(sys.argv) > 1 else "cls"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda")
if what == "cls":
    import bench                                   # puts the classification sub-project on sys.path
    from repsurf_amd.optim import Adam
    from util.utils import SmoothClsLoss
    import importlib
    Model = importlib.import_module("models.repsurf.repsurf_ssg_umb").Model
    torch.manual_seed(0)
    model = Model(bench.model_args()).to(dev).train()
    crit = SmoothClsLoss()
    opt = Adam(model.parameters(), lr=1e-3)
    points, label = bench.synthetic_batch(125, 32, 1024, dev)
    batches = [(points, label)]
else:
    sys.path.insert(0, os.path.join(ROOT, "repsurf_amd", "segmentation"))
    import argparse
    from repsurf_amd import ops
    from repsurf_amd.head import CrossEntropyLoss
    from repsurf_amd.optim import Adam
    from models.repsurf.repsurf_umb_ssg import Model
    torch.manual_seed(0)
    model = Model(argparse.Namespace(return_polar=False, in_channel=6, group_size=8, num_class=13)).to(dev).train()
    crit = CrossEntropyLoss(ignore_index=255)
    opt = Adam(model.parameters(), lr=1e-3)
    r = np.random.RandomState(1)
    batches = []
    for i in range(4):
        sizes = r.randint(2048, 4097, 16)
        nn = int(sizes.sum())
        batches.append(([torch.from_numpy((r.rand(nn, 3) * 2 - 1).astype(np.float32)).to(dev), torch.from_numpy(r.rand(nn, 3).astype(np.float32)).to(dev),
                         ops.offsets_tensor(np.cumsum(sizes).tolist(), dev)], torch.from_numpy(r.randint(0, 13, nn).astype(np.int64)).to(dev)))


def step(i):
    inp, lab = batches[i % len(batches)]
    opt.zero_grad()
    loss = crit(model(inp), lab)
    loss.backward()
    opt.step()
    return loss


for i in range(4):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"{what}: {steps} eager steps: host issue time {host / steps * 1e3:.2f} ms per step, with the device drained {total / steps * 1e3:.2f} ms per step")
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(22)
