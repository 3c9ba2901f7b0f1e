"""600 pipelined training steps on one fixed batch: the loss must fall, parameters stay finite, allocated memory stay flat, and
nothing may be left in the deferred-reduction queue or the per-pass zero pool between steps (all ASSERTED).
GPU box: python tools/soak.py"""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from repsurf_amd.graph import PipelinedStep
from repsurf_amd.optim import Adam
from repsurf_amd import mlp_hip, zeros
from util.utils import SmoothClsLoss
dev = torch.device("cuda")
Model = importlib.import_module("models.repsurf.repsurf_ssg_umb").Model
torch.manual_seed(0)
model = Model(bench.model_args()).to(dev).train()
opt = Adam(model.parameters(), lr=1e-3)
points, label = bench.synthetic_batch(125, 32, 1024, dev)
step = PipelinedStep(model, SmoothClsLoss(), opt, points, label, warmup=3)
losses = []
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
for i in range(600):
    l = step(points, label, sync=False)
    if i % 100 == 99:
        torch.cuda.synchronize(); losses.append(l.item())
        assert len(mlp_hip._pending_reduce) == 0 and len(zeros._pool) == 0, "deferred work left between steps"
        print(i + 1, "loss", losses[-1], "allocated MB", torch.cuda.memory_allocated() / 1e6, "pending", len(mlp_hip._pending_reduce), "pool", len(zeros._pool), flush=True)
torch.cuda.synchronize()
assert all(torch.isfinite(p).all() for p in model.parameters())
growth = (torch.cuda.memory_allocated() - m0) / 1e6
print("memory growth MB", growth, "loss first/last", losses[0], losses[-1])
assert losses[-1] < losses[0], "the loss did not fall"
assert abs(growth) < 1.0, "allocated memory is not flat"
print("soak ok")
