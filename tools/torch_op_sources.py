"""Which Python lines launch the torch-native kernels (fills, copies, adds, cats) of one classification training step?
A TorchDispatchMode prints every aten fill / copy / add / cat / zeros ... of one eager step with the repsurf_amd frames that
issued it ("(autograd engine)" = gradient accumulation).  GPU box: python tools/torch_op_sources.py"""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench                                   # puts the classification sub-project on sys.path
from repsurf_amd import head as _head
from repsurf_amd.optim import Adam
from util.utils import SmoothClsLoss
dev = torch.device("cuda")
Model = importlib.import_module("models.repsurf.repsurf_ssg_umb").Model
torch.manual_seed(0)
model = Model(bench.model_args()).to(dev).train()
crit = SmoothClsLoss()
opt = Adam(model.parameters(), lr=1e-3)
points, label = bench.synthetic_batch(125, 32, 1024, dev)
def step():
    opt.zero_grad(set_to_none=True)
    geo = model.geometry(points, fork=False)
    loss = crit(model(points, geo=geo), label)
    loss.backward(_head.unit_gradient(loss.device))
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        full = str(func)
        if any(k in full for k in ("fill_", "zero_", "copy_", "aten.add", "cat", "zeros", "clone", "mul", "sum", "ones", "full", "_foreach", "contiguous", "index")):
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)]
            st = [f"{os.path.relpath(f.filename, ROOT)}:{f.lineno} {f.name}" for f in traceback.extract_stack() if ("repsurf_amd" in f.filename or "tools/" in f.filename) and "torch_op_sources" not in f.filename]
            print(full, shapes, " <- ", " | ".join(st[-3:]) if st else "(autograd engine)")
        return func(*args, **(kwargs or {}))
with Spy():
    step()
torch.cuda.synchronize()
