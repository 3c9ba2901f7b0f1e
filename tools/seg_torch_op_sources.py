"""Which Python lines launch the torch-native (framework) kernels of one segmentation training step -- network part only: the geometry
is computed outside the spy.  A TorchDispatchMode prints every aten op that touches a device tensor with the repsurf_amd frames that
issued it.  GPU box: python tools/seg_torch_op_sources.py"""
import os, sys, importlib, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
seg_root = os.path.join(ROOT, "repsurf_amd", "segmentation")
cls_root = os.path.join(ROOT, "repsurf_amd", "classification")
for m in [k for k in sys.modules if k.split(".")[0] in ("modules", "models", "util")]:
    del sys.modules[m]
if cls_root in sys.path:
    sys.path.remove(cls_root)
sys.path.insert(0, seg_root)
from repsurf_amd import ops, head as _head
from repsurf_amd.optim import Adam
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda")
Model = importlib.import_module("models.repsurf.repsurf_umb_ssg").Model
torch.manual_seed(0)
import argparse
model = Model(argparse.Namespace(return_polar=False, in_channel=6, group_size=8, num_class=13)).to(dev).train()
opt = Adam(model.parameters(), lr=1e-3)
clouds, pts = 16, 4096
n = clouds * pts
rs = np.random.RandomState(7)
coord = torch.from_numpy((rs.rand(n, 3) * 2 - 1).astype(np.float32)).to(dev)
rgb = torch.from_numpy(rs.rand(n, 3).astype(np.float32)).to(dev)
label = torch.from_numpy(rs.randint(0, 13, (n,)).astype(np.int64)).to(dev)
offset = ops.offsets_tensor([pts * (i + 1) for i in range(clouds)], dev)
crit = _head.CrossEntropyLoss(ignore_index=255)
def step():
    opt.zero_grad(set_to_none=True)
    np.random.seed(3)
    geo = model.geometry([coord, rgb, offset])
    loss = crit(model([coord, rgb, offset], geo=geo), label)
    loss.backward(_head.unit_gradient(loss.device))
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
import collections
count = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        full = str(func)
        tens = [a for a in args if torch.is_tensor(a)] + [b for a in args if isinstance(a, (list, tuple)) for b in a if torch.is_tensor(b)]
        view_like = any(k in full for k in ("view", "reshape", "expand", "permute", "transpose", "slice", "select", "unsqueeze", "squeeze", "detach", "alias", "as_strided", "t.default", "_unsafe_view", "empty", "unbind", "split", "is_", "_local_scalar", "stride", "size", "sym_"))
        if tens and any(t.is_cuda for t in tens) and not view_like:
            shapes = [tuple(a.shape) for a in tens[:3]]
            st = [f"{os.path.relpath(f.filename, ROOT)}:{f.lineno} {f.name}" for f in traceback.extract_stack() if "repsurf_amd" in f.filename]
            where = " | ".join(st[-3:]) if st else "(autograd engine)"
            count[(full, where)] += 1
            print(full, shapes, " <- ", where)
        return func(*args, **(kwargs or {}))
np.random.seed(3)
geo = model.geometry([coord, rgb, offset])
torch.cuda.synchronize()
with Spy():
    opt.zero_grad(set_to_none=True)
    loss = crit(model([coord, rgb, offset], geo=geo), label)
    loss.backward(_head.unit_gradient(loss.device))
    opt.step()
torch.cuda.synchronize()
print("==== by source")
for (full, where), n in count.most_common():
    print(n, full, "<-", where)
