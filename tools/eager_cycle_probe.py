#!/usr/bin/env python3
"""Which objects of one eagerly launched step are only freed by Python's cycle collector (GPU box)?  Reference cycles through an autograd node
pin a whole step's activations until a generation-2 collection (1.5 GB per classification step).   python tools/eager_cycle_probe.py [cls|seg]"""
import gc, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
dev = torch.device("cuda")
what = sys.argv[1] if len(sys.argv) > 1 else "cls"
if what == "cls":
    import bench, importlib
    from repsurf_amd.optim import Adam
    from util.utils import SmoothClsLoss
    Model = importlib.import_module("models.repsurf.repsurf_ssg_umb").Model
    torch.manual_seed(0)
    model = Model(bench.model_args()).to(dev).train()
    crit = SmoothClsLoss()
    inp, lab = bench.synthetic_batch(125, 32, 1024, dev)
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
    r = np.random.RandomState(1)
    sizes = r.randint(2048, 4097, 16); nn = int(sizes.sum())
    inp = [torch.from_numpy((r.rand(nn, 3) * 2 - 1).astype(np.float32)).to(dev), torch.from_numpy(r.rand(nn, 3).astype(np.float32)).to(dev),
           ops.offsets_tensor(np.cumsum(sizes).tolist(), dev)]
    lab = torch.from_numpy(r.randint(0, 13, nn).astype(np.int64)).to(dev)
opt = Adam(model.parameters(), lr=1e-3)


def step():
    opt.zero_grad()
    loss = crit(model(inp), lab)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
gc.collect()
gc.disable()
a0 = torch.cuda.memory_allocated()
step()
torch.cuda.synchronize()
a1 = torch.cuda.memory_allocated()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
gc.set_debug(0)
print(f"{what}: one step left {(a1 - a0) >> 20} MiB allocated that only the cycle collector frees; {n} unreachable objects")
c = collections.Counter(type(o).__name__ for o in gc.garbage)
print(c.most_common(12))
for o in gc.garbage:
    tn = type(o).__name__
    if tn.endswith("Backward"):
        def walk(v, path, depth=0):
            if torch.is_tensor(v):
                if v.grad_fn is o:
                    print(f"  {tn}: {path} is an OUTPUT of this node (cycle)")
            elif isinstance(v, dict) and depth < 3:
                for k, x in v.items():
                    walk(x, f"{path}[{k!r}]", depth + 1)
            elif isinstance(v, (list, tuple)) and depth < 3:
                for i, x in enumerate(v):
                    walk(x, f"{path}[{i}]", depth + 1)
            elif hasattr(v, "__slots__") and depth < 3:
                for k in v.__slots__:
                    walk(getattr(v, k, None), f"{path}.{k}", depth + 1)
            elif hasattr(v, "__dict__") and depth < 3 and not isinstance(v, torch.nn.Module):
                for k, x in vars(v).items():
                    walk(x, f"{path}.{k}", depth + 1)
        for k, v in vars(o).items():
            walk(v, k)
gc.garbage.clear()
