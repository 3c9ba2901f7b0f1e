import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repsurf_amd import ops
dev = torch.device("cuda")
b, n, m, r, ns = 2048, 1024, 512, float(sys.argv[1]) if len(sys.argv) > 1 else 0.2, 32
g = torch.Generator().manual_seed(b)
xyz = (torch.rand(b, n, 3, generator=g) * 2 - 1).to(dev)
centres = xyz[:, torch.randperm(n, generator=g)[:m]].contiguous()
for _ in range(3):
    ops.ballquery(r, ns, xyz, centres, return_count=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.ballquery(r, ns, xyz, centres, return_count=True)
e1.record(); torch.cuda.synchronize()
print("dbg", os.environ.get("RS_BALLQUERY_DBG", "0"), "r", r, ":", round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us")
