"""Where the ragged segmentation step's time goes under CU-masked streams: geometry alone, network replay alone, both."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, argparse
sys.path.insert(0, os.path.join(os.getcwd(), "repsurf_amd", "segmentation"))
from repsurf_amd import ops
from repsurf_amd.graph import RaggedSegStep
from repsurf_amd.head import CrossEntropyLoss
from repsurf_amd.optim import Adam
from models.repsurf.repsurf_umb_ssg import Model
dev = torch.device("cuda")
torch.manual_seed(0)
model = Model(argparse.Namespace(return_polar=False, in_channel=6, group_size=8, num_class=13)).to(dev).train()
crit = CrossEntropyLoss(ignore_index=255)
opt = Adam(model.parameters(), lr=1e-3)
r = np.random.RandomState(1)
clouds, pts = 16, 4096
sizes = r.randint(pts // 2, pts + 1, clouds); nn = int(sizes.sum())
batch = [torch.from_numpy((r.rand(nn, 3) * 2 - 1).astype(np.float32)).to(dev), torch.from_numpy(r.rand(nn, 3).astype(np.float32)).to(dev),
         ops.offsets_tensor(np.cumsum(sizes).tolist(), dev)]
label = torch.from_numpy(r.randint(0, 13, nn).astype(np.int64)).to(dev)
rs = RaggedSegStep(model, crit, opt, batch, label, capacity=clouds * pts, max_cloud_rows=pts, overlap=True)

def wall(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3), round(th / n * 1e3, 3)

def geo():
    with torch.cuda.stream(rs.side):
        rs._prepare(1, batch, label)
def net():
    with torch.cuda.stream(rs.main):
        rs.g_net[0].replay()
def both():
    net(); geo()
print("geometry alone  (ms wall, ms host)", wall(geo))
print("network replay alone", wall(net))
print("both, no host waits", wall(both))
