import os, sys, time, copy
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
batches = []
for i in range(8):
    sizes = r.randint(pts // 2, pts + 1, clouds); nn = int(sizes.sum())
    batches.append(([torch.from_numpy((r.rand(nn, 3) * 2 - 1).astype(np.float32)).to(dev), torch.from_numpy(r.rand(nn, 3).astype(np.float32)).to(dev),
                     ops.offsets_tensor(np.cumsum(sizes).tolist(), dev)], torch.from_numpy(r.randint(0, 13, nn).astype(np.int64)).to(dev)))
def timed(overlap, steps=40):
    rs = RaggedSegStep(model, crit, opt, batches[0][0], batches[0][1], capacity=clouds * pts, max_cloud_rows=pts, overlap=overlap)
    for i in range(8):
        rs(batches[(i + 1) % 8][0], batches[(i + 1) % 8][1], sync=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        rs(batches[(i + 1) % 8][0], batches[(i + 1) % 8][1], sync=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    ids = (rs.main.cuda_stream, rs.side.cuda_stream)
    rs.close()
    return round(dt, 3), ids
for ov in sys.argv[1:]:
    print("overlap", ov, timed(ov == "1"))
