"""1-rank timing of the sharded (N > 1) step forms against the N = 1 form: what a rank pays for the flat-gradient path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "repsurf_amd", "classification"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
from repsurf_amd.graph import PipelinedStep, ShardedGraphedStep
from repsurf_amd.optim import Adam
from util.utils import SmoothClsLoss
from models.repsurf.repsurf_ssg_umb import Model
dev = torch.device("cuda", 0)
pts, lab = bench.synthetic_batch(1000, 32, 1024, dev)
def run(kind):
    torch.manual_seed(0)
    m = Model(bench.model_args()).to(dev).train()
    opt = Adam(m.parameters(), lr=1e-3)
    if kind == "pipe":
        s = PipelinedStep(m, SmoothClsLoss(), opt, pts, lab, warmup=3); f = lambda: s(sync=False)
    elif kind == "pipe_sharded":
        s = PipelinedStep(m, SmoothClsLoss(), opt, pts, lab, warmup=3, sharded=True); f = lambda: s(sync=False)
    else:
        s = ShardedGraphedStep(m, SmoothClsLoss(), opt, pts, lab, warmup=3); f = s
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    print(f"{kind:14s} {dt*1e3:.3f} ms/step  {32/dt:.0f} clouds/s", flush=True)
    s = f = None          # (the graphs with recorded collectives go before the communicator)
for k in sys.argv[1:] or ["pipe", "pipe_sharded", "two_graph"]:
    run(k)
from repsurf_amd import dist as rdist
rdist.finish()
