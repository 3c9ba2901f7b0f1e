#!/usr/bin/env python3
"""Is the stale read reproducible WITHOUT any kernel of this package on the reading side?  Side stream: a producer (torch copy of one of two
different sources into a fixed buffer) followed by a consumer (torch elementwise kernel reading that buffer); main stream: (A) nothing,
(B) eager torch matmuls, (C) the replayed classification network hipGraph (this package's kernels), (D) a replayed hipGraph of torch
matmuls only.  Counts consumer outputs that are not the source just copied.  (GPU box)"""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
torch.manual_seed(0)
srcs = [torch.randn(32 * 1024, 3, device=dev), torch.randn(32 * 1024, 3, device=dev)]
buf = torch.zeros_like(srcs[0])
side, main = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev)


def run(tag, beside):
    bad = 0
    for t in range(N):
        beside()
        with torch.cuda.stream(side):
            buf.copy_(srcs[t % 2], non_blocking=True)
            out = buf * 1.0
            out2 = out + 0.0
        torch.cuda.synchronize()
        if not (torch.equal(out, srcs[t % 2]) and torch.equal(out2, srcs[t % 2])):
            bad += 1
    print(f"{tag}: stale consumer outputs {bad} of {N}")


run("A nothing on the other stream", lambda: None)


def mm():
    with torch.cuda.stream(main):
        for _ in range(3):
            a @ a
run("B eager torch matmuls on the other stream", mm)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=main):
    x = a
    for _ in range(3):
        x = x @ a


def rg():
    with torch.cuda.stream(main):
        g.replay()
run("D a replayed hipGraph of torch matmuls on the other stream", rg)
import bench
from repsurf_amd.graph import GraphedStep
from util.utils import SmoothClsLoss
Model = importlib.import_module("models.repsurf.repsurf_ssg_umb").Model
model = Model(bench.model_args()).to(dev).train()
pts, lab = bench.synthetic_batch(125, 32, 1024, dev)
with torch.cuda.stream(main):
    gs = GraphedStep(model, SmoothClsLoss(), None, pts, lab, warmup=2)


def rn():
    with torch.cuda.stream(main):
        gs.graph.replay()
run("C the replayed classification step hipGraph on the other stream", rn)
