#!/usr/bin/env python3
"""GPU time of one repsurf_amd.optim.Adam step over the classifier's parameter list (replayed hipGraph of 20 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repsurf_amd import optim

torch.manual_seed(0)
shapes = [(10, 10), (10,), (10,), (10, 10), (10,), (10,), (10, 10), (10,)]
for cin_p, cin_f, widths in ((6, 10, [64, 64, 128]), (6, 138, [128, 128, 256]), (6, 266, [256, 512, 1024])):
    shapes += [(widths[0], cin_p), (widths[0],), (widths[0], cin_f), (widths[0],), (widths[0],), (widths[0],), (widths[0],), (widths[0],)]
    for a, b in zip(widths[:-1], widths[1:]):
        shapes += [(b, a), (b,), (b,), (b,)]
shapes += [(512, 1024), (512,), (512,), (512,), (256, 512), (256,), (256,), (256,), (15, 256), (15,)]
ps = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
print(len(ps), "tensors", sum(p.numel() for p in ps), "elements")
for p in ps:
    p.grad = torch.randn_like(p)
opt = optim.Adam(ps, lr=1e-3, weight_decay=1e-4)
for _ in range(3):
    opt.step()
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for _ in range(20):
            opt.step()
torch.cuda.current_stream().wait_stream(side)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 200 * 1e3
n = sum(p.numel() for p in ps)
print(f"adam step: {us:.1f} us  ({7 * 4 * n / us / 1e3:.0f} GB/s of 7 streams)")
