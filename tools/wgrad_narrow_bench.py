#!/usr/bin/env python3
"""First-layer weight gradients of the segmentation step (524 288 dense rows, 32 output columns, 16 / 3 input channels, BatchNorm-backward
operand): rs_mlp_wgrad timed with HIP events, 20 back-to-back launches, cold (a 600 MB buffer written between launches) and warm.
(The 16-byte-load variant this compared against was removed after the measurement: profiles/r04/wgrad_narrow_bench.txt;
REPSURF_WGRAD_CHUNKS sweeps the workgroup count.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repsurf_amd import mlp_hip as H

dev = torch.device("cuda")
rows = 524288
torch.manual_seed(0)
dz = torch.randn(rows, 32, device=dev); y = torch.randn(rows, 32, device=dev)
x = torch.randn(rows, 20, device=dev)                      # aligned layout: [pos 3 + pad][16 features]
p, q, r = (torch.randn(32, device=dev) for _ in range(3))
trash = torch.empty(150_000_000, device=dev)

def run(kcols, a_off, cold):
    p_op = H.operand(H.OP_AFF2, dz, 32, y, 32, s1=p, t1=r, s2=q)
    q_op = H.operand(H.OP_ID, x, 20, a_off=a_off)
    for _ in range(3):
        H.wgrad(rows, 32, kcols, p_op, q_op, dev)
    ts = []
    for _ in range(10):
        if cold:
            trash.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); H.wgrad(rows, 32, kcols, p_op, q_op, dev); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

for kcols, off in ((16, 4), (3, 0)):
    print(f"chunks={os.environ.get('REPSURF_WGRAD_CHUNKS', 'default')} kcols={kcols}: warm {run(kcols, off, False):7.1f} us, cold {run(kcols, off, True):7.1f} us  (kernel + partial reduction; 167 MB = 21 us at 8 TB/s)")
