#!/usr/bin/env python3
"""Ball query against its HBM roofline (algorithmic bytes 4*(3BN + 3BS + BS*nsample) / time / 8 TB/s) over batch sizes.
RS_BALLQUERY_GRID=0 selects the brute-force scan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from repsurf_amd import ops

dev = torch.device("cuda")
SHAPES = [(1024, 512, 0.2, 32), (1024, 32, 0.2, 32), (1024, 512, 0.05, 32), (512, 128, 0.4, 64)]
if os.environ.get("RS_BQ_ONLY"):            # one shape, 2 048 clouds only (PMC passes: a counter average must not mix shapes)
    SHAPES = [SHAPES[int(os.environ["RS_BQ_ONLY"])]]
for (n, m, r, ns) in SHAPES:
    for b in ((2048,) if os.environ.get("RS_BQ_ONLY") else (32, 2048)):
        g = torch.Generator().manual_seed(b)
        xyz = (torch.rand(b, n, 3, generator=g) * 2 - 1).to(dev)
        centres = xyz[:, torch.randperm(n, generator=g)[:m]].contiguous()
        for _ in range(3):
            ops.ballquery(r, ns, xyz, centres, return_count=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        e0.record()
        for _ in range(iters):
            idx, cnt = ops.ballquery(r, ns, xyz, centres, return_count=True)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        byt = 4.0 * (3 * b * n + 3 * b * m + b * m * ns)
        print(f"grid={os.environ.get('RS_BALLQUERY_GRID', '1')} B={b:5d} N={n} S={m} r={r} ns={ns}: {us:8.1f} us  "
              f"{byt / us / 1e3:8.1f} GB/s  = {byt / us / 1e3 / 8000:6.3f} of 8 TB/s   (mean distinct {cnt.float().mean().item():.2f})", flush=True)
