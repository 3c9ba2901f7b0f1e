#!/usr/bin/env python3
"""Whole-scene kNN-16 (the median filter's search, segmentation/util/utils.py:235-245): uniform-grid search (ops.knn_scene)
against the tiled scan (rs_knnquery_offset) on room-like surfaces of 1e5 .. 1e6 points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from repsurf_amd import ops

dev = torch.device("cuda")
for n in (100000, 300000, 1000000):
    r = np.random.RandomState(n)
    parts = []
    for axis, val in ((2, 0.0), (2, 3.0), (0, 0.0), (0, 8.0), (1, 0.0), (1, 6.0), (2, 0.8), (2, 0.45)):
        p = r.rand(n // 8, 3) * np.array([8.0, 6.0, 3.0])
        p[:, axis] = val + 0.01 * r.randn(n // 8)
        parts.append(p)
    x = torch.from_numpy(np.concatenate(parts).astype(np.float32)).to(dev)
    off = ops.offsets_tensor([x.shape[0]], dev)
    for name, fn in (("grid", lambda: ops.knn_scene(16, x, return_stats=True)), ("scan", lambda: ops.knnquery_offset(16, x, x, off, off))):
        if name == "scan" and n > 300000:
            print(f"N={n:8d} scan: skipped (O(N^2): ~{(n / 3e5) ** 2 * 1:.0f}x the 300k time)")
            continue
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"N={n:8d} {name}: {dt * 1e3:9.1f} ms" + (f"  {out[2]}" if name == "grid" else ""), flush=True)
