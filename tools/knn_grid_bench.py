#!/usr/bin/env python3
"""Packed-batch kNN of the segmentation geometry stage (BASELINE configs[3]: 16 clouds x 4096 points): the scan of the whole cloud per
query (rs_knnquery_offset) against the per-cloud uniform grids (rs_knn_grid_build + rs_knn_grid_query), HIP events, 20 launches each.
GPU box: python tools/knn_grid_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from repsurf_amd import ops

dev = torch.device("cuda")
r = np.random.RandomState(0)
B, N = 16, 4096
xyz = torch.from_numpy((r.rand(B * N, 3) * 2 - 1).astype(np.float32)).to(dev)


def level(x, n, stride):
    return x.reshape(B, n, 3)[:, ::stride].reshape(-1, 3).contiguous(), ops.offsets_tensor([(i + 1) * (n // stride) for i in range(B)], dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


off0 = ops.offsets_tensor([(i + 1) * N for i in range(B)], dev)
cases = [("umbrella fans: k=9, 65 536 self queries over 4096", 9, xyz, off0, xyz, off0)]
src, soff, n = xyz, off0, N
for lvl in range(4):
    q, qoff = level(src, n, 4)
    cases.append((f"grouping {lvl + 1}: k=32, {q.shape[0]} centres over {n}", 32, src, soff, q, qoff))
    cases.append((f"interpolation {lvl + 1}: k=3, {src.shape[0]} fine rows over {n // 4}", 3, q, qoff, src, soff))
    src, soff, n = q, qoff, n // 4
tot = [0.0, 0.0]
for name, k, s, so, q, qo in cases:
    a = ops.knnquery_offset(k, s, q, so, qo, grid=True)
    b = ops.knnquery_offset(k, s, q, so, qo, grid=False)
    same = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    tg = timed(lambda: ops.knnquery_offset(k, s, q, so, qo, grid=True))
    ts = timed(lambda: ops.knnquery_offset(k, s, q, so, qo, grid=False))
    tot[0] += tg; tot[1] += ts
    print(f"{name:62s} scan {ts:8.1f} us   grid (build + query) {tg:8.1f} us   identical: {same}")
print(f"{'sum':62s} scan {tot[1]:8.1f} us   grid {tot[0]:8.1f} us")
