#!/usr/bin/env python3
"""FPS latency over workgroup shapes (RS_FPS_WAVES is read per call): the classification stages (32 x 1024 -> 512,
32 x 512 -> 128, sample() 32 x 2048 -> 1024) and the packed segmentation stage (16 x 4096 -> 1024)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repsurf_amd import ops
dev = torch.device("cuda")


def t(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


xyz = (torch.rand(16 * 4096, 3, device=dev) * 2 - 1).contiguous()
off = ops.offsets_tensor([4096 * (i + 1) for i in range(16)], dev)
noff = ops.strided_offset(off, 4)
dense = {(n, m): (torch.rand(32, n, 3, device=dev) * 2 - 1).contiguous() for n, m in ((1024, 512), (512, 128), (2048, 1024))}
dense[(4096, 1024)] = xyz.view(16, 4096, 3)          # the packed case's clouds through the dense (non-tie-rule) kernel
start = torch.zeros(32, dtype=torch.int32, device=dev)[:]
for w in (1, 2, 4, 8, 16):
    os.environ["RS_FPS_WAVES"] = str(w)
    line = f"waves {w:2d}:"
    for (n, m), x in dense.items():
        us = t(lambda: ops.furthestsampling(x, m, start[:x.shape[0]]))
        line += f"  {x.shape[0]}x{n}->{m}: {us:7.1f} us ({us / (m - 1) * 1e3:4.0f} ns/pick)"
    us = t(lambda: ops.furthestsampling_offset(xyz, off, noff))
    line += f"  packed 16x4096->1024: {us:7.1f} us ({us / 1023 * 1e3:4.0f} ns/pick)"
    print(line, flush=True)
