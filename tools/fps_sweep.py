#!/usr/bin/env python3
"""FPS latency over workgroup shapes (RS_FPS_WAVES is read per call): packed 16 x 4096 -> 1024 and dense 32 x 1024 -> 512."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from repsurf_amd import ops
dev = torch.device("cuda")
xyz = (torch.rand(16 * 4096, 3, device=dev) * 2 - 1).contiguous()
off = ops.offsets_tensor([4096 * (i + 1) for i in range(16)], dev)
noff = ops.strided_offset(off, 4)
xyz2 = (torch.rand(16 * 1024, 3, device=dev) * 2 - 1).contiguous()
off2 = ops.offsets_tensor([1024 * (i + 1) for i in range(16)], dev)
noff2 = ops.strided_offset(off2, 4)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for w in (2, 4, 8, 16):
    os.environ["RS_FPS_WAVES"] = str(w)
    a = t(lambda: ops.furthestsampling_offset(xyz, off, noff))
    b = t(lambda: ops.furthestsampling_offset(xyz2, off2, noff2))
    print(f"waves {w:2d}: 16 x 4096 -> 1024: {a:7.1f} us ({a / 1023 * 1e3:5.0f} ns/pick)   16 x 1024 -> 256: {b:7.1f} us", flush=True)
