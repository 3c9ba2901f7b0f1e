#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'P'
import torch, sys
sys.path.insert(0, '.')
from repsurf_amd import ops
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return 1000 * e0.elapsed_time(e1) / reps
for b, n in ((32, 1024), (64, 2048), (16, 4096), (8, 8192)):
    x = torch.rand(b, n, 3, device='cuda') * 2 - 1
    r = {}
    for g in (True, False):
        ops.UMBRELLA_GRID = g; ops.UMBRELLA_GRID_MIN_ROWS = 0
        r[g] = t(lambda: ops.umbrella_features(x, 9))
    print(f"umbrella_features {b} x {n}, k = 9: grid {r[True]:.1f} us  scan {r[False]:.1f} us")
P
