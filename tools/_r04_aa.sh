#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04aa; mkdir -p $O
timeout 900 python -m pytest tests/test_geometry_gpu.py -q -m gpu -x --timeout 600 -k "umbrella or fixtures" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR|Error|assert" $O/tests.log | head
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
one cls_umb_grid --steps 40 --warmup 10
REPSURF_UMBRELLA_GRID=0 one cls_umb_scan --steps 40 --warmup 10
done | tee $O/ab.txt
python - <<'P'
import torch, numpy as np, sys
sys.path.insert(0, '.')
from repsurf_amd import ops
x = torch.rand(32, 1024, 3, device='cuda') * 2 - 1
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return 1000 * e0.elapsed_time(e1) / reps
for g in (True, False):
    ops.UMBRELLA_GRID = g
    print("umbrella_features 32 x 1024, k = 9:", "grid" if g else "scan", f"{t(lambda: ops.umbrella_features(x, 9)):.1f} us")
P
