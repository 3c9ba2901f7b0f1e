#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04v; mkdir -p $O
for f in 0.125,1.0; do echo "fill $f"; REPSURF_KNN_GRID_FILL=$f timeout 300 python tools/knn_grid_bench.py 2>&1 | grep -v amdgpu.ids | grep "k=32" | head -2; done | tee $O/fill_sweep.txt
timeout 900 python -m pytest tests/test_seg_gpu.py -q -m gpu -x --timeout 600 -k "knn" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -2
