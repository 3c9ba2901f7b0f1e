#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in "0.125,1.0" "1.0,1.0" "0.5,1.0"; do echo "wave for all, fill $f"; RS_KNN_GRID_WAVE_MIN=2 REPSURF_KNN_GRID_FILL=$f timeout 300 python tools/knn_grid_bench.py 2>&1 | grep -v amdgpu.ids | grep "k=9\|k=3" | head -3; done
