#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
for f in 0.25 0.45 0.8; do echo "fill $f"; REPSURF_KNN_GRID_FILL=$f timeout 300 python tools/knn_grid_bench.py 2>&1 | grep -v amdgpu.ids | head -4; done | tee $O/fill_sweep.txt
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
  one seg_grid --workload seg --steps 20 --warmup 5
  REPSURF_KNN_GRID=0 one seg_scan --workload seg --steps 20 --warmup 5
done | tee $O/ab.txt
