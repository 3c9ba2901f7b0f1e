#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for cfg in "100000000,2048,2048" "100000000,2048,100000000" "512,512,2048" "100000000,100000000,100000000" "0,0,0"; do
  REPSURF_KNN_GRID_MIN_ROWS=$cfg one seg_min_$cfg --workload seg --steps 20 --warmup 5
done | tee $O/ab.txt
REPSURF_KNN_GRID_MIN_ROWS=100000000,2048,2048 REPSURF_KNN_GRID_FILL=0.083 one seg_fill083 --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
