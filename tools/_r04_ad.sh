#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ad; mkdir -p $O
python -c "import torch; print(torch.cuda.Stream.priority_range())"
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
one cls_default --steps 40 --warmup 10
REPSURF_PIPE_PRIORITY=1 one cls_priority --steps 40 --warmup 10
done | tee $O/ab.txt
for r in 1 2; do
one seg_default --workload seg --steps 20 --warmup 5
REPSURF_PIPE_PRIORITY=1 one seg_priority --workload seg --steps 20 --warmup 5
done | tee -a $O/ab.txt
