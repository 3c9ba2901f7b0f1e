#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04as; mkdir -p $O
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
one seg_gather_both --workload seg --steps 20 --warmup 5
REPSURF_INTERP_GATHER=0 one seg_gather_grouping_only --workload seg --steps 20 --warmup 5
done | tee $O/ab.txt
