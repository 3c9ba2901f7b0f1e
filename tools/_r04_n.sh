#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
  REPSURF_PIPE_SKIP_GEO=1 one cls_network_alone --steps 40 --warmup 10
  one cls_with_geometry --steps 40 --warmup 10
  REPSURF_PIPE_SKIP_GEO=1 one seg_network_alone --workload seg --steps 20 --warmup 5
  one seg_with_geometry --workload seg --steps 20 --warmup 5
done | tee $O/ab.txt
