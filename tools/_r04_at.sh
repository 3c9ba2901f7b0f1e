#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04at; mkdir -p $O
one() { local tag=$1; shift; timeout 900 python bench.py "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
one seg_full --workload seg
one seg_no_cpu --workload seg --no-cpu-baseline
one seg_no_timing --workload seg --no-cpu-baseline --no-kernel-timing
REPSURF_GATHER_BACKWARD=0 one seg_full_scatter --workload seg --no-cpu-baseline
REPSURF_LAZY_ROWS=0 one seg_full_nolazy --workload seg --no-cpu-baseline
