#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ax; mkdir -p $O
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for w in 0 2 8 16; do
RS_FPS_WAVES=$w one seg_fps_waves_$w --workload seg --steps 20 --warmup 5
done | tee $O/ab.txt
REPSURF_PIPE_SKIP_GEO=1 one seg_network_alone --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
REPSURF_PIPE_SKIP_GEO=1 one cls_network_alone --steps 40 --warmup 10 | tee -a $O/ab.txt
one cls_default --steps 40 --warmup 10 | tee -a $O/ab.txt
