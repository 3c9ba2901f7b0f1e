#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ap; mkdir -p $O
true
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2 3 4; do
one cls_gather --steps 40 --warmup 10
REPSURF_COMPACT_CSR=0 one cls_scatter --steps 40 --warmup 10
done | tee $O/ab.txt
tail -3 $O/err_cls_gather.txt
