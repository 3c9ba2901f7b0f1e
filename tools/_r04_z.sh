#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
one cls_default --steps 40 --warmup 10
REPSURF_HIP_LIB=build_exp/librepsurf_fakeumb.so one cls_fake_umb_knn --steps 40 --warmup 10
done | tee $O/ab.txt
