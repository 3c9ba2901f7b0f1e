#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for s in 512 480 448 384; do
  RS_GEMM_SLOTS=$s RS_GEMM_SLOTS64=$s one seg_slots_$s --workload seg --steps 20 --warmup 5
  RS_GEMM_SLOTS=$s RS_GEMM_SLOTS64=$s REPSURF_PIPE_SKIP_GEO=1 one seg_alone_slots_$s --workload seg --steps 20 --warmup 5
  RS_GEMM_SLOTS=$s RS_GEMM_SLOTS64=$s one cls_slots_$s --steps 40 --warmup 10
  RS_GEMM_SLOTS=$s RS_GEMM_SLOTS64=$s REPSURF_PIPE_SKIP_GEO=1 one cls_alone_slots_$s --steps 40 --warmup 10
done | tee $O/ab.txt
