#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for s in 512 768 1024 2048; do
  export REPSURF_PARTIAL_BLOCKS=$s RS_GEMM_SLOTS=$s RS_GEMM_SLOTS64=$s
  one seg_slots_$s --workload seg --steps 20 --warmup 5
  one cls_slots_$s --steps 40 --warmup 10
done | tee $O/ab.txt
export REPSURF_PARTIAL_BLOCKS=1024 RS_GEMM_SLOTS=1024 RS_GEMM_SLOTS64=1024 REPSURF_WGRAD_CHUNKS=1024
one seg_slots_wg_1024 --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
one cls_slots_wg_1024 --steps 40 --warmup 10 | tee -a $O/ab.txt
