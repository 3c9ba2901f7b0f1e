#!/bin/bash
# Round 6: unit-4 experiment builds (tools/build_exp_split.sh <names>) against the product build and the round-5 tree, one box:
# the classification / segmentation step and the dominant GEMM classes stand-alone.
#   gpurun -- 'bash tools/r06_variants.sh tag NAME1 NAME2 ...'
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
step() { # tree lib tag args
  local tree=$1 lib=$2 tag=$3; shift 3
  ( cd $tree && REPSURF_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --no-alt-arithmetic "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])" )
}
{
for r in 1 2; do
  step build_exp/head_tree "" head_cls --steps 60 --warmup 10
  step . "" product_cls --steps 60 --warmup 10
  for v in "$@"; do step . $PWD/build_exp/librepsurf_$v.so ${v}_cls --steps 60 --warmup 10; done
done
step build_exp/head_tree "" head_seg --workload seg --steps 30 --warmup 5
step . "" product_seg --workload seg --steps 30 --warmup 5
for v in "$@"; do step . $PWD/build_exp/librepsurf_$v.so ${v}_seg --workload seg --steps 30 --warmup 5; done
} 2>&1 | tee $O/steps.txt
gb() { ( cd $1 && REPSURF_HIP_LIB=$2 timeout 300 python tools/gemm_bench.py one $3 $4 $5 $6 2>&1 | grep "^gemm" ); }
{
for shape in "4096 1024 512" "4096 512 1024" "4096 256 512" "48234 128 256" "48234 256 128" "66584 64 128" "66584 128 64"; do
  echo "== $shape"
  echo -n "head      "; gb build_exp/head_tree "" $shape
  echo -n "product   "; gb . "" $shape
  for v in "$@"; do echo -n "$v "; gb . $PWD/build_exp/librepsurf_$v.so $shape; done
done
} 2>&1 | tee $O/gemm.txt
