#!/bin/bash
# Round 6 A/B on ONE box: build_exp/head_tree (a built copy of the round-5 final commit, see tools/ab_vs_head.sh) against the working tree,
# interleaved bench lines; then the dominant GEMM classes stand-alone (tools/gemm_bench.py) in both trees.
#   gpurun -- 'bash tools/r06_ab.sh [tag]'
cd $GRAFT_REPO_ROOT
TAG=${1:-ab}
O=gpurun_out/$TAG; mkdir -p $O
one() { # tree tag args...
  local tree=$1 tag=$2; shift 2
  ( cd $tree && timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --no-alt-arithmetic "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])" )
}
for r in 1 2 3; do
  one build_exp/head_tree head_cls --steps 60 --warmup 10
  one . new_cls --steps 60 --warmup 10
  one build_exp/head_tree head_seg --workload seg --steps 30 --warmup 5
  one . new_seg --workload seg --steps 30 --warmup 5
done | tee $O/ab.txt
for tree in build_exp/head_tree .; do
  echo "== $tree"; ( cd $tree && timeout 600 python tools/gemm_bench.py 2>&1 | tail -40 )
done > $O/gemm_bench.txt 2>&1
