#!/bin/bash
# A/B on ONE box: a built copy of the last commit (build_exp/head_tree: `git worktree add /tmp/h HEAD && make -C /tmp/h`, copied without
# .git/build; git-ignored, travels with the gpurun snapshot) against the working tree, interleaved runs of the same bench lines.
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
one() { # tree tag args...
  local tree=$1 tag=$2; shift 2
  ( cd $tree && timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])" )
}
for r in 1 2 3; do
  one build_exp/head_tree head_fp32_b32 --steps 40 --warmup 10
  one . new_fp32_b32 --steps 40 --warmup 10
  one build_exp/head_tree head_bf16_b32 --dtype bf16 --steps 40 --warmup 10
  one . new_bf16_b32 --dtype bf16 --steps 40 --warmup 10
  REPSURF_BF16_STORE=0 one . new_bf16_b32_store0 --dtype bf16 --steps 40 --warmup 10
  one build_exp/head_tree head_bf16_b64 --dtype bf16 --batch 64 --points 2048 --steps 20 --warmup 5
  one . new_bf16_b64 --dtype bf16 --batch 64 --points 2048 --steps 20 --warmup 5
  REPSURF_BF16_STORE=0 one . new_bf16_b64_store0 --dtype bf16 --batch 64 --points 2048 --steps 20 --warmup 5
done | tee $O/ab.txt
