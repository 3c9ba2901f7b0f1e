#!/bin/bash
# A/B on one box: the bf16-storage GEMM unit (RS_MLP_TU=3) with / without the SLP vectorizer -- bf16 bench lines of both libraries, interleaved
for rep in 1 2; do
  for l in product lib_sb_noslp; do
    if [ $l = product ]; then unset REPSURF_HIP_LIB; else export REPSURF_HIP_LIB=$PWD/build_exp/$l.so; fi
    python bench.py --no-cpu-baseline --dtype bf16 --batch 64 --points 2048 --no-extra-legs --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l bf16 b64x2048', d['ms_per_step'], d['value'])"
    python bench.py --no-cpu-baseline --dtype bf16 --no-extra-legs --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l bf16 b32x1024', d['ms_per_step'], d['value'])"
  done
done
