#!/bin/bash
# Same-box interleaved A/B of environment switches on the working tree: gpurun -- 'bash tools/env_ab.sh tag rounds "ENV=0 ENV2=0" "ENV=1" ...'
# each quoted argument is one arm (a space-separated env assignment list, "-" = none); classification and segmentation lines per arm and round.
cd $GRAFT_REPO_ROOT
TAG=$1; ROUNDS=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O; : > $O/ab.txt
for r in $(seq $ROUNDS); do
  for arm in "$@"; do
    E=""; [ "$arm" != "-" ] && E="$arm"
    for wl in cls seg; do
      X="--steps 60 --warmup 10"; [ $wl = seg ] && X="--workload seg --steps 30 --warmup 5"
      env $E timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --no-alt-arithmetic --no-extra-legs $X 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '[$arm]', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
    done
  done
done
