#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04af; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
one seg_fp_node --workload seg --steps 20 --warmup 5
REPSURF_FP_FRONT=0 one seg_fp_layers --workload seg --steps 20 --warmup 5
done | tee $O/ab.txt
for tag in node layers; do
  if [ $tag = layers ]; then export REPSURF_FP_FRONT=0; else unset REPSURF_FP_FRONT; fi
  REPSURF_PIPE_SKIP_GEO=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o seg -- python bench.py --workload seg --steps 20 --warmup 3 --no-kernel-timing --no-cpu-baseline > $O/$tag.log 2>&1; echo "$tag rc=$?"
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); cp $f $O/seg_${tag}_kernel_stats.csv; rm -rf $O/$tag
done
