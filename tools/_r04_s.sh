#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for tag in with alone; do
  if [ $tag = alone ]; then export REPSURF_PIPE_SKIP_GEO=1; else unset REPSURF_PIPE_SKIP_GEO; fi
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o seg -- python bench.py --workload seg --steps 20 --warmup 3 --no-kernel-timing --no-cpu-baseline > $O/$tag.log 2>&1; echo "$tag rc=$?"
  grep -o '"ms_per_step": [0-9.]*' $O/$tag.log | tail -1
  f=$(find $O/$tag -name "*kernel_trace.csv" | head -1)
  python tools/trace_streams.py $f > $O/streams_$tag.txt 2>&1; head -50 $O/streams_$tag.txt
  rm -rf $O/$tag
done
