#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04ba; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o knn -- python tools/knn_grid_bench.py > $O/run.txt 2>&1; echo "rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/knn_grid_kernel_stats.csv; rm -rf $O/prof
grep -E "grid_|knn_packed|inverse" $O/knn_grid_kernel_stats.csv | cut -c1-200
