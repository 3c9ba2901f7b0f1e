#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o knn -- python $R/tools/knn_grid_bench.py > $O/run.txt 2>&1; echo "rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $f
python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "grid_" in n or "knn_packed" in n:
        print(f"{int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}  {n[:90]}")
P
