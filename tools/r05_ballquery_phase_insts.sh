#!/bin/bash
# VALU / SALU / LDS instructions per launch of the round-5 ball-query kernel up to each phase boundary (RS_BALLQUERY_DBG early exits:
# 1 = build, 2 = + walk, 3 = + decode / sort, 0 = whole kernel), 2 048 clouds, the flagship shape.  Output: gpurun_out/r05pmc2/summary.txt
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export RS_BALLQUERY_GRID=1 RS_BQ_ONLY=0
for D in 1 2 3 0; do
  RS_BALLQUERY_DBG=$D timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/dbg$D -o c -- python $GRAFT_REPO_ROOT/tools/ballquery_bench.py > $O/log_$D.txt 2>&1; echo "dbg=$D rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r05pmc2/summary.txt
import csv, glob, collections
for d in (1, 2, 3, 0):
    for f in sorted(glob.glob("gpurun_out/r05pmc2/dbg%d/**/*counter_collection.csv" % d, recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "ballquery_cells" in r["Kernel_Name"] and r["Grid_Size"] == "1048576":
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("dbg=%d " % d + "  ".join("%s %.0f (per wave %.0f)" % (c, sum(x) / len(x), sum(x) / len(x) / 16384) for c, x in sorted(acc.items())))
PY
