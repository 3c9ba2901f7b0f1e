#!/bin/bash
# PMC counters of rs_ballquery at 2 048 clouds, one counter set per run (rocprofv3 --pmc alone), for the round-5 kernel (RS_BALLQUERY_CELLS=2)
# and the round-3 kernel (=1) on the same box.  Output: gpurun_out/r05pmc/summary.txt
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export RS_BALLQUERY_GRID=1 RS_BQ_ONLY=0
for CELLS in 2 1; do
  for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    N=$(echo $C | tr ' ' '_')
    RS_BALLQUERY_CELLS=$CELLS timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/cells$CELLS -o $N -- python $GRAFT_REPO_ROOT/tools/ballquery_bench.py > $O/log_${CELLS}_$N.txt 2>&1; echo "pmc cells=$CELLS $N rc=$?"
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r05pmc/summary.txt
import csv, glob, collections
for cells in (2, 1):
    print("RS_BALLQUERY_CELLS=%d" % cells)
    for f in sorted(glob.glob("gpurun_out/r05pmc/cells%d/**/*counter_collection.csv" % cells, recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "ballquery_cells" not in k: continue
            acc[(k.split("(")[0][-40:], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            for c, x in v.items(): print("  %-24s %-42s grid %8s %14.0f (%d launches)" % (c, k[0], k[1], sum(x)/len(x), len(x)))
PY
