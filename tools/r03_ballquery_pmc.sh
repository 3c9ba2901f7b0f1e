#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/$O/bq_pmc -o $N -- python $GRAFT_REPO_ROOT/tools/ballquery_bench.py > $GRAFT_REPO_ROOT/$O/bq_pmc_$N.log 2>&1; echo "pmc $N rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r03t/bq_pmc/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ballquery_cells" not in k: continue
        acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, x in v.items(): print("%-28s grid %8s %14.0f (%d launches)" % (c, k, sum(x)/len(x), len(x)))
PY
