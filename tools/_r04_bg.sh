#!/bin/bash
# the PMC passes of tools/gpu_profile.sh (cls) alone, for the split-product default: traffic table + MFMA-busy summary
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof_r04s_cls
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --workload cls"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -o pmc_$N -- $BENCH --steps 2 --warmup 1 --no-graph --launch-log $D/launch_$N.json > $D/pmc_$N.log 2>&1; echo "pmc $N rc=$?"
done
cd $R
python tools/traffic_from_pmc.py --fetch-log $D/launch_FETCH_SIZE.json --fetch-csv $D/pmc_FETCH_SIZE_counter_collection.csv \
  --write-log $D/launch_WRITE_SIZE.json --write-csv $D/pmc_WRITE_SIZE_counter_collection.csv --out $D/traffic.json > $D/traffic.log 2>&1; tail -2 $D/traffic.log
python tools/pmc_summary.py $D > $D/pmc_summary.log 2>&1; tail -1 $D/pmc_summary.log
mkdir -p $R/gpurun_out/keep_r04s_cls
cp $D/traffic.json $D/pmc_summary.csv $R/gpurun_out/keep_r04s_cls/
rm -rf $D
head -12 $R/gpurun_out/keep_r04s_cls/pmc_summary.csv | cut -c1-220
