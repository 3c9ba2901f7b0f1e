#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats + PMC passes (each counter set in its own run),
# then the HBM-traffic table bench.py's roofline.traffic reads.  Usage (GPU box): tools/gpu_profile.sh r01_e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
TAG=${1:-r01_e}
D=$R/gpurun_out/prof_$TAG
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline"
# 1. kernel trace of the graph-replayed step (what the throughput number runs) and of the eager step
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o graph -- $BENCH --steps 20 --warmup 3 --no-kernel-timing > $D/graph.log 2>&1; echo "graph trace rc=$?" >> $D/graph.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o eager -- $BENCH --steps 5 --warmup 2 --no-kernel-timing --no-graph > $D/eager.log 2>&1; echo "eager trace rc=$?" >> $D/eager.log
# 2. PMC passes, eager launches, with the ordered launch log of the same run
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_')
  timeout 900 rocprofv3 --pmc $C --output-format csv -d $D -o pmc_$N -- $BENCH --steps 2 --warmup 1 --no-graph --launch-log $D/launch_$N.json > $D/pmc_$N.log 2>&1; echo "pmc $N rc=$?" >> $D/pmc_$N.log
done
cd $R
python tools/traffic_from_pmc.py --fetch-log $D/launch_FETCH_SIZE.json --fetch-csv $D/pmc_FETCH_SIZE_counter_collection.csv \
  --write-log $D/launch_WRITE_SIZE.json --write-csv $D/pmc_WRITE_SIZE_counter_collection.csv --out $D/traffic.json > $D/traffic.log 2>&1
python tools/pmc_summary.py $D > $D/pmc_summary.log 2>&1
ls -la $D | head -40
tail -q -n 3 $D/*.log
