#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the SAME bench.py command the driver runs (graph replay) and of
# the eager step, PMC passes (each counter set in its own run, never combined with a trace option), then the HBM-traffic
# table bench.py's roofline.traffic reads and the MFMA-busy summary.      Usage (GPU box): tools/gpu_profile.sh r02 [seg]
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
WL=${2:-cls}
D=$R/gpurun_out/prof_${TAG}_$WL
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --workload $WL"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o graph -- $BENCH --steps 20 --warmup 3 --no-kernel-timing > $D/graph.log 2>&1; echo "graph trace rc=$?" >> $D/graph.log
if [ "$WL" = "cls" ]; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o eager -- $BENCH --steps 5 --warmup 2 --no-kernel-timing --no-graph > $D/eager.log 2>&1; echo "eager trace rc=$?" >> $D/eager.log
  for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES" "TCC_EA_RDREQ TCC_EA_RDREQ_32B"; do
    N=$(echo $C | tr ' ' '_')
    timeout 900 rocprofv3 --pmc $C --output-format csv -d $D -o pmc_$N -- $BENCH --steps 2 --warmup 1 --no-graph --launch-log $D/launch_$N.json > $D/pmc_$N.log 2>&1; echo "pmc $N rc=$?" >> $D/pmc_$N.log
  done
  cd $R
  python tools/traffic_from_pmc.py --fetch-log $D/launch_FETCH_SIZE.json --fetch-csv $D/pmc_FETCH_SIZE_counter_collection.csv \
    --write-log $D/launch_WRITE_SIZE.json --write-csv $D/pmc_WRITE_SIZE_counter_collection.csv --out $D/traffic.json > $D/traffic.log 2>&1
  python tools/pmc_summary.py $D > $D/pmc_summary.log 2>&1
fi
if [ "$WL" = "seg" ]; then      # HBM traffic of the segmentation step's launch classes (distinct keys: merged into profiles/traffic.json)
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d $D -o pmc_$C -- $BENCH --steps 2 --warmup 1 --no-graph --launch-log $D/launch_$C.json > $D/pmc_$C.log 2>&1; echo "pmc $C rc=$?" >> $D/pmc_$C.log
  done
  cd $R
  python tools/traffic_from_pmc.py --fetch-log $D/launch_FETCH_SIZE.json --fetch-csv $D/pmc_FETCH_SIZE_counter_collection.csv \
    --write-log $D/launch_WRITE_SIZE.json --write-csv $D/pmc_WRITE_SIZE_counter_collection.csv --out $D/traffic.json > $D/traffic.log 2>&1
fi
cd $R
ls -la $D | head -40
tail -q -n 2 $D/*.log
# keep what is judged small: the stats CSVs, summaries, traffic table (the raw traces stay in scratch)
mkdir -p $R/gpurun_out/keep_${TAG}_$WL
python tools/kernel_stats_by_grid.py $D/graph_kernel_trace.csv > $D/graph_kernel_stats_by_grid.csv 2>/dev/null    # per shape (the --stats summary mixes an instance's shapes)
cp $D/*kernel_stats.csv $D/traffic.json $D/pmc_summary.csv $D/*.log $R/gpurun_out/keep_${TAG}_$WL/ 2>/dev/null
