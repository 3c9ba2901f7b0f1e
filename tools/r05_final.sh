#!/bin/bash
# final-tree evidence (GPU box): the GPU suite five times with -x, smoke, the default bench line, the rocprofv3 kernel trace of the replayed step.
# -> gpurun_out/r05ev (suite / smoke), gpurun_out/r05fin (bench, trace).      gpurun --timeout 1500 -- 'bash tools/r05_final.sh'
cd $GRAFT_REPO_ROOT
bash tools/r05_evidence.sh tests
O=$GRAFT_REPO_ROOT/gpurun_out/r05fin; mkdir -p $O
timeout 600 python bench.py > $O/bench_cls.json 2> $O/bench_cls.err; echo "cls rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_cls.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], d["fp32_mfma_ms_per_step"], r["frac"], r["avg_launch_us"], r["all_mfma_launches"]["frac"], d["roofline_ballquery"]["clouds_per_launch"]["2048"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o graph -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-alt-arithmetic --steps 20 --warmup 3 --no-kernel-timing > $O/graph.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python tools/kernel_stats_by_grid.py $O/graph_kernel_trace.csv > $O/graph_kernel_stats_by_grid.csv 2>/dev/null
rm -f $O/graph_kernel_trace.csv $O/*agent_info.csv
head -12 $O/graph_kernel_stats.csv | cut -c1-150
