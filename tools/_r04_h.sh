#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
timeout 300 python tools/seg_torch_op_sources.py > $O/seg_ops.txt 2>&1; echo "spy rc=$?"; sed -n '/==== by source/,$p' $O/seg_ops.txt | head -70
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cls.json 2> $O/bench_cls.err; echo "bench rc=$?"; tail -2 $O/bench_cls.err
python - <<PY
import json
d=json.loads(open("$O/bench_cls.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r.get("eager_avg_launch_us"), r["dims"])
print(json.dumps(r["all_mfma_launches"])[:700])
PY
export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/$O/trace; mkdir -p $D
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o graph -- python $GRAFT_REPO_ROOT/bench.py --workload seg --steps 10 --warmup 3 --no-kernel-timing --no-cpu-baseline > $D/run.log 2>&1)
f=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/kernel_stats_by_grid.py $f > $O/stats_seg.csv; rm -rf $D
head -40 $O/stats_seg.csv | cut -c1-130
