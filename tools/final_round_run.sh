#!/bin/bash
# Round-end evidence on one GPU box: GPU test suite, smoke(), the default bench line (+ breakdown), the bf16 line,
# the in-step (non-pipelined) form, and rocprofv3 kernel-trace stats of the fp32 and bf16 runs.  Usage: tools/final_round_run.sh r01_k
TAG=${1:-r01_k}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_$TAG
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -q -m gpu ) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
timeout 600 python bench.py --breakdown $O/breakdown.json > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-pipeline > $O/bench_no_pipeline.json 2> $O/bench_np.err
timeout 300 python bench.py --no-cpu-baseline --dtype bf16 --breakdown $O/breakdown_bf16.json > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 300 python bench.py --no-cpu-baseline --dtype bf16 --batch 64 --points 2048 > $O/bench_bf16_b64x2048.json 2> $O/bench_bf16_c4.err
timeout 300 python tools/sharded_time.py 2>&1 | grep "ms/step" > $O/sharded_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bf16_graph -- python $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 3 --dtype bf16 > $O/bf16_graph.log 2>&1
cd $R
tail -n 4 $O/pytest_gpu.log; tail -n 2 $O/smoke.log; cut -c1-260 $O/bench.json; cut -c1-200 $O/bench_bf16.json; cat $O/sharded_time.txt
