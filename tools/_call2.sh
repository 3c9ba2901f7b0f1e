#!/bin/bash
mkdir -p gpurun_out/c2
O=gpurun_out/c2
timeout 400 python -m pytest tests/test_mlp_gpu.py -q -x -k "bf16" > $O/pytest_bf16.log 2>&1; echo "rc=$?" >> $O/pytest_bf16.log
timeout 200 python bench.py --steps 30 --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 200 python bench.py --steps 20 --no-cpu-baseline --batch 64 --points 2048 --dtype bf16 > $O/bench_bf16_c4.json 2> $O/bench_bf16_c4.err
( time timeout 1200 python -m pytest tests -q -x -m gpu ) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_bf16.log
cat $O/bench_*.json | cut -c1-300
tail -8 $O/pytest_gpu.log
