#!/bin/bash
mkdir -p gpurun_out/c5
O=gpurun_out/c5
for v in 0 1; do
  REPSURF_DEFER_REDUCE=$v timeout 200 python bench.py --steps 40 --no-cpu-baseline --no-kernel-timing > $O/bench_dr_$v.json 2> $O/bench_dr_$v.err
done
REPSURF_DEFER_REDUCE=1 timeout 200 python bench.py --steps 40 --no-cpu-baseline --no-kernel-timing --dtype bf16 > $O/bench_dr_bf16_1.json 2> $O/bench_dr_bf16_1.err
( time timeout 1200 python -m pytest tests -q -x -m gpu ) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
for f in $O/bench_*.json; do echo $f; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $f | tr '\n' ' '; echo; done
tail -n 8 $O/pytest_gpu.log
