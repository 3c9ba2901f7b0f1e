#!/bin/bash
# one-off: bf16 parity tests + fp32 / bf16 bench lines on the same box
mkdir -p gpurun_out/c1
O=gpurun_out/c1
timeout 400 python -m pytest tests/test_mlp_gpu.py -q -x -k "bf16" > $O/pytest_bf16.log 2>&1; echo "rc=$?" >> $O/pytest_bf16.log
timeout 200 python bench.py --steps 30 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err
timeout 200 python bench.py --steps 30 --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 200 python bench.py --steps 20 --no-cpu-baseline --batch 64 --points 2048 > $O/bench_fp32_c4.json 2> $O/bench_fp32_c4.err
timeout 200 python bench.py --steps 20 --no-cpu-baseline --batch 64 --points 2048 --dtype bf16 > $O/bench_bf16_c4.json 2> $O/bench_bf16_c4.err
tail -5 $O/pytest_bf16.log
cat $O/bench_*.json | cut -c1-400
tail -3 $O/*.err
