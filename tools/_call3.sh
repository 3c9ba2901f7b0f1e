#!/bin/bash
mkdir -p gpurun_out/c3
O=gpurun_out/c3
for v in 0 1 2; do
  RS_GEMM_BM32=$v timeout 200 python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing > $O/bench_bm32_$v.json 2> $O/bench_bm32_$v.err
done
RS_GEMM_BM32=1 RS_GEMM_SLOTS32=1024 timeout 200 python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing > $O/bench_bm32_1_s1024.json 2> $O/err
RS_GEMM_BM32=1 RS_GEMM_SLOTS32=512 timeout 200 python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing > $O/bench_bm32_1_s512.json 2> $O/err
RS_GEMM_BM32=2 timeout 400 python -m pytest tests/test_mlp_gpu.py tests/test_model_gpu.py -q -x > $O/pytest.log 2>&1
for f in $O/bench_*.json; do echo $f; cut -c1-200 $f | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done
tail -3 $O/pytest.log
