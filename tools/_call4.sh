#!/bin/bash
mkdir -p gpurun_out/c4
O=gpurun_out/c4
for v in 0 1; do
  REPSURF_WGRAD_STREAM=$v timeout 200 python bench.py --steps 40 --no-cpu-baseline --no-kernel-timing > $O/bench_ws_$v.json 2> $O/bench_ws_$v.err
  REPSURF_WGRAD_STREAM=$v timeout 200 python bench.py --steps 40 --no-cpu-baseline --no-kernel-timing --dtype bf16 > $O/bench_ws_bf16_$v.json 2> $O/bench_ws_bf16_$v.err
done
for f in $O/bench_*.json; do echo $f; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $f | tr '\n' ' '; echo; done
tail -2 $O/*.err
