#!/bin/bash
mkdir -p gpurun_out/c6
O=gpurun_out/c6
run() { # name, lib
  REPSURF_HIP_LIB=$2 timeout 200 python bench.py --steps 40 --no-cpu-baseline --no-kernel-timing > $O/bench_$1.json 2> $O/bench_$1.err
  REPSURF_HIP_LIB=$2 timeout 200 python bench.py --steps 40 --no-cpu-baseline --no-kernel-timing --dtype bf16 > $O/bench_bf16_$1.json 2> $O/bench_bf16_$1.err
}
run base ""
for v in NTSTORE NTLOAD EARLYPF ALL3; do run $v $PWD/build_exp/librepsurf_$v.so; done
run base2 ""
for f in $O/bench_*.json; do echo -n "$f  "; grep -o '"ms_per_step": [0-9.]*' $f | tr '\n' ' '; echo; done
