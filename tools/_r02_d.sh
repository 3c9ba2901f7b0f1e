#!/bin/bash
cd /root/repo
O=gpurun_out/r02_d; mkdir -p $O
timeout 900 python -m pytest tests/test_seg_gpu.py tests/test_geometry_gpu.py tests/test_parity_full_gpu.py tests/test_dropin.py tests/test_graph_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 300 python tools/fps_sweep.py > $O/fps_sweep.txt 2>&1; cat $O/fps_sweep.txt
for v in product NO_EARLY_PREFETCH EARLY_PREFETCH_ALL; do
  if [ $v = product ]; then L=""; else L="build_exp/librepsurf_$v.so"; fi
  for i in 1 2; do
    REPSURF_HIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 > $O/bench_${v}_$i.json 2>$O/bench_${v}.err; python -c "import json;d=json.load(open('$O/bench_${v}_$i.json'));print('$v',d['ms_per_step'])"
  done
done
RS_GEMM_SLOTS64_FWD=768 REPSURF_PARTIAL_BLOCKS=768 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 > $O/bench_fwd768.json 2>$O/bench_fwd768.err; python -c "import json;d=json.load(open('$O/bench_fwd768.json'));print('fwd768',d['ms_per_step'])"
timeout 600 python bench.py --workload seg --steps 20 > $O/bench_seg.json 2>$O/bench_seg.err; tail -c 2500 $O/bench_seg.json; tail -5 $O/bench_seg.err
