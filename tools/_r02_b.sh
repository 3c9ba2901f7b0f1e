#!/bin/bash
# round-2 GPU session B: new epilogue correctness, A/B micro-benchmark and step time, graph-event probe
cd /root/repo
O=gpurun_out/r02_b; mkdir -p $O
timeout 600 python -m pytest tests/test_mlp_gpu.py tests/test_model_gpu.py tests/test_seg_gpu.py tests/test_graph_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 120 python tools/probe_graph_events.py > $O/graph_events.txt 2>&1; tail -6 $O/graph_events.txt
for shape in "48171 128 128" "66754 64 128" "48171 128 256" "48171 256 128" "262144 128 128"; do
  echo "== new $shape"; timeout 120 python tools/gemm_bench.py one $shape 2>&1 | tail -1
  echo "== old $shape"; REPSURF_HIP_LIB=build_exp/librepsurf_LDSEPI.so timeout 120 python tools/gemm_bench.py one $shape 2>&1 | tail -1
done > $O/gemm_ab.txt 2>&1
cat $O/gemm_ab.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_new_$i.json 2>$O/bench_new_$i.err; python -c "import json;d=json.load(open('$O/bench_new_$i.json'));print('new',d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_us'])"
REPSURF_HIP_LIB=build_exp/librepsurf_LDSEPI.so timeout 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_old_$i.json 2>$O/bench_old_$i.err; python -c "import json;d=json.load(open('$O/bench_old_$i.json'));print('old',d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_us'])"
done
RS_GEMM_SLOTS64=768 REPSURF_PARTIAL_BLOCKS=768 timeout 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_new_768.json 2>$O/bench_768.err; python -c "import json;d=json.load(open('$O/bench_new_768.json'));print('new768',d['ms_per_step'],d['roofline']['frac'])"
