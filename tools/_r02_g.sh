#!/bin/bash
cd /root/repo
O=gpurun_out/r02_g; mkdir -p $O
timeout 900 python -m pytest tests/test_geometry_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -6 $O/tests.log
echo "== new cells kernel"; timeout 300 python tools/ballquery_bench.py 2>&1 | tee $O/ballquery_new.txt
echo "== first grid kernel"; RS_BALLQUERY_CELLS=0 timeout 300 python tools/ballquery_bench.py 2>&1 | tee $O/ballquery_old.txt
timeout 300 python -m pytest tests/test_seg_gpu.py -m gpu -q -x -k "row_linear or scene_scale" > $O/tests2.log 2>&1; tail -3 $O/tests2.log
timeout 600 python bench.py --workload seg --steps 20 --no-cpu-baseline > $O/bench_seg.json 2>$O/bench_seg.err; python -c "import json;d=json.load(open('$O/bench_seg.json'));print(d['ms_per_step'],d['value'],d['roofline'])"
