#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u; mkdir -p $O
timeout 900 python -m pytest tests/test_seg_gpu.py -q -m gpu -x --timeout 600 -k "knn" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR|Error" $O/tests.log | head
for f in 0.125 0.25 0.45 0.7 1.0; do echo "fill $f"; REPSURF_KNN_GRID_FILL=$f timeout 300 python tools/knn_grid_bench.py 2>&1 | grep -v amdgpu.ids | grep "k=32" | head -5; done | tee $O/fill_sweep.txt
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
REPSURF_KNN_GRID_MIN_ROWS=512,512,2048 one seg_wave_l1 --workload seg --steps 20 --warmup 5 | tee $O/ab.txt
REPSURF_KNN_GRID_MIN_ROWS=512,512,0 one seg_wave_all --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
REPSURF_KNN_GRID_MIN_ROWS=512,512,512 one seg_wave_512 --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
REPSURF_KNN_GRID_MIN_ROWS=512,512,100000000 one seg_scan32 --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
