#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; mkdir -p $O
timeout 300 python tools/knn_grid_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/knn_grid_bench.txt
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
one seg_grid_default --workload seg --steps 20 --warmup 5
REPSURF_KNN_GRID=0 one seg_scan --workload seg --steps 20 --warmup 5
REPSURF_KNN_GRID_MIN_ROWS=100000000,512,128 one seg_k3scan --workload seg --steps 20 --warmup 5
done | tee $O/ab.txt
timeout 1500 python -m pytest tests/test_seg_gpu.py tests/test_parity_full_gpu.py -q -m gpu -x --timeout 900 > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/tests.log | head
