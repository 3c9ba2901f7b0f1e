#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04bi; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 100 python tools/gemm_split_ab.py 2>&1 | grep "rows=" > $O/presplit.txt; cut -c1-200 $O/presplit.txt | head -14
b() { python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
echo "cls fp32      $(RS_GEMM_SPLIT3=0 b)"
echo "cls default   $(b)"
echo "cls fp32      $(RS_GEMM_SPLIT3=0 b)"
echo "cls default   $(b)"
echo "seg fp32      $(RS_GEMM_SPLIT3=0 b --workload seg)"
echo "seg default   $(b --workload seg)"
