#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04bc; mkdir -p $O
cd $R
RS_GEMM_SPLIT3=1 RS_WGRAD_SPLIT3_WIDE=1 timeout 200 python tools/gemm_split_ab.py 2>&1 | grep "us " > $O/ab_split_wide.txt; echo "rc=$?"
cat $O/ab_split_wide.txt | head -8
b() { python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
echo "cls fp32        $(b)"
echo "cls split       $(RS_GEMM_SPLIT3=1 b)"
echo "cls split+wide  $(RS_GEMM_SPLIT3=1 RS_WGRAD_SPLIT3_WIDE=1 b)"
echo "cls fp32        $(b)"
echo "cls split       $(RS_GEMM_SPLIT3=1 b)"
echo "seg fp32        $(b --workload seg)"
echo "seg split       $(RS_GEMM_SPLIT3=1 b --workload seg)"
echo "seg split+wide  $(RS_GEMM_SPLIT3=1 RS_WGRAD_SPLIT3_WIDE=1 b --workload seg)"
RS_GEMM_SPLIT3=1 timeout 400 python -m pytest tests -x -q -m gpu > $O/gpu_tests_split.log 2>&1; echo "rc=$?"; tail -5 $O/gpu_tests_split.log
