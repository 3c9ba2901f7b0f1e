#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04bk; mkdir -p $O
cd $R
RS_GEMM_WS=1 timeout 200 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu -k "not fp32_mfma_instances" 2>&1 | tail -1
RS_GEMM_WS=1 timeout 100 python tools/gemm_split_ab.py 2>&1 | grep "us " | head -8 > $O/ws.txt; cut -c1-130 $O/ws.txt
b() { python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
echo "cls default   $(b)"
echo "cls WS=1      $(RS_GEMM_WS=1 b)"
echo "cls default   $(b)"
echo "cls WS=1      $(RS_GEMM_WS=1 b)"
echo "seg default   $(b --workload seg)"
echo "seg WS=1      $(RS_GEMM_WS=1 b --workload seg)"
