#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04be; mkdir -p $O
cd $R
b() { python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
RS_WGRAD_SPLIT3_WIDE=2 timeout 100 python tools/gemm_split_ab.py 2>&1 | grep "rows=" | head -8 > $O/wide2.txt; cut -c1-240 $O/wide2.txt
RS_WGRAD_SPLIT3_WIDE=2 timeout 200 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu -k "not fp32_mfma_instances" 2>&1 | tail -1
echo "cls default   $(b)"
echo "cls wide=2    $(RS_WGRAD_SPLIT3_WIDE=2 b)"
echo "cls default   $(b)"
echo "cls wide=2    $(RS_WGRAD_SPLIT3_WIDE=2 b)"
echo "seg default   $(b --workload seg)"
echo "seg wide=2    $(RS_WGRAD_SPLIT3_WIDE=2 b --workload seg)"
