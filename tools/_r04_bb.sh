#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04bb; mkdir -p $O
cd $R
timeout 200 python tools/gemm_split_ab.py > $O/ab_fp32.txt 2>&1; echo "rc=$?"
RS_GEMM_SPLIT3=1 timeout 200 python tools/gemm_split_ab.py > $O/ab_split.txt 2>&1; echo "rc=$?"
cat $O/ab_fp32.txt $O/ab_split.txt | grep -v "^$" | tail -40
RS_GEMM_SPLIT3=1 timeout 300 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu > $O/test_mlp_split.txt 2>&1; echo "rc=$?"; tail -5 $O/test_mlp_split.txt
for i in 1; do
timeout 120 python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fp32 cls', d['ms_per_step'], d['value'])"
RS_GEMM_SPLIT3=1 timeout 120 python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('split cls', d['ms_per_step'], d['value'])"
done
