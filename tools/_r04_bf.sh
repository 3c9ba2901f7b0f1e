#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04bf; mkdir -p $O
cd $R
b() { python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
echo "cls default              $(b)"
echo "cls BN64_BELOW64=256     $(RS_GEMM_BN64_BELOW64=256 b)"
echo "cls BN64_BELOW64=0       $(RS_GEMM_BN64_BELOW64=0 b)"
echo "cls BN64_BELOW64=1024    $(RS_GEMM_BN64_BELOW64=1024 b)"
echo "cls BM64=0               $(RS_GEMM_BM64=0 b)"
echo "cls default              $(b)"
echo "seg default              $(b --workload seg)"
echo "seg BN64_BELOW64=0       $(RS_GEMM_BN64_BELOW64=0 b --workload seg)"
echo "seg BM64=0               $(RS_GEMM_BM64=0 b --workload seg)"
