#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04bd; mkdir -p $O
cd $R
b() { python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
timeout 100 python tools/gemm_split_ab.py 2>&1 | grep "us " | head -7 > $O/acc2.txt
REPSURF_HIP_LIB=$R/build_exp/librepsurf_noacc2.so timeout 100 python tools/gemm_split_ab.py 2>&1 | grep "us " | head -7 > $O/noacc2.txt
paste -d'\n' $O/noacc2.txt $O/acc2.txt | cut -c1-150
echo "cls one acc   $(REPSURF_HIP_LIB=$R/build_exp/librepsurf_noacc2.so b)"
echo "cls two acc   $(b)"
echo "cls one acc   $(REPSURF_HIP_LIB=$R/build_exp/librepsurf_noacc2.so b)"
echo "cls two acc   $(b)"
