#!/bin/bash
# what-if builds of the split-product row GEMM (timing only, results wrong by construction): which resource the loop is sensitive to
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04bo; mkdir -p $O
cd $R
for v in "" ONE_MFMA ONE_FRAG ONE_STORE ONE_MFMA_ONE_FRAG_ONE_STORE; do
  if [ -n "$v" ]; then export REPSURF_HIP_LIB=$R/build_exp/librepsurf_$v.so; fi
  echo "== ${v:-product build}"
  timeout 100 python tools/gemm_split_ab.py 2>&1 | grep "us " | head -7 | cut -c26-88
done > $O/whatif.txt 2>&1
cat $O/whatif.txt
