#!/bin/bash
# per-phase shader-clock sums of wave 0 (RS_EXP_TIMING build of units 0 and 4): the row GEMM's loop with fp32 MFMAs and with the split products
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04bm; mkdir -p $O
cd $R
export REPSURF_HIP_LIB=$R/build_exp/librepsurf_TIMING.so
for shape in "4096 512 1024" "49152 128 256" "66560 64 128"; do
  echo "== split products, $shape"; timeout 100 python tools/gemm_bench.py timing $shape 2>&1 | grep -v amdgpu.ids
  echo "== fp32 MFMAs, $shape"; RS_GEMM_SPLIT3=0 timeout 100 python tools/gemm_bench.py timing $shape 2>&1 | grep -v amdgpu.ids
done > $O/gemm_phase_timing.txt 2>&1
cat $O/gemm_phase_timing.txt
