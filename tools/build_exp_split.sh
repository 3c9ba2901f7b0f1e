#!/bin/bash
# Experiment variants of UNIT 4 of mlp.hip (the split-product GEMMs: -DRS_EXP_<NAME> -DRS_MLP_TU=4) linked against the CURRENT objects of the
# product build.  Run HERE after `make`; the .so files travel to the GPU box under build_exp/ (git-ignored; ~21 MB each: a push of
# several takes tens of seconds) and are selected with REPSURF_HIP_LIB=build_exp/librepsurf_<NAME>.so.
#   (SP_PIPE, the software-pipelined loop, was measured in round 5 and removed: profiles/r05/sp_pipe_ab.txt)
#   tools/build_exp_split.sh SP_ONE_MFMA SP_ONE_FRAG SP_ONE_STORE SP_ONE_MFMA+SP_ONE_FRAG+SP_ONE_STORE      the what-if builds of round 4 (profiles/r04/gemm_split3_whatif.txt)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_exp
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wno-unused-function"
for v in "$@"; do
  defs=""; for d in ${v//+/ }; do defs="$defs -DRS_EXP_$d"; done
  n=${v//+/_}
  ( /opt/rocm/bin/hipcc $FL $defs -DRS_MLP_TU=4 -x hip -c repsurf_amd/csrc/mlp.hip -o build_exp/mlp_split_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_exp/librepsurf_$n.so $(ls build/*.o | grep -v "^build/mlp_split.hip.o") build_exp/mlp_split_$n.o &&
    rm -f build_exp/mlp_split_$n.o ) &
done
wait
ls -la build_exp/*.so
