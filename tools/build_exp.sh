#!/bin/bash
# Experiment variants of mlp.hip (-DRS_EXP_<NAME>) linked against the CURRENT objects of the product build, so their ABI
# matches the Python binding.  Run HERE after `make`; the .so files travel to the GPU box under build_exp/ (git-ignored)
# and are selected with REPSURF_HIP_LIB=build_exp/librepsurf_<NAME>.so.     tools/build_exp.sh LDS_EPILOGUE NO_EARLY_PREFETCH ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build_exp
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wno-unused-function"
for v in "$@"; do
  defs=""; for d in ${v//+/ }; do defs="$defs -DRS_EXP_$d"; done      # NAME1+NAME2: several RS_EXP_ switches in one variant
  ( /opt/rocm/bin/hipcc $FL $defs -DRS_MLP_TU=2 -x hip -c repsurf_amd/csrc/mlp.hip -o build_exp/mlp_$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_exp/librepsurf_$v.so $(ls build/*.o | grep -v "^build/mlp") build_exp/mlp_$v.o ) &
done
wait
ls -la build_exp/*.so
