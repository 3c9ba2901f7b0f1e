#!/bin/bash
# Build experiment variants of the library (no operand loads / no MFMA) next to the product build.
# Run HERE (needs hipcc); the variant .so files travel to the GPU box under build_exp/ (git-ignored).
set -e
cd "$(dirname "$0")/.."
mkdir -p build_exp
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wno-unused-function"
for v in NOLOAD NOMFMA TIMING; do
  /opt/rocm/bin/hipcc $FL -DRS_EXP_$v -x hip -c repsurf_amd/csrc/mlp.hip -o build_exp/mlp_$v.o &
done
wait
for v in NOLOAD NOMFMA TIMING; do
  objs=$(ls build/*.o | grep -v "^build/mlp")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_exp/librepsurf_$v.so $objs build_exp/mlp_$v.o
done
ls -la build_exp/*.so
