#!/bin/bash
# Round-6 validation of the two-stream fix on one box: the product library (geometry units without vectorizer-made packed fp32) against
# build_exp/lib_slp.so (the same tree built WITH it), same probes.  Output: gpurun_out/two_stream_validation.txt
mkdir -p gpurun_out
{
for lib in product lib_slp; do
  if [ $lib = product ]; then unset REPSURF_HIP_LIB; else export REPSURF_HIP_LIB=$PWD/build_exp/$lib.so; fi
  [ $lib != product ] && [ ! -f "$REPSURF_HIP_LIB" ] && continue
  echo "== library: $lib"
  for i in 1 2 3; do python tools/pipelined_flake.py 8000 2>&1 | grep -i "deviating"; done
  for i in 1 2 3; do OVERLAP=1 python tools/ragged_flake3.py 200 2>&1 | grep "calls"; done
  python tools/victim_probe.py 20000 60 4096 8 9 2>&1 | grep "fan-feature" | cut -c1-110
done
} > gpurun_out/two_stream_validation.txt 2>&1
cat gpurun_out/two_stream_validation.txt
