#!/bin/bash
# A/B on one box: step times of the library built with / without SLP vectorization in the geometry translation units (build_exp/lib_*.so)
for rep in 1 2; do
  for l in lib_slp lib_all_geom lib_side_only; do
    for w in cls seg; do
      REPSURF_HIP_LIB=$PWD/build_exp/$l.so python bench.py --workload $w --no-extra-legs --no-alt-arithmetic --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l $w', d['ms_per_step'], d.get('geometry_ms', ''))"
    done
  done
done
