#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04fin; mkdir -p $O
cd $R
timeout 300 python bench.py --no-cpu-baseline --data real > $O/bench_cls_real.json 2>/dev/null; echo "real rc=$?"
REPSURF_COMPACT=0 timeout 300 python bench.py --no-cpu-baseline > $O/bench_cls_dense.json 2>/dev/null; echo "dense rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-pipeline > $O/bench_cls_nopipe.json 2>/dev/null; echo "nopipe rc=$?"
for f in real dense nopipe; do python -c "import json; d=json.loads(open('$O/bench_cls_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
