#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ay; mkdir -p $O
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do one seg_foreach_copy --workload seg --steps 20 --warmup 5; done | tee $O/ab.txt
timeout 600 python -m pytest tests/test_seg_gpu.py -q -m gpu -x -k "model" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -1
