#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ar; mkdir -p $O
timeout 1500 python -m pytest tests/test_seg_gpu.py tests/test_parity_full_gpu.py tests/test_dropin.py -q -m gpu -x --timeout 900 > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR|Error" $O/tests.log | head
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
one seg_gather --workload seg --steps 20 --warmup 5
REPSURF_GATHER_BACKWARD=0 one seg_scatter --workload seg --steps 20 --warmup 5
done | tee $O/ab.txt
tail -3 $O/err_seg_gather.txt
