#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
REPSURF_TAIL_STREAM=1 timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_graph_gpu.py -q -m gpu -x --timeout 600 > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/tests.log | head
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
  REPSURF_TAIL_STREAM=1 one cls_tailstream --steps 40 --warmup 10
  one cls_before --steps 40 --warmup 10
done | tee $O/ab.txt
for r in 1 2; do
  REPSURF_TAIL_STREAM=1 one seg_tailstream --workload seg --steps 20 --warmup 5
  one seg_before --workload seg --steps 20 --warmup 5
done | tee -a $O/ab.txt
tail -3 $O/err_cls_tailstream.txt
