#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
timeout 600 python -m pytest tests/test_mlp_gpu.py tests/test_syncbn_gpu.py -q -m gpu -x --timeout 600 > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/tests.log | head
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
  one cls_prefetch --steps 40 --warmup 10
  REPSURF_HIP_LIB=build_exp/librepsurf_noprefetch.so one cls_before --steps 40 --warmup 10
done | tee $O/ab.txt
timeout 300 python bench.py --gpus 2 --dry-run > $O/bench_spawn_dry_run.json 2> $O/bench_spawn.err; echo "spawn rc=$?"; tail -1 $O/bench_spawn_dry_run.json
