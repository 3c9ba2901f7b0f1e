#!/bin/bash
# round 4, call A: the matrix-pipe constructor MLP -- parity tests, micro-benchmark, step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 600 python -m pytest tests/test_mlp_gpu.py -q -m gpu -x -k "constructor or umbrella" --timeout 300 > $O/umb_tests.log 2>&1; echo "umb tests rc=$?"; tail -15 $O/umb_tests.log
timeout 300 python tools/umb_bench.py > $O/umb_bench.txt 2>&1; echo "umb bench rc=$?"; cat $O/umb_bench.txt
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
  REPSURF_UMB_MFMA=0 one cls_valu --steps 40 --warmup 10
  one cls_mfma --steps 40 --warmup 10
  REPSURF_UMB_MFMA=0 one seg_valu --workload seg --steps 20 --warmup 5
  one seg_mfma --workload seg --steps 20 --warmup 5
done | tee $O/ab.txt
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/gpu_tests.log
