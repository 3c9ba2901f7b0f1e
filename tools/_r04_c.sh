#!/bin/bash
# round 4, call C: merged constructor launches (grid meeting), SyncBN tests, head rows path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py -q -m gpu -x -k "constructor or umbrella" --timeout 600 > $O/umb_tests.log 2>&1; echo "umb tests rc=$?"; tail -5 $O/umb_tests.log
timeout 300 python tools/umb_bench.py 256 > $O/umb_bench.txt 2>&1; echo "umb bench rc=$?"; cat $O/umb_bench.txt
REPSURF_UMB_MERGED=0 timeout 300 python tools/umb_bench.py 256 > $O/umb_bench_sep.txt 2>&1; cat $O/umb_bench_sep.txt
timeout 900 python -m pytest tests/test_syncbn_gpu.py -q -m gpu -x --timeout 600 -s > $O/syncbn.log 2>&1; echo "syncbn rc=$?"; tail -15 $O/syncbn.log
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
  REPSURF_UMB_MFMA=0 one cls_valu --steps 40 --warmup 10
  one cls_mfma --steps 40 --warmup 10
done | tee $O/ab.txt
REPSURF_UMB_MFMA=0 one seg_valu --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
one seg_mfma --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.log | tail -3
