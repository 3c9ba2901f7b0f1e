#!/bin/bash
# round 4, call D: constructor at 168 VGPRs, light meeting (sc1 stores / loads), SyncBN fix, in-step trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py -q -m gpu -x -k "constructor or umbrella" --timeout 600 > $O/umb_tests.log 2>&1; echo "umb tests rc=$?"; tail -5 $O/umb_tests.log
timeout 300 python tools/umb_bench.py 256 > $O/umb_bench.txt 2>&1; echo "umb bench (merged) rc=$?"; cat $O/umb_bench.txt
REPSURF_UMB_MERGED=0 timeout 300 python tools/umb_bench.py 256 > $O/umb_bench_sep.txt 2>&1; echo "separate:"; cat $O/umb_bench_sep.txt
timeout 900 python -m pytest tests/test_syncbn_gpu.py -q -m gpu -x --timeout 600 -s > $O/syncbn.log 2>&1; echo "syncbn rc=$?"; tail -6 $O/syncbn.log
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
  REPSURF_UMB_MFMA=0 one cls_valu --steps 40 --warmup 10
  REPSURF_UMB_MERGED=0 one cls_mfma_sep --steps 40 --warmup 10
  one cls_mfma_merged --steps 40 --warmup 10
done | tee $O/ab.txt
REPSURF_UMB_MFMA=0 one seg_valu --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
one seg_mfma --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
export TMPDIR=/tmp
for tag in valu mfma; do
  D=$GRAFT_REPO_ROOT/$O/trace_$tag; mkdir -p $D
  if [ $tag = valu ]; then export REPSURF_UMB_MFMA=0; else unset REPSURF_UMB_MFMA; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o graph -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-kernel-timing --no-cpu-baseline > $D/run.log 2>&1)
  f=$(find $D -name "*kernel_trace.csv" | head -1)
  python tools/kernel_stats_by_grid.py $f > $O/stats_$tag.csv
  rm -rf $D
done
unset REPSURF_UMB_MFMA
head -12 $O/stats_mfma.csv
