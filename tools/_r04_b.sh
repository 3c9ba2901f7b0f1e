#!/bin/bash
# round 4, call B: constructor MLP restructured (whole tile preloaded, 8 waves per workgroup)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout 600 python -m pytest tests/test_mlp_gpu.py -q -m gpu -x -k "constructor or umbrella" --timeout 300 > $O/umb_tests.log 2>&1; echo "umb tests rc=$?"; tail -5 $O/umb_tests.log
timeout 300 python tools/umb_bench.py 128 256 > $O/umb_bench.txt 2>&1; echo "umb bench rc=$?"; cat $O/umb_bench.txt
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
  REPSURF_UMB_MFMA=0 one cls_valu --steps 40 --warmup 10
  one cls_mfma --steps 40 --warmup 10
done | tee $O/ab.txt
REPSURF_UMB_MFMA=0 one seg_valu --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
one seg_mfma --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
