#!/bin/bash
# round 4, call G: scatter fix, in-graph GEMM timing (safe pool), wgrad_narrow4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout 600 python -m pytest tests/test_mlp_gpu.py tests/test_seg_gpu.py tests/test_model_gpu.py -q -m gpu -x --timeout 600 > $O/some_tests.log 2>&1; echo "mlp/seg/model tests rc=$?"; grep -E "passed|failed" $O/some_tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/some_tests.log | head
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
  one cls --steps 40 --warmup 10
  RS_WGRAD_NARROW4=0 one cls_no_narrow4 --steps 40 --warmup 10
done | tee $O/ab.txt
one seg --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
RS_WGRAD_NARROW4=0 one seg_no_narrow4 --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cls.json 2> $O/bench_cls.err; echo "bench rc=$?"; tail -3 $O/bench_cls.err
python - <<PY
import json
d=json.loads(open("$O/bench_cls.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r.get("eager_avg_launch_us"), r["dims"])
print(json.dumps(r["all_mfma_launches"])[:700])
PY
export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/$O/trace; mkdir -p $D
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o graph -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-kernel-timing --no-cpu-baseline > $D/run.log 2>&1)
f=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/kernel_stats_by_grid.py $f > $O/stats_cls.csv; rm -rf $D
grep -E "compact|backward_tail|reduce_partials|wgrad_narrow" $O/stats_cls.csv | cut -c1-120
