#!/bin/bash
# round 4, call F: 16-byte partial reduction, vectorized compact gather / scatter, in-graph GEMM timing on the bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.log | tail -3; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
  one cls --steps 40 --warmup 10
done | tee $O/ab.txt
one seg --workload seg --steps 20 --warmup 5 | tee -a $O/ab.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cls.json 2> $O/bench_cls.err; echo "bench rc=$?"; tail -3 $O/bench_cls.err
python - <<PY
import json
d=json.loads(open("$O/bench_cls.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r.get("eager_avg_launch_us"), r["dims"])
print(json.dumps(r["all_mfma_launches"])[:600])
PY
