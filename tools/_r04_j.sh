#!/bin/bash
# round 4, call J: fused interpolation + skip + ReLU, full GPU tests, seg bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.log | tail -3; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do one seg --workload seg --steps 20 --warmup 5; done | tee $O/ab.txt
timeout 900 python bench.py --workload seg > $O/bench_seg.json 2> $O/bench_seg.err; echo "seg bench rc=$?"; tail -2 $O/bench_seg.err
python - <<PY
import json
d=json.loads(open("$O/bench_seg.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r.get("eager_avg_launch_us"), r["dims"], r["bound"])
print(json.dumps(r["all_mfma_launches"])[:600])
print(json.dumps(d["cpu_baseline"])[:900]); print(d.get("gpu_over_cpu"))
PY
