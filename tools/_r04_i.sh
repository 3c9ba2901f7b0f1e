#!/bin/bash
# round 4, call I: direct epilogue for 128-row tiles (A/B against the LDS epilogue build), tail fix, bench estimator
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_seg_gpu.py tests/test_model_gpu.py tests/test_parity_full_gpu.py -q -m gpu -x --timeout 600 > $O/some_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/some_tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/some_tests.log | head
one() { local tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>$O/err_$tag.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['value'])"; }
for r in 1 2; do
  one seg_direct128 --workload seg --steps 20 --warmup 5
  REPSURF_HIP_LIB=build_exp/librepsurf_nod128.so one seg_lds128 --workload seg --steps 20 --warmup 5
done | tee $O/ab.txt
one cls --steps 40 --warmup 10 | tee -a $O/ab.txt
REPSURF_HIP_LIB=build_exp/librepsurf_nod128.so one cls_lds128 --steps 40 --warmup 10 | tee -a $O/ab.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench_cls.json 2> $O/bench_cls.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_cls.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r.get("eager_avg_launch_us"), r.get("alone_avg_launch_us"), r["dims"])
print(json.dumps(r["all_mfma_launches"])[:700])
PY
