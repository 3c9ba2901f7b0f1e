#!/bin/bash
# round 3 evidence: bench lines (all workloads / data modes), rocprof kernel stats + PMC traffic for cls and seg, full GPU tests with the parity report
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
export REPSURF_PARITY_REPORT=$GRAFT_REPO_ROOT/$O/parity_report.jsonl
rm -f $REPSURF_PARITY_REPORT
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c 1-200
unset REPSURF_PARITY_REPORT
timeout 900 python bench.py > $O/bench_cls.json 2> $O/bench_cls.err; echo "cls rc=$?"
timeout 600 python bench.py --no-cpu-baseline --data real > $O/bench_cls_real.json 2>/dev/null; echo "real rc=$?"
REPSURF_COMPACT=0 timeout 600 python bench.py --no-cpu-baseline > $O/bench_cls_dense.json 2>/dev/null; echo "dense rc=$?"
timeout 600 python bench.py --no-cpu-baseline --model repsurf_ssg_umb_2x > $O/bench_cls_2x.json 2>/dev/null; echo "2x rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-pipeline > $O/bench_cls_nopipe.json 2>/dev/null; echo "nopipe rc=$?"
timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --batch 64 --points 2048 > $O/bench_cls_bf16_b64.json 2>/dev/null; echo "bf16 b64 rc=$?"
timeout 900 python bench.py --workload seg > $O/bench_seg.json 2> $O/bench_seg.err; echo "seg rc=$?"
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$f".split("/")[-1], d["value"], d["ms_per_step"], d.get("steps_timed"), d["config"].get("distinct_slot_fraction_sa1_sa2"), r.get("frac"), r.get("dims"), r.get("all_mfma_launches"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("kind"))
except Exception as e: print("$f", "ERR", e)
PY
done
for D in 0 1 2 3 4; do echo "== dbg $D"; RS_BALLQUERY_DBG=$D timeout 120 python tools/ballquery_bench.py 2>&1 | grep "N=1024 S=512 r=0.2"; done > $O/ballquery_phases.txt; cat $O/ballquery_phases.txt
bash tools/gpu_profile.sh r03 cls > $O/profile_cls.log 2>&1; tail -5 $O/profile_cls.log
bash tools/gpu_profile.sh r03 seg > $O/profile_seg.log 2>&1; tail -3 $O/profile_seg.log
