#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
export REPSURF_PARITY_REPORT=$GRAFT_REPO_ROOT/$O/parity_report.jsonl
rm -f $REPSURF_PARITY_REPORT
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.log
unset REPSURF_PARITY_REPORT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cls.json 2> $O/bench_cls.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03z/bench_cls.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["steps"], d["steps_timed"], d["roofline"]["frac"], d["roofline"]["all_mfma_launches"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["gpu_over_cpu"], d["roofline_ballquery"]["clouds_per_launch"]["2048"])
PY
