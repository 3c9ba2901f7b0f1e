#!/bin/bash
# round 5 evidence (GPU box): the whole GPU suite twice (-x), the sharded soaks, bench lines, rocprof kernel stats + PMC for cls / seg.
# -> gpurun_out/r06ev, copied into profiles/r06/ by tools/copy_evidence.sh r06.      gpurun --timeout 2400 -- 'bash tools/r06_evidence.sh [part ...]'
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ev; mkdir -p $O
PARTS=${@:-tests soak bench profile}
for PART in $PARTS; do case $PART in
tests)
  export REPSURF_TEST_LOGDIR=$GRAFT_REPO_ROOT/$O/dist_logs
  : > $O/gpu_tests_x2.log
  for i in 1 2; do
    [ $i = 1 ] && export REPSURF_PARITY_REPORT=$GRAFT_REPO_ROOT/$O/parity_report.jsonl && rm -f $REPSURF_PARITY_REPORT
    echo "=== run $i: python -m pytest tests -x -q -m gpu" >> $O/gpu_tests_x2.log
    timeout 1500 python -m pytest tests -x -q -m gpu >> $O/gpu_tests_x2.log 2>&1; echo "=== run $i rc=$?" >> $O/gpu_tests_x2.log
    unset REPSURF_PARITY_REPORT
  done
  grep -E "^=== run|passed|failed" $O/gpu_tests_x2.log | cut -c1-160
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log ;;
soak)
  #  50 steps x 5 processes of the forced-RCCL captured step on a 1-rank group, and of bench.py --gpus 2 over gloo on one GPU
  : > $O/sharded_soak.txt
  REPSURF_SOAK_STEPS=50 python tools/soak_sharded.py $O/soak 5 _pipelined_sharded_step_single_rank_process_group 2>&1 | sed 's/ \[rank0\].*//' >> $O/sharded_soak.txt
  for i in 1 2 3 4 5; do
    REPSURF_DIST_BACKEND=gloo REPSURF_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 50 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/soak_bench2_$i.json 2> $O/soak_bench2_$i.err
    echo "bench.py --gpus 2 (gloo, two ranks on one GPU) 50 steps, process $i: rc $? $(tail -1 $O/soak_bench2_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], 'allreduce_us', d['config'].get('allreduce_us'))" 2>&1 | tail -1)" >> $O/sharded_soak.txt
  done
  echo "== sharded step forms, 1-rank RCCL group, collective forced (tools/sharded_time.py)" >> $O/sharded_soak.txt
  REPSURF_FORCE_ALLREDUCE=1 timeout 300 python tools/sharded_time.py pipe pipe_sharded graphed_sharded 2>&1 | grep "ms/step\|rror" | cut -c1-200 >> $O/sharded_soak.txt
  cat $O/sharded_soak.txt ;;
bench)
  timeout 900 python bench.py > $O/bench_cls.json 2> $O/bench_cls.err; echo "cls rc=$?"
  timeout 600 python bench.py --no-cpu-baseline --no-alt-arithmetic --data real > $O/bench_cls_real.json 2>/dev/null; echo "real rc=$?"
  REPSURF_COMPACT=0 timeout 600 python bench.py --no-cpu-baseline --no-alt-arithmetic > $O/bench_cls_dense.json 2>/dev/null; echo "dense rc=$?"
  timeout 600 python bench.py --no-cpu-baseline --no-alt-arithmetic --model repsurf_ssg_umb_2x > $O/bench_cls_2x.json 2>/dev/null; echo "2x rc=$?"
  timeout 600 python bench.py --no-cpu-baseline --no-alt-arithmetic --no-pipeline > $O/bench_cls_nopipe.json 2>/dev/null; echo "nopipe rc=$?"
  timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --batch 64 --points 2048 > $O/bench_cls_bf16_b64.json 2>/dev/null; echo "bf16 b64 rc=$?"
  timeout 900 python bench.py --workload seg > $O/bench_seg.json 2> $O/bench_seg.err; echo "seg rc=$?"
  timeout 600 python bench.py --workload seg --ragged --steps 40 --warmup 5 > $O/bench_seg_ragged.json 2> $O/bench_seg_ragged.err; echo "seg ragged rc=$?"
  timeout 600 python bench.py --workload seg --ragged --batch 8 --points 80000 --steps 16 --warmup 3 > $O/bench_seg_ragged_s3dis.json 2> $O/bench_seg_ragged_s3dis.err; echo "seg ragged S3DIS-sized rc=$?"
  for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$f".split("/")[-1], d.get("value"), d.get("ms_per_step"), d.get("arithmetic"), d.get("fp32_mfma_ms_per_step"), r.get("frac"), r.get("avg_launch_us"), (r.get("all_mfma_launches") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"), ((d.get("roofline_ballquery") or {}).get("clouds_per_launch") or {}).get("2048"))
except Exception as e: print("$f", "ERR", e)
PY
  done ;;
profile)
  bash tools/gpu_profile.sh r06 cls > $O/profile_cls.log 2>&1; tail -5 $O/profile_cls.log
  bash tools/gpu_profile.sh r06 seg > $O/profile_seg.log 2>&1; tail -3 $O/profile_seg.log ;;
esac; done
