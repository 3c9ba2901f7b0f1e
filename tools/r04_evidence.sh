#!/bin/bash
# round 4 evidence: full GPU tests with the parity report, bench lines (all workloads / data modes), rocprof kernel stats + PMC traffic
# for cls and seg, constructor / grid-meeting micro-benchmarks, the sharded step on a 1-rank RCCL group.  -> gpurun_out/r04ev,
# copied into profiles/r04/ by tools/copy_evidence.sh r04
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ev; mkdir -p $O
export REPSURF_PARITY_REPORT=$GRAFT_REPO_ROOT/$O/parity_report.jsonl
rm -f $REPSURF_PARITY_REPORT
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c 1-200
unset REPSURF_PARITY_REPORT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_cls.json 2> $O/bench_cls.err; echo "cls rc=$?"
timeout 600 python bench.py --no-cpu-baseline --data real > $O/bench_cls_real.json 2>/dev/null; echo "real rc=$?"
REPSURF_COMPACT=0 timeout 600 python bench.py --no-cpu-baseline > $O/bench_cls_dense.json 2>/dev/null; echo "dense rc=$?"
timeout 600 python bench.py --no-cpu-baseline --model repsurf_ssg_umb_2x > $O/bench_cls_2x.json 2>/dev/null; echo "2x rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-pipeline > $O/bench_cls_nopipe.json 2>/dev/null; echo "nopipe rc=$?"
timeout 600 python bench.py --no-cpu-baseline --dtype bf16 --batch 64 --points 2048 > $O/bench_cls_bf16_b64.json 2>/dev/null; echo "bf16 b64 rc=$?"
timeout 900 python bench.py --workload seg > $O/bench_seg.json 2> $O/bench_seg.err; echo "seg rc=$?"
timeout 600 python bench.py --gpus 2 --dry-run > $O/bench_spawn_dry_run.json 2> $O/bench_spawn.err; echo "spawn dry-run rc=$?"; tail -1 $O/bench_spawn_dry_run.json
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$f".split("/")[-1], d.get("value"), d.get("ms_per_step"), d.get("steps_timed"), (d.get("config") or {}).get("distinct_slot_fraction_sa1_sa2"), r.get("frac"), r.get("avg_launch_us"), r.get("dims"), r.get("all_mfma_launches"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("kind"))
except Exception as e: print("$f", "ERR", e)
PY
done
timeout 300 python tools/umb_bench.py 256 > $O/umb_bench.txt 2>&1; cat $O/umb_bench.txt
timeout 120 tools/probes/grid_meet > $O/grid_meet.txt 2>&1; cat $O/grid_meet.txt
timeout 300 python tools/knn_grid_bench.py 2>&1 | grep -v amdgpu.ids > $O/knn_grid_bench.txt; cat $O/knn_grid_bench.txt
bash tools/_r04_ab.sh 2>&1 | grep umbrella_features > $O/umbrella_grid_bench.txt; cat $O/umbrella_grid_bench.txt
echo "== sharded step, 1-rank RCCL group, collective forced"; REPSURF_FORCE_ALLREDUCE=1 timeout 300 python tools/sharded_time.py pipe pipe_sharded pipe_sharded > $O/sharded_time.txt 2>&1; echo rc=$?; grep "ms/step\|rror" $O/sharded_time.txt | cut -c 1-300
bash tools/gpu_profile.sh r04 cls > $O/profile_cls.log 2>&1; tail -5 $O/profile_cls.log
bash tools/gpu_profile.sh r04 seg > $O/profile_seg.log 2>&1; tail -3 $O/profile_seg.log
