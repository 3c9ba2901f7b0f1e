#!/bin/bash
# round 4, after the split-product GEMMs became the default: GPU tests + parity report, smoke, the driver-form bench lines (cls, seg), the
# secondary lines, the same-box A/B against the fp32 MFMA instances, rocprofv3 kernel stats of the replayed steps, MFMA-busy counters.
# -> gpurun_out/r04fin (copied into profiles/r04/ by hand: the names are the ones DESIGN.md cites)
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04fin; mkdir -p $O
export REPSURF_PARITY_REPORT=$R/$O/parity_report.jsonl
rm -f $REPSURF_PARITY_REPORT
timeout 600 python -m pytest tests -q -m gpu --timeout 600 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log | cut -c 1-200
unset REPSURF_PARITY_REPORT
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_cls.json 2> $O/bench_cls.err; echo "cls rc=$?"
timeout 600 python bench.py --workload seg > $O/bench_seg.json 2> $O/bench_seg.err; echo "seg rc=$?"
b() { timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
{
echo "same box, bench.py --steps 30 --no-cpu-baseline --no-kernel-timing: ms per step, clouds/s"
echo "cls RS_GEMM_SPLIT3=0 (fp32 MFMAs)                 $(RS_GEMM_SPLIT3=0 b)"
echo "cls default                                       $(b)"
echo "seg RS_GEMM_SPLIT3=0 (fp32 MFMAs)                 $(RS_GEMM_SPLIT3=0 b --workload seg)"
echo "seg default                                       $(b --workload seg)"
} > $O/split3_step_ab.txt 2>&1; cat $O/split3_step_ab.txt
cd /tmp && export TMPDIR=/tmp
for WL in cls seg; do
  D=$R/$O/prof_$WL; mkdir -p $D
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o graph -- python $R/bench.py --no-cpu-baseline --workload $WL --steps 20 --warmup 3 --no-kernel-timing > $D/graph.log 2>&1; echo "$WL trace rc=$?"
  (cd $R && python tools/kernel_stats_by_grid.py $D/graph_kernel_trace.csv > $R/$O/${WL}_graph_kernel_stats_by_grid.csv 2>/dev/null)
  cp $D/graph_kernel_stats.csv $R/$O/${WL}_graph_kernel_stats.csv
  rm -rf $D
done
cd $R; timeout 300 python bench.py --no-cpu-baseline --model repsurf_ssg_umb_2x > $O/bench_cls_2x.json 2>/dev/null; echo "2x rc=$?"
cd /tmp
D=$R/$O/pmc; mkdir -p $D
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D -o pmc_mfma -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-graph --no-kernel-timing > $D/pmc.log 2>&1; echo "pmc rc=$?"
cd $R
python - <<PY > $O/cls_mfma_busy.txt 2>&1
import csv, collections, glob
f = glob.glob("$O/pmc/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:90]
    v = float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": acc[k][0] += v; acc[k][2] += 1
    elif r["Counter_Name"] == "SQ_BUSY_CYCLES": acc[k][1] += v
print("kernel, dispatches, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, ratio (raw sums over the dispatches of a 2-step eager run; MI355X_MICROARCH.md for units)")
for k, (m, b, n) in sorted(acc.items(), key=lambda x: -x[1][0])[:14]:
    print(f"{k}, {n}, {m:.0f}, {b:.0f}, {m / b if b else 0:.4f}")
PY
rm -rf $O/pmc; head -8 $O/cls_mfma_busy.txt | cut -c1-200
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open("$f").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$f".split("/")[-1], d.get("value"), d.get("ms_per_step"), r.get("frac"), r.get("avg_launch_us"), r.get("dims"), (r.get("all_mfma_launches") or {}).get("frac"), r.get("mfma_pipe"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print("$f", "ERR", e)
PY
done
