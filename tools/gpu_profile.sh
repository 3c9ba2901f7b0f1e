#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats + PMC passes (each counter set in its own run)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
TAG=${1:-r01_c}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-graph"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o trace -- $CMD > $R/gpurun_out/prof_trace.log 2>&1; echo "trace rc=$?" >> $R/gpurun_out/prof_trace.log
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_')
  timeout 900 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/prof_$TAG -o pmc_$N -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > $R/gpurun_out/prof_pmc_$N.log 2>&1; echo "pmc $N rc=$?" >> $R/gpurun_out/prof_pmc_$N.log
done
cd $R
find gpurun_out/prof_$TAG -type f | head -40
python - "$TAG" <<'PY'
import csv, glob, os, sys, collections, json
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_c"
base = f"gpurun_out/prof_{tag}"
out = {}
def short(n): return n.split("(")[0][-70:]
for f in glob.glob(base + "/**/*kernel_stats*.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats_file"] = f
    out["kernel_stats"] = [{k: r[k] for k in r} for r in rows[:60]]
for f in glob.glob(base + "/**/pmc_*counter_collection*.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = short(r.get("Kernel_Name", "?"))
        c = r.get("Counter_Name"); v = float(r.get("Counter_Value", 0) or 0)
        a = agg[k][c]; a[0] += v; a[1] += 1
    out[os.path.basename(f)] = {k: {c: {"sum": v[0], "n": v[1], "avg": v[0] / max(v[1], 1)} for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open(f"gpurun_out/prof_{tag}_summary.json", "w"), indent=1)
print("summary keys", list(out.keys()))
PY
