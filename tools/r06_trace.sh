#!/bin/bash
# Round 6: in-step kernel durations (rocprofv3 --kernel-trace of the replayed step, by kernel instance AND grid) of the round-5 tree, the working
# tree and experiment libraries, one box.     gpurun -- 'bash tools/r06_trace.sh tag [cls|seg] [NAME ...]'
R=$GRAFT_REPO_ROOT
TAG=$1; WL=${2:-cls}; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { # name tree lib
  local D=$O/trace_$1
  mkdir -p $D
  ( cd /tmp && REPSURF_HIP_LIB=$3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o graph -- python $2/bench.py --no-cpu-baseline --no-alt-arithmetic --workload $WL --steps 20 --warmup 3 --no-kernel-timing > $D/graph.log 2>&1 )
  python $R/tools/kernel_stats_by_grid.py $D/graph_kernel_trace.csv > $O/${WL}_by_grid_$1.csv 2>/dev/null
  cp $D/graph_kernel_stats.csv $O/${WL}_kernel_stats_$1.csv 2>/dev/null
  tail -1 $D/graph.log | cut -c1-200
  rm -rf $D
}
one head $R/build_exp/head_tree ""
one product $R ""
for v in "$@"; do one $v $R $R/build_exp/librepsurf_$v.so; done
cd $R
python - <<PY
import csv, glob, os
O = "$O"
tabs = {}
for f in sorted(glob.glob(f"{O}/${WL}_by_grid_*.csv")):
    name = os.path.basename(f)[len("${WL}_by_grid_"):-4]
    tabs[name] = {(r["kernel"], r["workgroups_x"], r["workgroups_y"], r["workgroups_z"]): (float(r["avg_us"]), int(r["calls"])) for r in csv.DictReader(open(f))}
names = list(tabs)
base = tabs.get("head", {})
steps = 23.0
print("kernel,grid," + ",".join(f"{n}_us" for n in names) + ",launches_per_step")
fam = {n: 0.0 for n in names}
tot = {n: 0.0 for n in names}
for k, (us, calls) in sorted(base.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    row = [tabs[n].get(k, (float("nan"), 0))[0] for n in names]
    per = round(calls / steps)
    for n, v in zip(names, row):
        if v == v:
            tot[n] += v * calls / steps
            if k[0].startswith(("gemm_", "wgrad_")):
                fam[n] += v * calls / steps
    if us * calls / steps > 4.0:
        print(f'"{k[0]}",{k[1]}x{k[2]}x{k[3]},' + ",".join(f"{v:.2f}" for v in row) + f",{per}")
print("GEMM-family us per step (approx: calls / 23 steps)," + ",".join(f"{n}={fam[n]:.1f}" for n in names))
print("all kernels us per step," + ",".join(f"{n}={tot[n]:.1f}" for n in names))
PY
