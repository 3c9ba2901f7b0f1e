#!/bin/bash
# Ordered kernel list (start, duration, name) of ONE replayed training step, from a rocprofv3 kernel trace of bench.py: the step
# between the last two Adam launches.  GPU box:  tools/step_kernel_sequence.sh [extra bench.py flags, e.g. --workload seg]
# -> gpurun_out/trace/last_step_sequence.txt   (durations of concurrent streams overlap: use it for counts and order)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace; rm -rf $O; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O -o eager -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
# last step = last quarter of the launches: find the last fps_reg_kernel<8, 2 start as the step marker
idx=[i for i,n in enumerate(names) if "adam_kernel" in n]
groups=[]
for i in idx:
    if groups and i == groups[-1][-1] + 1: groups[-1].append(i)
    else: groups.append([i])
print("kernels total", len(names), "adam groups", [(g[0], len(g)) for g in groups[-4:]])
import re
seq=rows[groups[-2][-1]+1:groups[-1][-1]+1]
out=open("$O/last_step_sequence.txt","w")
t0=int(seq[0]["Start_Timestamp"])
for r in seq:
    n=r["Kernel_Name"]
    n=re.sub(r"\(anonymous namespace\)::","",n)
    out.write("%9.1f us  %7.1f us  %s\n" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, n[:150]))
out.close()
print("wrote", len(seq))
PY
rm -f $O/*/*.csv $O/*.csv 2>/dev/null; find $O -name "*.csv" -delete; ls $O
