#!/bin/bash
# kernel timeline of one graph-replayed step (ordered, with queue ids) -> gpurun_out/timeline_step.csv
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -o tl -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing $BENCH_EXTRA > $R/gpurun_out/prof_tl.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_tl/**/tl_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'umbrella_kernel' in r['Kernel_Name']]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
print('kernels per step', len(step), 'span us', (t1 - t0) / 1e3, 'sum us', sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3)
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print('union busy us', busy / 1e3, 'idle us', (t1 - t0 - busy) / 1e3)
with open('gpurun_out/timeline_step.csv', 'w') as o:
    for r in step:
        o.write('%s,%d,%d,%s,%s\n' % (r['Kernel_Name'].replace(',', ';')[:90], int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r.get('Queue_Id', ''), r.get('Grid_Size', '')))
PY
rm -rf gpurun_out/prof_tl
tail -n 2 gpurun_out/prof_tl.log | cut -c1-200
