#!/bin/bash
# kernel timeline of graph replay: where does a step's wall time go?
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider > gpurun_out/pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_all.log
timeout 600 python bench.py --steps 30 --warmup 3 --breakdown gpurun_out/breakdown_g.json > gpurun_out/bench_graph.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_graph.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -o tl -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_tl.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_tl/**/tl_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last replay = last N kernels; find step boundaries via the umbrella_kernel launches
starts = [i for i, r in enumerate(rows) if 'umbrella_kernel' in r['Kernel_Name']]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
print('kernels per step', len(step), 'span us', (t1 - t0) / 1e3, 'sum us', sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3)
# union busy time
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print('union busy us', busy / 1e3, 'idle us', (t1 - t0 - busy) / 1e3)
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:60]
    agg[n][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); agg[n][1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:45]:
    print('%-62s %8.1f us %4d' % (n, t / 1e3, c))
with open('gpurun_out/timeline_step.csv', 'w') as o:
    for r in step:
        o.write('%s,%d,%d,%s\n' % (r['Kernel_Name'].replace(',', ';')[:80], int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r.get('Queue_Id', '')))
PY
rm -rf gpurun_out/prof_tl
grep -E "passed|failed" gpurun_out/pytest_all.log | tail -2; tail -n 2 gpurun_out/bench_graph.log | cut -c1-200
