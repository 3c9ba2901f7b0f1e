#!/bin/bash
# GPU pass 2: HIP shared-MLP executor — unit parity, diagnostics, model parity, bench, rocprof
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export REPSURF_MLP=hip
timeout 600 python tools/mlp_diag.py > gpurun_out/mlp_diag.log 2>&1; echo "diag rc=$?" >> gpurun_out/mlp_diag.log
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=40 --timeout=600 -p no:cacheprovider > gpurun_out/pytest_mlp.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_mlp.log
timeout 600 python bench.py --steps 20 --warmup 3 --breakdown gpurun_out/breakdown_hip.json > gpurun_out/bench_hip.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_hip.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof2 -o r1b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/rocprof2.log 2>&1; echo "rocprof rc=$?" >> $R/gpurun_out/rocprof2.log
cd $R
python - <<'PY'
import sqlite3, csv, glob
for f in glob.glob('gpurun_out/prof2/*.db'):
    cur = sqlite3.connect(f).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open('gpurun_out/top_kernels_hip.csv','w',newline='') as o:
        w = csv.writer(o); w.writerow(['kernel','calls','total_us','avg_us','percent'])
        for r in rows: w.writerow([r[0][:160], r[1], round(r[2],3), round(r[3],3), round(r[4],3)])
PY
rm -rf gpurun_out/prof2
tail -n 4 gpurun_out/mlp_diag.log; tail -n 4 gpurun_out/pytest_mlp.log; tail -n 3 gpurun_out/bench_hip.log
