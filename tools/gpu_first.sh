#!/bin/bash
# GPU pass: parity tests + smoke + bench + rocprof kernel trace (outputs under gpurun_out/)
mkdir -p gpurun_out
export REPSURF_MLP=${REPSURF_MLP:-torch}
R=$GRAFT_REPO_ROOT
cd $R
python -c "import torch; print(torch.cuda.get_device_name(0)); import os; print('cpus', os.cpu_count())" > gpurun_out/dev.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 --timeout=600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 --breakdown gpurun_out/breakdown.json > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?" >> $R/gpurun_out/rocprof.log
cd $R
find gpurun_out/prof -name '*stats*' | head; 
tail -3 gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.log gpurun_out/rocprof.log
