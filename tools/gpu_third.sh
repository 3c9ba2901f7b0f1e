#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
export REPSURF_MLP=hip
timeout 300 python tools/mlp_unit_diag.py > gpurun_out/unit_diag.log 2>&1; echo "unit rc=$?" >> gpurun_out/unit_diag.log
timeout 600 python tools/mlp_diag.py > gpurun_out/mlp_diag.log 2>&1; echo "diag rc=$?" >> gpurun_out/mlp_diag.log
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=40 --timeout=600 -p no:cacheprovider > gpurun_out/pytest_mlp.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_mlp.log
timeout 600 python bench.py --steps 20 --warmup 3 --breakdown gpurun_out/breakdown_hip.json > gpurun_out/bench_hip.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_hip.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_hip_notiming.log 2>&1
grep -E "BAD|EXC|rc=" gpurun_out/unit_diag.log | head; tail -n 3 gpurun_out/pytest_mlp.log; tail -n 2 gpurun_out/bench_hip.log | cut -c1-400; tail -n 1 gpurun_out/bench_hip_notiming.log | cut -c1-200
