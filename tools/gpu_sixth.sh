#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_mlp_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider > gpurun_out/pytest_mlp.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_mlp.log
timeout 600 python bench.py --steps 30 --warmup 3 --breakdown gpurun_out/breakdown_g.json > gpurun_out/bench_graph.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_graph.log
bash tools/gpu_profile.sh r01_c > gpurun_out/profile_run.log 2>&1
tail -n 3 gpurun_out/pytest_mlp.log; tail -n 2 gpurun_out/bench_graph.log | cut -c1-200; tail -5 gpurun_out/profile_run.log
